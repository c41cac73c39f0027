// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales) on gfx950: which lane/byte feeds which
// product, and which lane's scale applies to which 32-value block.  Prints the max error of each hypothesis.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
  out[l] = c;
}
static float dec(unsigned char v) {  // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -f : f;
}
int main() {
  std::vector<unsigned char> A(64 * 32), B(64 * 32);
  std::vector<int> SA(64), SB(64);
  srand(7);
  for (auto& v : A) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f); }
  for (auto& v : B) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f); }
  for (int variant = 0; variant < 3; ++variant) {
    // variant 0: unit scales; 1: per-lane scale in byte 0; 2: per-lane scale replicated in all four bytes
    for (int l = 0; l < 64; ++l) {
      const int ea = variant ? 120 + rand() % 14 : 127, eb = variant ? 120 + rand() % 14 : 127;
      SA[l] = variant == 2 ? ea * 0x01010101 : ea;
      SB[l] = variant == 2 ? eb * 0x01010101 : eb;
    }
    void *da, *db, *dsa, *dsb, *dout;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 1024);
    hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, (const int*)dsa, (const int*)dsb, (f32x4*)dout);
    std::vector<float> out(256);
    hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
    // hypothesis H1: D[4*(l>>4)+q][l&15] = sum_g sum_i A[(r,g)][i] * B[(c,g)][i] * 2^(sa[(r,g)]-127) * 2^(sb[(c,g)]-127), lane (r,g) = r + 16 g
    // hypothesis H2: same products, but block g's scale comes from byte g of lane r (lanes 0..15)
    double e1 = 0, e2 = 0, mag = 0;
    for (int l = 0; l < 64; ++l)
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * (l >> 4) + q, c = l & 15;
        double s1 = 0, s2 = 0;
        for (int g = 0; g < 4; ++g) {
          double blk = 0;
          for (int i = 0; i < 32; ++i) blk += (double)dec(A[(r + 16 * g) * 32 + i]) * (double)dec(B[(c + 16 * g) * 32 + i]);
          s1 += blk * ldexp(1.0, (SA[r + 16 * g] & 0xff) - 127) * ldexp(1.0, (SB[c + 16 * g] & 0xff) - 127);
          s2 += blk * ldexp(1.0, ((SA[r] >> (8 * g)) & 0xff) - 127) * ldexp(1.0, ((SB[c] >> (8 * g)) & 0xff) - 127);
        }
        e1 = fmax(e1, fabs(s1 - out[l * 4 + q])); e2 = fmax(e2, fabs(s2 - out[l * 4 + q])); mag = fmax(mag, fabs(s1));
      }
    printf("variant %d: max|D| %.4g  err(H1 own-lane scale, byte 0) %.4g  err(H2 lanes 0-15 byte g) %.4g\n", variant, mag, e1, e2);
  }
  return 0;
}
