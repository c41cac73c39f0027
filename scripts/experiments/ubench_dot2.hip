// semantics probe of v_dot2c_f32_bf16 (builtin __builtin_amdgcn_fdot2_f32_bf16) on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* a, const unsigned* b, const float* c, float* o, float* o2) {
  int i = threadIdx.x;
  o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, b[i]), c[i], false);
  float alo = __uint_as_float(a[i] << 16), ahi = __uint_as_float(a[i] & 0xffff0000u);
  float blo = __uint_as_float(b[i] << 16), bhi = __uint_as_float(b[i] & 0xffff0000u);
  o2[i] = fmaf(ahi, bhi, fmaf(alo, blo, c[i]));
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
int main() {
  const int n = 64;
  unsigned ha[n], hb[n]; float hc[n], ho[n], ho2[n];
  for (int i = 0; i < n; ++i) {
    float x0 = 0.5f + 0.01f * i, x1 = -1.25f + 0.03f * i, y0 = 2.0f - 0.05f * i, y1 = 0.75f + 0.02f * i;
    ha[i] = f2bf(x0) | ((unsigned)f2bf(x1) << 16); hb[i] = f2bf(y0) | ((unsigned)f2bf(y1) << 16); hc[i] = 0.1f * i;
  }
  unsigned *a, *b; float *c, *o, *o2;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&o, n * 4); hipMalloc(&o2, n * 4);
  hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c, o, o2);
  hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ho2, o2, n * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; ++i) worst = fmax(worst, fabs(ho[i] - ho2[i]));
  printf("dot2 vs fma chain: max abs diff %.3e ; sample: dot2 %.6f fma %.6f | %.6f %.6f\n", worst, ho[3], ho2[3], ho[40], ho2[40]);
  return 0;
}
