// Which ingredient of the decode GEMV costs the ~2.5 us of extra fixed time over a plain streaming read?
// Variants morph a streaming kernel toward the GEMV: row-per-wave mapping, in-flight depth, LDS prologue, FMA, reduce+store.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// V: 0 grid-stride 4 in flight | 1 row-per-wave, DEPTH loads in flight, no compute | 2 +LDS prologue | 3 +FMA | 4 +reduce/store
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f4v;
// fragment-shaped weight loads: lane l reads W[row0 + (l & 15)][k0 + 8 * (l >> 4) .. +8]  (16 rows x 64 B per instruction)
template <int DEPTH>
__global__ __launch_bounds__(256, 2) void kfrag(const u4* __restrict__ W, const u4* __restrict__ x, float* __restrict__ out, int N, int K16) {
  __shared__ u4 xs[512];  // one activation row (bandwidth experiment: every batch column reads the same x)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < K16; c += 256) xs[c] = x[c];
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  const int nsteps = K16 / 4;  // 32-wide k steps
  for (int unit = blockIdx.x * 4 + wave; unit * 16 < N; unit += gridDim.x * 4) {
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    const u4* p = W + (size_t)(unit * 16 + r) * K16 + g;
    for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
      u4 w[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) w[j] = __builtin_nontemporal_load(p + (s0 + j) * 4);
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        const u4 xv = xs[(s0 + j) * 4 + g];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[j]), __builtin_bit_cast(bf16x8, xv), acc, 0, 0, 0);
      }
    }
    if (r == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) out[unit * 16 + g * 4 + q] = acc[q];
    }
  }
}


// fp8 weights, fragment-shaped: lane l reads 16 bytes W8[row0 + (l & 15)][64 s + 16 (l >> 4) .. +16] (16 rows x 64 B per instruction),
// widened in registers to the bf16 fragments of two k steps -- no LDS on the weight side
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
template <int DEPTH>
__global__ __launch_bounds__(256, 2) void kfrag8(const u4* __restrict__ W, const u4* __restrict__ x, float* __restrict__ out, int N, int K16) {
  __shared__ u4 xs[512];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < 512; c += 256) xs[c] = x[c];
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  const int nsteps = K16 / 4;  // 64-byte steps per row
  for (int unit = blockIdx.x * 4 + wave; unit * 16 < N; unit += gridDim.x * 4) {
    f4v acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    const u4* p = W + (size_t)(unit * 16 + r) * K16 + g;
    for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
      u4 w[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) w[j] = __builtin_nontemporal_load(p + (s0 + j) * 4);
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        const u4 xv = xs[((s0 + j) * 8 + g) & 511], xv2 = xs[((s0 + j) * 8 + 4 + g) & 511];
        u4 lo, hi;
        lo[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][0], 1.0f, false));
        lo[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][0], 1.0f, true));
        lo[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][1], 1.0f, false));
        lo[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][1], 1.0f, true));
        hi[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][2], 1.0f, false));
        hi[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][2], 1.0f, true));
        hi[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][3], 1.0f, false));
        hi[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[j][3], 1.0f, true));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo), __builtin_bit_cast(bf16x8, xv), acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), __builtin_bit_cast(bf16x8, xv2), acc2, 0, 0, 0);
      }
    }
    if (r == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) out[unit * 16 + g * 4 + q] = acc[q] + acc2[q];
    }
  }
}

template <int V, int DEPTH>
__global__ __launch_bounds__(256, 2) void k(const u4* __restrict__ W, const u4* __restrict__ x, float* __restrict__ out, int N, int K16) {
  __shared__ u4 xs[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (V == 0) {
    unsigned acc = 0;
    const size_t n16 = (size_t)N * K16, stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + tid;
    for (; i + 3 * stride < n16; i += 4 * stride) {
      u4 a = __builtin_nontemporal_load(W + i), b = __builtin_nontemporal_load(W + i + stride), c = __builtin_nontemporal_load(W + i + 2 * stride), d = __builtin_nontemporal_load(W + i + 3 * stride);
      acc ^= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    if (acc == 0x12345u) out[0] = 1.f;
    return;
  }
  if (V >= 2) {
    for (int c = tid; c < K16; c += 256) xs[c] = x[c];
    __syncthreads();
  }
  const int nit = K16 / 64;  // chunk iterations per row
  float facc = 0.f;
  unsigned uacc = 0;
  for (int row = blockIdx.x * 4 + wave; row < N; row += gridDim.x * 4) {
    const u4* p = W + (size_t)row * K16;
    for (int it0 = 0; it0 < nit; it0 += DEPTH) {
      u4 r[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) r[j] = __builtin_nontemporal_load(p + (it0 + j) * 64 + lane);
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        if (V >= 3) {
          const u4 xv = xs[(it0 + j) * 64 + lane];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            facc = fmaf(__uint_as_float(r[j][q] << 16), __uint_as_float(xv[q] << 16), facc);
            facc = fmaf(__uint_as_float(r[j][q] & 0xffff0000u), __uint_as_float(xv[q] & 0xffff0000u), facc);
          }
        } else {
          uacc ^= r[j][0] ^ r[j][3];
        }
      }
    }
    if (V >= 4) {
      float s = facc;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) out[row] = s;
      facc = 0.f;
    }
  }
  if (V < 4 && (uacc == 0x12345u || facc == 1.2345f)) out[0] = 1.f;
}

__global__ void fill_random(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // two bf16 values ~ N(0, small): random sign/mantissa, exponent around 2^-6
    p[i] = (h & 0x807f807fu) | 0x3c003c00u;
  }
}


// GEMM-shaped weight streaming: a block owns 128 rows and walks K in 64-element steps; one wave-instruction = 8 rows x 128 B
// (MODE 0: row-major [N][K], what the LDS-DMA GEMMs read) or 1 KiB contiguous (MODE 1: the same bytes pre-tiled [N/128][K/64][128][64]).
template <int MODE>
__global__ __launch_bounds__(256, 2) void ktile(const u4* __restrict__ W, float* __restrict__ out, int N, int K) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = K / 64, per = nk / gridDim.y, kt0 = blockIdx.y * per;
  const char* base = reinterpret_cast<const char*>(W);
  unsigned acc = 0;
  for (int kt = kt0; kt < kt0 + per; kt += 2) {
    u4 v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        size_t off;
        if (MODE == 0) off = ((size_t)(blockIdx.x * 128 + (wave * 4 + i) * 8 + (lane >> 3)) * K + (size_t)(kt + h) * 64) * 2 + (lane & 7) * 16;
        else off = ((size_t)blockIdx.x * nk + (kt + h)) * 16384 + (size_t)(wave * 4 + i) * 1024 + lane * 16;
        v[h * 4 + i] = *reinterpret_cast<const u4*>(base + off);
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][3];
  }
  if (acc == 0x12345u) out[0] = 1.f;
}
template <int MODE>
int runtile(const char* name, hipStream_t s, std::vector<u4*>& Ws, float* out, int N, int K, int splits) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (auto W : Ws) hipLaunchKernelGGL((ktile<MODE>), dim3(N / 128, splits), dim3(256), 0, s, W, out, N, K);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double us = ms * 1e3 / Ws.size(), mb = (double)N * K * 2 / 1e6;
  printf("  %-58s %7.2f us  (%.2f TB/s)\n", name, us, mb / us);
  return 0;
}

template <int DEPTH>
int runfrag(const char* name, hipStream_t s, std::vector<u4*>& Ws, u4* x, float* out, int N, int K) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (auto W : Ws) hipLaunchKernelGGL((kfrag<DEPTH>), dim3(512), dim3(256), 0, s, W, x, out, N, K / 8);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double us = ms * 1e3 / Ws.size(), mb = (double)N * K * 2 / 1e6;
  printf("  %-44s %7.2f us  (%.2f TB/s; fixed vs 7.05 TB/s %5.2f us)\n", name, us, mb / us, us - mb / 7.05);
  return 0;
}


template <int DEPTH>
int runfrag8(const char* name, hipStream_t s, std::vector<u4*>& Ws, u4* x, float* out, int N, int K, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  // the bf16 buffers hold N x K x 2 bytes: read as fp8 matrices of 2N rows (same bytes per launch as the bf16 variants)
  for (auto W : Ws) hipLaunchKernelGGL((kfrag8<DEPTH>), dim3(grid), dim3(256), 0, s, W, x, out, 2 * N, K / 16);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double us = ms * 1e3 / Ws.size(), mb = (double)N * K * 2 / 1e6;
  printf("  %-44s %7.2f us  (%.2f TB/s; fixed vs 7.05 TB/s %5.2f us)\n", name, us, mb / us, us - mb / 7.05);
  return 0;
}

template <int V, int DEPTH>
int run(const char* name, hipStream_t s, std::vector<u4*>& Ws, u4* x, float* out, int N, int K) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (auto W : Ws) hipLaunchKernelGGL((k<V, DEPTH>), dim3(512), dim3(256), 0, s, W, x, out, N, K / 8);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double us = ms * 1e3 / Ws.size(), mb = (double)N * K * 2 / 1e6;
  printf("  %-44s %7.2f us  (%.2f TB/s; fixed vs 7.05 TB/s %5.2f us)\n", name, us, mb / us, us - mb / 7.05);
  return 0;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int N = cfg == 0 ? 4096 : 14336, K = 4096, L = 24;
    std::vector<u4*> Ws(L);
    for (auto& W : Ws) { CK(hipMalloc(&W, (size_t)N * K * 2)); hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, s, (unsigned*)W, (size_t)N * K / 2, (unsigned)(size_t)W); }
    CK(hipStreamSynchronize(s));
    u4* x; CK(hipMalloc(&x, K * 2)); CK(hipMemset(x, 0x3c, K * 2));
    float* out; CK(hipMalloc(&out, N * 4));
    printf("N=%d K=%d (%.1f MB per launch)\n", N, K, (double)N * K * 2 / 1e6);
    run<0, 4>("V0 grid-stride, 4 in flight", s, Ws, x, out, N, K);
    run<1, 4>("V1 row/wave, 4 in flight, no compute", s, Ws, x, out, N, K);
    run<1, 8>("V1 row/wave, 8 in flight", s, Ws, x, out, N, K);
    run<2, 8>("V2 + LDS x prologue (barrier)", s, Ws, x, out, N, K);
    run<3, 8>("V3 + bf16 FMA on the data", s, Ws, x, out, N, K);
    run<4, 8>("V4 + wave reduce + store", s, Ws, x, out, N, K);
    run<4, 4>("V4 depth 4", s, Ws, x, out, N, K);
    runfrag<8>("MFMA 16x16x32, fragment-shaped loads, depth 8", s, Ws, x, out, N, K);
    runfrag<16>("MFMA 16x16x32, fragment-shaped loads, depth 16", s, Ws, x, out, N, K);
    runtile<0>("GEMM-shaped, row-major: 8 rows x 128 B per instr, 8 K splits", s, Ws, out, N, K, 8);
    runtile<1>("GEMM-shaped, pre-tiled: 1 KiB contiguous per instr, 8 K splits", s, Ws, out, N, K, 8);
    runtile<0>("GEMM-shaped, row-major, 2 K splits", s, Ws, out, N, K, 2);
    runtile<1>("GEMM-shaped, pre-tiled, 2 K splits", s, Ws, out, N, K, 2);
    float* out2; CK(hipMalloc(&out2, 2 * N * 4));
    runfrag8<4>("fp8 fragment loads (16 B/lane), depth 4, 512 blocks", s, Ws, x, out2, N, K, 512);
    runfrag8<8>("fp8 fragment loads, depth 8, 512 blocks", s, Ws, x, out2, N, K, 512);
    runfrag8<8>("fp8 fragment loads, depth 8, 768 blocks", s, Ws, x, out2, N, K, 768);
    runfrag8<16>("fp8 fragment loads, depth 16, 512 blocks", s, Ws, x, out2, N, K, 512);
    for (auto W : Ws) CK(hipFree(W));
  }
  return 0;
}
