// How fast can ONE workgroup pull cold bytes?  `nblocks` workgroups of `nthreads` threads each read `kb` KiB of their own (cold)
// region with every 16-byte load issued up front (NLD loads per thread in flight), reduce, and store one value.  Timed as a
// hipGraph chain of launches over distinct buffers.   ubench_cu_pull
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int NLD>
__global__ void pull(const u32x4* __restrict__ src, unsigned* __restrict__ dst, size_t per_block_vec) {
  const u32x4* p = src + (size_t)blockIdx.x * per_block_vec;
  u32x4 v[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) v[i] = __builtin_nontemporal_load(p + (size_t)i * blockDim.x + threadIdx.x);
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) s += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
  if (s == 0x12345678u) dst[blockIdx.x] = s;
}
template <int NLD>
int run(int nblocks, int nthreads, hipStream_t s) {
  const size_t per_block_vec = (size_t)NLD * nthreads;  // 16-byte vectors per block
  const int L = 40;
  const size_t vec_per_launch = per_block_vec * nblocks;
  // buffers: enough distinct data that nothing is cached: 40 launches x bytes; cap total at 1.5 GB by re-striding inside a big arena
  const size_t arena_vec = (size_t)3 << 26;  // 3 GiB / 16
  u32x4* arena; unsigned* dst;
  CK(hipMalloc(&arena, arena_vec * 16)); CK(hipMalloc(&dst, 4096 * 4));
  CK(hipMemsetAsync(arena, 1, arena_vec * 16, s));
  CK(hipStreamSynchronize(s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < L; ++l) {
    // spread the launches over the arena (64 MiB apart: a different region every time, far beyond L2; MALL holds 256 MiB)
    const u32x4* src = arena + ((size_t)l * (arena_vec / L));
    hipLaunchKernelGGL(pull<NLD>, dim3(nblocks), dim3(nthreads), 0, s, src, dst, per_block_vec);
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, ms;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(arena, rep + 2, arena_vec * 16, s));  // evict / refresh: the reads below are cold again
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double us = best * 1e3 / L, kb = per_block_vec * 16 / 1024.0;
  printf("blocks %4d x %4d thr, %6.1f KiB per block (%2d loads/thread): %6.2f us per launch  -> %6.1f GB/s per block, %5.2f TB/s total\n",
         nblocks, nthreads, kb, NLD, us, kb * 1024 / (us * 1e3), kb * 1024 * nblocks / (us * 1e6));
  CK(hipFree(arena)); CK(hipFree(dst));
  return 0;
}
__global__ void empty_kernel(unsigned* d) { if (d == nullptr) d[0] = 1; }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int nb : {8, 32, 64}) {
    run<4>(nb, 1024, s); run<8>(nb, 1024, s); run<12>(nb, 1024, s); run<16>(nb, 1024, s); run<24>(nb, 1024, s);
  }
  run<12>(8, 512, s); run<24>(8, 512, s);
  run<3>(128, 256, s); run<6>(128, 256, s);   // the split geometry: 128 blocks x 12-24 KiB
  run<3>(512, 256, s); run<6>(512, 256, s);
  return 0;
}
