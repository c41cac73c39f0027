// The product GEMV (srgpt_gemv) timed in a raw hipGraph chain, same harness as ubench_stream.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/srgpt.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void fill_random(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (h & 0x807f807fu) | 0x3c003c00u;
  }
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1;
  printf("batch %d\n", B);
  hipStream_t s; CK(hipStreamCreate(&s));
  struct Cfg { const char* name; int N, K, norm, res, swiglu; };
  Cfg cfgs[] = {{"o plain", 4096, 4096, 0, 0, 0}, {"o+res", 4096, 4096, 0, 1, 0}, {"qkv+norm", 6144, 4096, 1, 0, 0},
                {"n14336 plain", 14336, 4096, 0, 0, 0}, {"gateup+norm+swiglu", 14336, 4096, 1, 0, 1}, {"down+res", 4096, 14336, 0, 1, 0}};
  for (auto& c : cfgs) {
    const int L = c.swiglu ? 12 : 24;
    const size_t rows = (size_t)c.N * (c.swiglu ? 2 : 1);
    std::vector<void*> Ws(L);
    for (auto& W : Ws) { CK(hipMalloc(&W, rows * c.K * 2)); hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, s, (unsigned*)W, rows * c.K / 2, (unsigned)(size_t)W); }
    void *x, *g, *res, *out;
    CK(hipMalloc(&x, (size_t)B * c.K * 2)); CK(hipMalloc(&g, c.K * 2)); CK(hipMalloc(&res, (size_t)B * c.N * 2)); CK(hipMalloc(&out, (size_t)B * c.N * 4));
    hipLaunchKernelGGL(fill_random, dim3(8), dim3(256), 0, s, (unsigned*)x, (size_t)B * c.K / 2, 1u);
    hipLaunchKernelGGL(fill_random, dim3(8), dim3(256), 0, s, (unsigned*)g, (size_t)c.K / 2, 2u);
    hipLaunchKernelGGL(fill_random, dim3(8), dim3(256), 0, s, (unsigned*)res, (size_t)B * c.N / 2, 3u);
    CK(hipStreamSynchronize(s));
    for (auto W : Ws) srgpt_gemv(x, W, c.norm ? g : nullptr, 1e-5f, c.res ? res : nullptr, out, B, c.N, c.K, c.swiglu, 0, SRGPT_BF16, s);
    CK(hipStreamSynchronize(s));
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (auto W : Ws) srgpt_gemv(x, W, c.norm ? g : nullptr, 1e-5f, c.res ? res : nullptr, out, B, c.N, c.K, c.swiglu, 0, SRGPT_BF16, s);
    CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    const double us = ms * 1e3 / L, mb = (double)rows * c.K * 2 / 1e6;
    printf("%-22s %7.1f MB %7.2f us  %.2f TB/s  fixed vs 7.05: %.2f us\n", c.name, mb, us, mb / us, us - mb / 7.05);
    for (auto W : Ws) CK(hipFree(W));
  }
  return 0;
}
