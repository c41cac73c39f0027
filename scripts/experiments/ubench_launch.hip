// Microbenchmark: cost of a dependent kernel boundary on MI355X (eager vs hipGraph), trivial kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_empty(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p == nullptr) p[0] = 1; }
__global__ void k_touch(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
typedef __attribute__((ext_vector_type(4))) float f4;
__global__ void k_big(const f4* __restrict__ in, float* out, size_t n) {  // streaming read
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    f4 v = __builtin_nontemporal_load(in + i);
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1.2345f) out[0] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  int* d; CK(hipMalloc(&d, 4096 * 4)); CK(hipMemset(d, 0, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  for (int grid : {1, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, d);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("eager  chain grid=%4d: %.2f us/kernel\n", grid, ms * 1e3 / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, d);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("graph  chain grid=%4d: %.2f us/kernel\n", grid, ms * 1e3 / 2000);
    }
  }
  // streaming kernels of various sizes inside a graph chain: t = a + bytes/BW
  const size_t TOT = (size_t)6 << 30;
  f4* big; CK(hipMalloc(&big, TOT)); CK(hipMemset(big, 1, TOT));
  float* o; CK(hipMalloc(&o, 64));
  for (size_t mb : {8, 32, 64, 128, 256, 1024}) {
    size_t bytes = mb << 20, n = bytes / 16;
    int cnt = (int)(TOT / bytes); if (cnt > 64) cnt = 64;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < cnt; ++i) hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, s, big + (size_t)i * n, o, n);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("stream %5zu MB x%d: %.2f us/kernel  -> %.2f TB/s\n", mb, cnt, ms * 1e3 / cnt, bytes / (ms * 1e-3 / cnt) / 1e12);
    }
  }
  return 0;
}
