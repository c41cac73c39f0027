// Per-wave timeline of the decode GEMV (build: hipcc -DSRGPT_TUNING_KNOBS -DSRGPT_GEMV_TS ubench_gemv_ts.hip ../spatialrgpt_amd/csrc/gemv.hip ../spatialrgpt_amd/csrc/misc.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../include/srgpt.h"
extern "C" void* srgpt_gemv_ts_ptr();
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096;
  const int L = 8;
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<void*> W(L);
  for (int i = 0; i < L; ++i) { CK(hipMalloc(&W[i], (size_t)N * K * 2)); CK(hipMemset(W[i], 0x11, (size_t)N * K * 2)); }
  void *x, *out; CK(hipMalloc(&x, K * 2)); CK(hipMemset(x, 0x11, K * 2)); CK(hipMalloc(&out, N * 2));
  // touch a big buffer between launches so weights are cold in L2/L3
  for (int i = 0; i < L; ++i) { int rc = srgpt_gemv(x, W[i], nullptr, 0.f, nullptr, out, 1, N, K, 0, 0, SRGPT_BF16, s); if (rc) { printf("rc %d %s\n", rc, srgpt_last_error()); return 1; } }
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  srgpt_gemv(x, W[0], nullptr, 0.f, nullptr, out, 1, N, K, 0, 0, SRGPT_BF16, s);
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> ts(8 * 8192);
  CK(hipMemcpy(ts.data(), srgpt_gemv_ts_ptr(), ts.size() * 8, hipMemcpyDeviceToHost));
  const int waves = 2048;
  long long t0 = ts[0];
  for (int w = 0; w < waves; ++w) t0 = std::min(t0, ts[w * 8]);
  const char* names[5] = {"entry", "loads issued", "prologue done", "first batch consumed", "exit"};
  printf("N=%d K=%d  event time %.2f us (100 MHz wall clock => 10 ns ticks)\n", N, K, ms * 1e3);
  for (int sl = 0; sl < 5; ++sl) {
    std::vector<double> v;
    for (int w = 0; w < waves; ++w) v.push_back((ts[w * 8 + sl] - t0) * 0.01);
    std::sort(v.begin(), v.end());
    printf("  %-22s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", names[sl], v[0], v[waves / 2], v[waves * 9 / 10], v[waves - 1]);
  }
  return 0;
}
