// The decode-path products (srgpt_gemv / srgpt_gemv_w8) of one LLM layer + lm_head, timed in a raw hipGraph chain per shape.
//   ubench_decode_mv <libsrgpt_hip*.so> <batch> <bf16|fp8> [pub]
// pub: through srgpt_gemv_rowss as the batched decode step calls them (the RMSNorm products read a published row-statistics table,
// the residual products publish one)
// The library is dlopen'ed so that one binary times every build variant (scripts/build_skinny_variants.sh).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int (*gemv_fn)(const void*, const void*, const void*, float, const void*, void*, int, int, int, int, int, int, void*);
typedef int (*gemv_w8_fn)(const void*, const void*, const float*, const void*, float, const void*, void*, int, int, int, int, int, void*);
typedef int (*gemv_rowss_fn)(const void*, const void*, const void*, const float*, const void*, float, const void*, void*, int, int, int, int, int,
                             const float*, float*, int, void*);
__global__ void fill_bf16(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (h & 0x807f807fu) | 0x3c003c00u;
  }
}
__global__ void fill_fp8(unsigned* p, size_t n, unsigned seed) {  // random e4m3 codes, exponent bit 0 cleared: never NaN
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = h & 0xf7f7f7f7u;
  }
}
__global__ void fill_f32(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: %s lib.so batch bf16|fp8\n", argv[0]); return 2; }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { printf("dlopen: %s\n", dlerror()); return 2; }
  gemv_fn gemv = (gemv_fn)dlsym(lib, "srgpt_gemv");
  gemv_w8_fn gemv_w8 = (gemv_w8_fn)dlsym(lib, "srgpt_gemv_w8");
  const char* (*last_error)() = (const char* (*)())dlsym(lib, "srgpt_last_error");
  gemv_rowss_fn gemv_rowss = (gemv_rowss_fn)dlsym(lib, "srgpt_gemv_rowss");
  if (!gemv || !gemv_w8) { printf("symbols missing\n"); return 2; }
  const int B = atoi(argv[2]);
  const bool fp8 = !strcmp(argv[3], "fp8");
  const bool pub = argc > 4 && !strcmp(argv[4], "pub");
  if (pub && !gemv_rowss) { printf("srgpt_gemv_rowss missing\n"); return 2; }
  float *ss_a, *ss_b;  // row-statistics tables [B][512]
  CK(hipMalloc(&ss_a, (size_t)B * 512 * 4)); CK(hipMalloc(&ss_b, (size_t)B * 512 * 4));
  CK(hipMemset(ss_a, 0, (size_t)B * 512 * 4)); CK(hipMemset(ss_b, 0, (size_t)B * 512 * 4));
  const int web = fp8 ? 1 : 2;
  hipStream_t s; CK(hipStreamCreate(&s));
  struct Cfg { const char* name; int N, K, norm, res, swiglu, f32; int L; };
  Cfg cfgs[] = {{"qkv+norm", 6144, 4096, 1, 0, 0, 0, 24}, {"o+res", 4096, 4096, 0, 1, 0, 0, 24}, {"gateup+norm+swiglu", 14336, 4096, 1, 0, 1, 0, 12},
                {"down+res", 4096, 14336, 0, 1, 0, 0, 24}, {"lm_head+norm f32", 128258, 4096, 1, 0, 0, 1, 3}};
  double layer_us = 0;
  for (auto& c : cfgs) {
    const int L = c.L;
    const size_t rows = (size_t)c.N * (c.swiglu ? 2 : 1);
    std::vector<void*> Ws(L);
    for (auto& W : Ws) {
      CK(hipMalloc(&W, (rows + 32) * c.K * web));  // slack: the packed layout pads the rows to whole 16-row tiles
      if (fp8) hipLaunchKernelGGL(fill_fp8, dim3(2048), dim3(256), 0, s, (unsigned*)W, rows * c.K / 4, (unsigned)(size_t)W);
      else hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, s, (unsigned*)W, rows * c.K / 2, (unsigned)(size_t)W);
    }
    void *x, *g, *res, *out; float* sc;
    CK(hipMalloc(&x, (size_t)B * c.K * 2)); CK(hipMalloc(&g, c.K * 2)); CK(hipMalloc(&res, (size_t)B * c.N * 2)); CK(hipMalloc(&out, (size_t)B * c.N * 4));
    CK(hipMalloc(&sc, rows * 4));
    hipLaunchKernelGGL(fill_bf16, dim3(8), dim3(256), 0, s, (unsigned*)x, (size_t)B * c.K / 2, 1u);
    hipLaunchKernelGGL(fill_bf16, dim3(8), dim3(256), 0, s, (unsigned*)g, (size_t)c.K / 2, 2u);
    hipLaunchKernelGGL(fill_bf16, dim3(8), dim3(256), 0, s, (unsigned*)res, (size_t)B * c.N / 2, 3u);
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, s, sc, rows, 0.001f);
    CK(hipStreamSynchronize(s));
    auto run = [&](void* W) -> int {
      if (pub && c.N != 128258)
        return gemv_rowss(x, fp8 ? nullptr : W, fp8 ? W : nullptr, fp8 ? sc : nullptr, c.norm ? g : nullptr, 1e-5f, c.res ? res : nullptr, out, B,
                          c.N, c.K, c.swiglu, c.f32, c.norm ? ss_a : nullptr, c.res ? ss_b : nullptr, 0, s);
      if (fp8) return gemv_w8(x, W, sc, c.norm ? g : nullptr, 1e-5f, c.res ? res : nullptr, out, B, c.N, c.K, c.swiglu, c.f32, s);
      return gemv(x, W, c.norm ? g : nullptr, 1e-5f, c.res ? res : nullptr, out, B, c.N, c.K, c.swiglu, c.f32, 1 /*SRGPT_BF16*/, s);
    };
    for (auto W : Ws) { int rc = run(W); if (rc) { printf("rc %d: %s\n", rc, last_error ? last_error() : "?"); return 1; } }
    CK(hipStreamSynchronize(s));
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (auto W : Ws) run(W);
    CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms; }
    const double us = best * 1e3 / L, mb = (double)rows * c.K * web / 1e6;
    printf("  %-20s %7.1f MB %7.2f us  %.2f TB/s\n", c.name, mb, us, mb / us);
    if (c.N != 128258) layer_us += us;
    for (auto W : Ws) CK(hipFree(W));
    CK(hipFree(x)); CK(hipFree(g)); CK(hipFree(res)); CK(hipFree(out)); CK(hipFree(sc));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
  }
  printf("  sum of the four layer products: %.2f us\n", layer_us);
  return 0;
}
