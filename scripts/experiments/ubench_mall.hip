// Round 6: does the 256-MiB Infinity Cache (MALL) serve a weight stream faster than HBM, and can a second stream fill it ahead of the
// consumer?  (1) the one-row GEMV's fetch pattern over ONE buffer, replayed: cold (L distinct buffers > 256 MiB between reuses) vs warm
// (the same buffer every launch), default and non-temporal loads.  (2) a decode-step stand-in on stream A -- per "layer" four fetch-only
// launches over that layer's matrices plus one 9-us kernel that touches no memory (the attention chain) -- alone, and beside a
// PREFETCHER on stream B: a small persistent grid that reads the same bytes in the same order with default-policy loads, never more
// than WINDOW bytes ahead of the consumer's progress word.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench_mall.hip -o ubench_mall && ./ubench_mall
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT>
__global__ __launch_bounds__(256, 2) void kfetch(const u4* __restrict__ W, unsigned* __restrict__ out, int rows, int K16,
                                                 unsigned long long* progress, unsigned long long done_after) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int nit = (K16 + 63) >> 6;
  unsigned acc = 0;
  for (int row = gw; row < rows; row += nw) {
    const u4* p = W + (size_t)row * K16;
    for (int it0 = 0; it0 < nit; it0 += 8) {
      u4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u4* q = p + min((min(it0 + j, nit - 1)) * 64 + lane, K16 - 1);
        r[j] = NT ? __builtin_nontemporal_load(q) : *q;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= r[j][0] ^ r[j][3];
    }
  }
  if (acc == 0x12345u) out[0] = 1;
  // progress word: bytes of the weight sequence the consumer has been launched past (first block to get here publishes)
  if (progress && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(progress, done_after, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void kspin(long long cycles) {  // the attention chain: ~9 us, no memory traffic
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
// prefetcher: grid-stride over the whole sequence in 64-KiB units, default-policy loads, window-limited by the progress word
__global__ __launch_bounds__(256) void kprefetch(const u4* __restrict__ base, unsigned long long total_bytes, unsigned long long window,
                                                 const unsigned long long* progress, unsigned* __restrict__ out, int* stop) {
  const unsigned long long unit = 65536ull, nunits = total_bytes / unit;
  unsigned acc = 0;
  for (unsigned long long u = blockIdx.x; u < nunits; u += gridDim.x) {
    const unsigned long long off = u * unit;
    if (threadIdx.x == 0) {
      int spins = 0;
      while (off > __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + window && spins < 2000000) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    __syncthreads();
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    const u4* p = base + off / 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const u4 v = p[j * 256 + threadIdx.x]; acc ^= v[0]; }
  }
  if (acc == 0x12345u) out[0] = 1;
}
__global__ void fill(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = h;
  }
}
int main() {
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // ---- (1) warm vs cold ----
  struct Shape { const char* name; int rows, K; } shapes[] = {{"o 33.6 MB", 4096, 4096}, {"down 117 MB", 4096, 14336}, {"gate/up 235 MB", 28672, 4096}};
  for (auto& sh : shapes) {
    const size_t bytes = (size_t)sh.rows * sh.K * 2;
    const int L = (int)(1.2e9 / bytes) + 2;
    std::vector<u4*> Ws(L);
    for (auto& W : Ws) { CK(hipMalloc(&W, bytes)); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, sa, (unsigned*)W, bytes / 4, (unsigned)(size_t)W); }
    CK(hipStreamSynchronize(sa));
    for (int nt = 0; nt < 2; ++nt) for (int warm = 0; warm < 2; ++warm) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < L; ++i) {
        if (nt) hipLaunchKernelGGL(kfetch<1>, dim3(512), dim3(256), 0, sa, Ws[warm ? 0 : i], out, sh.rows, sh.K / 8, nullptr, 0ull);
        else hipLaunchKernelGGL(kfetch<0>, dim3(512), dim3(256), 0, sa, Ws[warm ? 0 : i], out, sh.rows, sh.K / 8, nullptr, 0ull);
      }
      CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      float ms = 0, best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) { CK(hipEventRecord(e0, sa)); CK(hipGraphLaunch(ge, sa)); CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms; }
      printf("%-16s %s loads, %s: %7.2f us per launch = %.2f TB/s\n", sh.name, nt ? "nt     " : "default", warm ? "same buffer every launch (MALL-warm)" : "distinct buffers (cold)           ",
             best * 1e3 / L, bytes / 1e6 / (best * 1e3 / L));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    for (auto W : Ws) CK(hipFree(W));
  }
  // ---- (2) decode-step stand-in with a prefetcher on a second stream ----
  const int LAYERS = 32;
  struct M { int rows, K; } mats[4] = {{6144, 4096}, {4096, 4096}, {28672, 4096}, {4096, 14336}};
  size_t layer_bytes = 0; for (auto& m : mats) layer_bytes += (size_t)m.rows * m.K * 2;
  const size_t total = layer_bytes * LAYERS;
  u4* all; CK(hipMalloc(&all, total));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, sa, (unsigned*)all, total / 4, 7u);
  unsigned long long* progress; CK(hipMalloc(&progress, 8)); int* stop; CK(hipMalloc(&stop, 4));
  CK(hipStreamSynchronize(sa));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const long long spin = (long long)(9e-6 * 100e6);  // wall_clock64 ticks at 100 MHz
  for (int nt = 0; nt < 2; ++nt) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    size_t off = 0;
    for (int l = 0; l < LAYERS; ++l)
      for (int k = 0; k < 4; ++k) {
        const size_t b = (size_t)mats[k].rows * mats[k].K * 2;
        if (k == 1) hipLaunchKernelGGL(kspin, dim3(64), dim3(64), 0, sa, spin);  // attention sits in front of o_proj
        if (nt) hipLaunchKernelGGL(kfetch<1>, dim3(512), dim3(256), 0, sa, all + off / 16, out, mats[k].rows, mats[k].K / 8, progress, (unsigned long long)(off + b));
        else hipLaunchKernelGGL(kfetch<0>, dim3(512), dim3(256), 0, sa, all + off / 16, out, mats[k].rows, mats[k].K / 8, progress, (unsigned long long)(off + b));
        off += b;
      }
    CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    auto run = [&](const char* name, int pf_blocks, unsigned long long window) -> int {
      float best = 1e9f, ms = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(progress, 0, 8, sa)); CK(hipMemsetAsync(stop, 0, 4, sa)); CK(hipStreamSynchronize(sa));
        if (pf_blocks) hipLaunchKernelGGL(kprefetch, dim3(pf_blocks), dim3(256), 0, sb, all, (unsigned long long)total, window, progress, out, stop);
        CK(hipEventRecord(e0, sa)); CK(hipGraphLaunch(ge, sa)); CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        int one = 1; CK(hipMemcpyAsync(stop, &one, 4, hipMemcpyHostToDevice, sa)); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      }
      printf("step stand-in (%s consumer loads) %-52s %7.3f ms per step = %.2f TB/s\n", nt ? "nt" : "default", name, best, total / 1e9 / best);
      return 0;
    };
    if (run("alone", 0, 0)) return 1;
    if (run("+ prefetcher 32 blocks, window 64 MB", 32, 64ull << 20)) return 1;
    if (run("+ prefetcher 64 blocks, window 128 MB", 64, 128ull << 20)) return 1;
    if (run("+ prefetcher 128 blocks, window 128 MB", 128, 128ull << 20)) return 1;
    if (run("+ prefetcher 256 blocks, window 192 MB", 256, 192ull << 20)) return 1;
    if (run("+ prefetcher 64 blocks, window 32 MB", 64, 32ull << 20)) return 1;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
