// Feasibility probe for DESIGN section 9.1: can consecutive weight-streaming GEMVs of a batch-1 decode layer overlap their
// ramp / drain when they are launched on two alternating streams and ordered by an in-kernel arrival counter instead of a
// kernel boundary?  Consumer blocks request their first weight batch, THEN wait for the producer's counter, then read the
// activation vector with agent-scope loads.  Spins are bounded (an error word is set, nothing hangs).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/experiments/ubench_overlap.hip -o scripts/ubench_overlap
//   run:   scripts/ubench_overlap            -> us per "layer" (qkv 6144x4096, o 4096x4096, gate/up 28672x4096, down 4096x14336)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));            \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// out[n] = sum_k W[n,k] x[k]   (bf16 in, bf16 out); one wave per row, 8 x 1 KiB loads per batch; x staged in LDS.
// SYNC: wait for *wait_ctr >= wait_target before reading x; publish out write-through and bump *sig_ctr.
template <bool SYNC>
__global__ __launch_bounds__(256, 2) void gemv(const unsigned short* __restrict__ W, const unsigned int* x,
                                               unsigned int* out, int N, int K, int* wait_ctr, int wait_target,
                                               int* sig_ctr, int* err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* xs = reinterpret_cast<unsigned short*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = K / 8, nit = (nchunks + 63) / 64;
  // ---- first weight batch of this wave's first row: requested before anything that depends on the producer ----
  const int unit0 = blockIdx.x * 4 + wave;
  u32x4 w0[8];
  {
    const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)min(unit0, N - 1) * K);
#pragma unroll
    for (int j = 0; j < 8; ++j) w0[j] = __builtin_nontemporal_load(p + min(j * 64 + lane, nchunks - 1));
  }
  if (SYNC && wait_ctr) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_target) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > 20000) {  // ~ a few ms: give up, flag, never hang
          atomicExch(err, 1);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ONE acquire after the match; plain loads below
    }
    __syncthreads();
  }
  // ---- x -> LDS (agent-scope 8-byte loads when it was published inside a concurrently running launch) ----
  // activations travel as one 32-bit word per element (bf16 in the low half): the smallest agent-scope store is 4 bytes
  for (int c = tid; c < K / 2; c += 256) {
    const unsigned long long v = reinterpret_cast<const unsigned long long*>(x)[c];
    reinterpret_cast<unsigned int*>(xs)[c] = ((unsigned)v & 0xffffu) | ((unsigned)(v >> 32) << 16);
  }
  __syncthreads();
  auto consume = [&](const u32x4 (&w)[8], int it0, float& acc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = (it0 + j) * 64 + lane;
      const bool valid = ch < nchunks;
      const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + (size_t)(valid ? ch : 0) * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned wq = valid ? w[j][q] : 0u;
        acc = fmaf(bf_lo(wq), bf_lo(xv[q]), acc);
        acc = fmaf(bf_hi(wq), bf_hi(xv[q]), acc);
      }
    }
  };
  auto finish = [&](int unit, float acc) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0 && unit < N) {
      const unsigned int r = __float_as_uint(acc) >> 16;
      if (SYNC)
        __hip_atomic_store(out + unit, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        out[unit] = r;
    }
  };
  // first row: batch 0 is already in flight
  {
    float acc = 0.f;
    consume(w0, 0, acc);
    const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)min(unit0, N - 1) * K);
    for (int it0 = 8; it0 < nit; it0 += 8) {
      u32x4 w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = __builtin_nontemporal_load(p + min((it0 + j) * 64 + lane, nchunks - 1));
      consume(w, it0, acc);
    }
    finish(unit0, acc);
  }
  for (int unit = unit0 + gridDim.x * 4; unit < N; unit += gridDim.x * 4) {
    float acc = 0.f;
    const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)unit * K);
    for (int it0 = 0; it0 < nit; it0 += 8) {
      u32x4 w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = __builtin_nontemporal_load(p + min((it0 + j) * 64 + lane, nchunks - 1));
      consume(w, it0, acc);
    }
    finish(unit, acc);
  }
  if (SYNC && sig_ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(sig_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct Op {
  int N, K;
};

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 16;
  const Op ops[4] = {{6144, 4096}, {4096, 4096}, {14336, 4096}, {4096, 14336}};  // gate/up modelled as 14336 x 4096 x 2 below
  const int grid = 512;
  // weights: distinct per layer so that nothing stays in the 256 MB Infinity Cache
  std::vector<unsigned short*> W(layers * 4);
  for (int l = 0; l < layers; ++l)
    for (int o = 0; o < 4; ++o) {
      const size_t n = (size_t)ops[o].N * ops[o].K * (o == 2 ? 2 : 1);
      CHECK(hipMalloc(&W[l * 4 + o], n * 2));
      CHECK(hipMemset(W[l * 4 + o], 0x3c, n * 2));
    }
  unsigned int *xa, *xb;
  CHECK(hipMalloc(&xa, 65536 * 4));
  CHECK(hipMalloc(&xb, 65536 * 4));
  CHECK(hipMemset(xa, 0, 65536 * 4));
  CHECK(hipMemset(xb, 0, 65536 * 4));
  int *ctr, *err;
  CHECK(hipMalloc(&ctr, layers * 4 * sizeof(int) + 4));
  CHECK(hipMalloc(&err, 4));
  CHECK(hipMemset(err, 0, 4));
  CHECK(hipFuncSetAttribute((const void*)gemv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CHECK(hipFuncSetAttribute((const void*)gemv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t efork, ejoin, t0, t1;
  CHECK(hipEventCreate(&efork));
  CHECK(hipEventCreate(&ejoin));
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));

  auto enqueue = [&](bool overlap) {
    // one "token": layers x {qkv, o, gate/up (2N rows), down}; x ping-pongs between two buffers
    if (overlap) CHECK(hipMemsetAsync(ctr, 0, layers * 4 * sizeof(int) + 4, sa));
    if (overlap) {
      CHECK(hipEventRecord(efork, sa));
      CHECK(hipStreamWaitEvent(sb, efork, 0));
    }
    int k = 0;
    for (int l = 0; l < layers; ++l)
      for (int o = 0; o < 4; ++o, ++k) {
        const int N = ops[o].N * (o == 2 ? 2 : 1), K = ops[o].K;
        unsigned int* xin = (k & 1) ? xb : xa;
        unsigned int* xout = (k & 1) ? xa : xb;
        hipStream_t s = (overlap && (k & 1)) ? sb : sa;
        if (overlap)
          hipLaunchKernelGGL(gemv<true>, dim3(grid), dim3(256), (size_t)K * 2, s, W[l * 4 + o], xin, xout, N, K,
                             k ? ctr + (k - 1) : nullptr, grid, ctr + k, err);
        else
          hipLaunchKernelGGL(gemv<false>, dim3(grid), dim3(256), (size_t)K * 2, s, W[l * 4 + o], xin, xout, N, K, nullptr, 0,
                             nullptr, err);
      }
    if (overlap) {
      CHECK(hipEventRecord(ejoin, sb));
      CHECK(hipStreamWaitEvent(sa, ejoin, 0));
    }
  };
  for (int mode = 0; mode < 4; ++mode) {
    const bool overlap = mode & 1, graph = mode < 2;
    CHECK(hipMemset(err, 0, 4));
    hipGraph_t g;
    hipGraphExec_t ge = nullptr;
    if (graph) {
      CHECK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
      enqueue(overlap);
      CHECK(hipStreamEndCapture(sa, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    auto run = [&]() {
      if (graph)
        CHECK(hipGraphLaunch(ge, sa));
      else
        enqueue(overlap);
    };
    for (int i = 0; i < 3; ++i) run();
    CHECK(hipStreamSynchronize(sa));
    CHECK(hipStreamSynchronize(sb));
    CHECK(hipEventRecord(t0, sa));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) run();
    CHECK(hipEventRecord(t1, sa));
    CHECK(hipStreamSynchronize(sa));
    CHECK(hipStreamSynchronize(sb));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, t0, t1));
    int herr = 0;
    CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-7s %s: %.2f us per layer (%d layers, %d replays)%s\n", graph ? "graph" : "eager",
           overlap ? "two streams + in-kernel counters" : "one stream, kernel boundaries   ", ms * 1e3 / reps / layers, layers, reps,
           herr ? "   [SPIN TIMEOUT FLAGGED]" : "");
  }
  return 0;
}
