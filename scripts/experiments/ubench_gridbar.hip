// What does a grid-wide barrier cost on MI355X (8 XCDs, non-coherent L2s), and does a persistent multi-phase GEMV with
// weight prefetch across the barrier beat one launch per phase?  Every spin is bounded: a stuck barrier sets err and falls through.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// FENCE: 1 = agent-scope release/acquire on the counter (L2 writeback + invalidate); 0 = relaxed counter, data moved with sc1 atomics
template <int FENCE>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target, int* err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (FENCE) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else { __builtin_amdgcn_s_waitcnt(0); __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    int spins = 0;
    while (true) {
      unsigned v = FENCE ? __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= target) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 4000000) { *err = 1; break; }
    }
  }
  __syncthreads();
}

// mode 0: barriers only. mode 1: every block publishes a word, after the barrier every block sums all words and checks.
template <int FENCE>
__global__ __launch_bounds__(256, 2) void kbar(unsigned* ctr, unsigned* data, int* err, int iters, int mode) {
  const int nb = gridDim.x;
  for (int it = 0; it < iters; ++it) {
    if (mode == 1) {
      unsigned* slot = data + (it & 1) * nb;
      if (threadIdx.x == 0) {
        if (FENCE) slot[blockIdx.x] = it * 7 + blockIdx.x;
        else __hip_atomic_store(slot + blockIdx.x, (unsigned)(it * 7 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    grid_barrier<FENCE>(ctr, (unsigned)(it + 1) * nb, err);
    if (mode == 1) {
      const unsigned* slot = data + (it & 1) * nb;
      unsigned s = 0;
      for (int i = threadIdx.x; i < nb; i += 256)
        s += FENCE ? slot[i] : __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
      __shared__ unsigned ws[4];
      if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
      __syncthreads();
      const unsigned tot = ws[0] + ws[1] + ws[2] + ws[3];
      const unsigned want = (unsigned)nb * (it * 7) + (unsigned)nb * (nb - 1) / 2;
      if (tot != want && threadIdx.x == 0) *err = 2;
      __syncthreads();
    }
  }
}

// Persistent multi-phase streaming "GEMV": phase p has rows[p] rows of K16[p] 16-byte chunks; wave per row, DEPTH loads in flight;
// the activation vector of phase p+1 is what phase p wrote (published with sc1 stores, read after the barrier).
// PREFETCH: issue the first DEPTH loads of the next phase's first row before waiting on the barrier.
struct Phases { const u4* W[4]; int rows[4]; int k16[4]; };
template <int DEPTH, int PREFETCH>
__global__ __launch_bounds__(256, 2) void kpersist(Phases ph, unsigned* ctr, float* act, int* err, int layers, size_t lstride) {
  __shared__ u4 xs[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bar = 0;
  u4 pre[DEPTH];
  bool have_pre = false;
  for (int l = 0; l < layers; ++l) {
    for (int p = 0; p < 4; ++p) {
      const int K16 = ph.k16[p], rows = ph.rows[p];
      // activation prologue (8 KB..28 KB read by every block, like the RMSNorm prologue)
      const u4* xin = (const u4*)(act + ((l * 4 + p) & 1) * 16384);
      for (int c = tid; c < K16; c += 256) {
        u4 v;
        v[0] = __hip_atomic_load((const unsigned*)(xin + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[1] = v[2] = v[3] = v[0];
        xs[c] = v;
      }
      __syncthreads();
      float* xout = act + ((l * 4 + p + 1) & 1) * 16384;
      const int nit = K16 / 64;
      bool first = true;
      for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const u4* pw = ph.W[p] + l * lstride + (size_t)row * K16 + lane;
        float acc = 0.f;
        for (int i0 = 0; i0 < nit; i0 += DEPTH) {
          u4 w[DEPTH];
          if (PREFETCH && first && have_pre && i0 == 0) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) w[j] = pre[j];
          } else {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) w[j] = __builtin_nontemporal_load(pw + (size_t)((i0 + j < nit) ? (i0 + j) : 0) * 64);
          }
#pragma unroll
          for (int j = 0; j < DEPTH; ++j) {
            if (i0 + j < nit) {
              const u4 xv = xs[(i0 + j) * 64 + lane];
              acc += __uint_as_float(w[j][0] & xv[0]) + __uint_as_float(w[j][1] ^ xv[1]) + __uint_as_float(w[j][2] | xv[2]) + __uint_as_float(w[j][3] + xv[3]);
            }
          }
        }
        first = false;
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0 && row < 16384) __hip_atomic_store(xout + row, acc * 1e-30f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // prefetch the next phase's first row for this wave
      have_pre = false;
      if (PREFETCH) {
        const int np = (p + 1) & 3;
        const int nrow = blockIdx.x * 4 + wave;
        if (nrow < ph.rows[np] && (l + 1 < layers || p < 3)) {
          const u4* pw = ph.W[np] + (size_t)(l + (p == 3)) * lstride + (size_t)nrow * ph.k16[np] + lane;
          const int nnit = ph.k16[np] / 64;
#pragma unroll
          for (int j = 0; j < DEPTH; ++j) pre[j] = __builtin_nontemporal_load(pw + (size_t)((j < nnit) ? j : 0) * 64);
          have_pre = true;
        }
      }
      ++bar;
      grid_barrier<0>(ctr, bar * gridDim.x, err);
    }
  }
}

// the same phases as separate launches (one kernel per phase), for the comparison
template <int DEPTH>
__global__ __launch_bounds__(256, 2) void kphase(const u4* __restrict__ W, int rows, int K16, const float* xin_f, float* xout) {
  __shared__ u4 xs[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u4* xin = (const u4*)xin_f;
  for (int c = tid; c < K16; c += 256) { u4 v; v[0] = ((const unsigned*)(xin + c))[0]; v[1] = v[2] = v[3] = v[0]; xs[c] = v; }
  __syncthreads();
  const int nit = K16 / 64;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const u4* pw = W + (size_t)row * K16 + lane;
    float acc = 0.f;
    for (int i0 = 0; i0 < nit; i0 += DEPTH) {
      u4 w[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) w[j] = __builtin_nontemporal_load(pw + (size_t)((i0 + j < nit) ? (i0 + j) : 0) * 64);
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        if (i0 + j < nit) {
          const u4 xv = xs[(i0 + j) * 64 + lane];
          acc += __uint_as_float(w[j][0] & xv[0]) + __uint_as_float(w[j][1] ^ xv[1]) + __uint_as_float(w[j][2] | xv[2]) + __uint_as_float(w[j][3] + xv[3]);
        }
      }
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0 && row < 16384) xout[row] = acc * 1e-30f;
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, nb = cus * 2;
  unsigned *ctr, *data;
  int* err;
  float* act;
  CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&data, 2 * nb * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&act, 2 * 16384 * 4));
  CK(hipMemset(act, 0, 2 * 16384 * 4));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int herr = 0;
  for (int fence = 0; fence < 2; ++fence)
    for (int mode = 0; mode < 2; ++mode)
      for (int blocks : {cus, nb}) {
        const int iters = 2000;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemsetAsync(ctr, 0, 64, s)); CK(hipMemsetAsync(err, 0, 4, s));
          CK(hipEventRecord(e0, s));
          if (fence) hipLaunchKernelGGL(kbar<1>, dim3(blocks), dim3(256), 0, s, ctr, data, err, iters, mode);
          else hipLaunchKernelGGL(kbar<0>, dim3(blocks), dim3(256), 0, s, ctr, data, err, iters, mode);
          CK(hipEventRecord(e1, s));
          CK(hipStreamSynchronize(s));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("barrier fence=%d exchange=%d blocks=%d: %.2f us/barrier  err=%d\n", fence, mode, blocks, best * 1000.f / iters, herr);
      }
  // multi-phase: qkv 6144x4096, o 4096x4096, gate/up 28672x4096, down 4096x14336 (bf16: K16 = K/8)
  Phases ph;
  const int rows[4] = {6144, 4096, 28672, 4096}, k16[4] = {512, 512, 512, 1792};
  const int layers = 8;
  // distinct weights per layer would need 3.5 GB; reuse would hit the 256 MB MALL -> allocate per-layer copies
  std::vector<u4*> bufs;
  size_t per_layer = 0;
  for (int p = 0; p < 4; ++p) per_layer += (size_t)rows[p] * k16[p] * 16;
  printf("bytes/layer %.1f MB\n", per_layer / 1e6);
  u4* big;
  CK(hipMalloc(&big, per_layer * layers));
  CK(hipMemset(big, 0x11, per_layer * layers));
  // persistent kernel takes one Phases (same pointers each layer) -> to defeat MALL reuse we offset by layer inside via rows trick:
  // simpler: run persistent with layers=1 per launch over different layer slices, and with layers=8 over one slice for the barrier-only effect
  for (int variant = 0; variant < 3; ++variant) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(ctr, 0, 64, s)); CK(hipMemsetAsync(err, 0, 4, s));
      CK(hipEventRecord(e0, s));
      for (int l = 0; l < (variant == 0 ? layers : 1); ++l) {
        size_t off = 0;
        for (int p = 0; p < 4; ++p) { ph.W[p] = (const u4*)((const char*)big + per_layer * l + off); ph.rows[p] = rows[p]; ph.k16[p] = k16[p]; off += (size_t)rows[p] * k16[p] * 16; }
        if (variant == 0) {
          for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(kphase<8>, dim3(nb), dim3(256), 0, s, ph.W[p], rows[p], k16[p], act + (p & 1) * 16384, act + ((p + 1) & 1) * 16384);
        } else if (variant == 1) hipLaunchKernelGGL((kpersist<8, 0>), dim3(nb), dim3(256), 0, s, ph, ctr, act, err, layers, per_layer / 16);
        else hipLaunchKernelGGL((kpersist<8, 1>), dim3(nb), dim3(256), 0, s, ph, ctr, act, err, layers, per_layer / 16);
      }
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    const char* names[3] = {"4 launches/layer", "persistent, no prefetch", "persistent, prefetch over barrier"};
    printf("%-34s %.2f us/layer  %.2f TB/s  err=%d\n", names[variant], best * 1000.f / layers, per_layer / (best / layers * 1e-3) / 1e12, herr);
  }
  return 0;
}
