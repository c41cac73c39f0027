// Shared device/host helpers for the srgpt HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/srgpt.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// error reporting across the C ABI (no exceptions, thread-local message)
// ------------------------------------------------------------------------------------------------
void srgpt_set_error(const char* fmt, ...);

#define SRGPT_CHECK(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      srgpt_set_error(__VA_ARGS__);       \
      return (code);                      \
    }                                     \
  } while (0)

#define SRGPT_LAUNCH_CHECK()                                                       \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      srgpt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return SRGPT_ERR_LAUNCH;                                                     \
    }                                                                              \
  } while (0)

// a HIP runtime call whose failure must surface through the C ABI
#define SRGPT_HIP_TRY(call, what)                                                     \
  do {                                                                                \
    const hipError_t e__ = (call);                                                    \
    if (e__ != hipSuccess) {                                                          \
      srgpt_set_error("%s failed: %s", (what), hipGetErrorString(e__));               \
      return SRGPT_ERR_LAUNCH;                                                        \
    }                                                                                 \
  } while (0)

#define SRGPT_TRY(expr)        \
  do {                         \
    int rc__ = (expr);         \
    if (rc__ != 0) return rc__; \
  } while (0)

// Tuning knobs: the product library is built WITHOUT SRGPT_TUNING_KNOBS and every knob is its compile-time default (no
// getenv, no globals).  `make TUNING=1` builds libsrgpt_hip_tuning.so for the A/B scripts under scripts/, where the same
// sites read the environment once.
#ifdef SRGPT_TUNING_KNOBS
#include <stdlib.h>
#define SRGPT_KNOB(name, dflt) ([]() -> int { static const int v__ = getenv(name) ? atoi(getenv(name)) : (dflt); return v__; }())
#else
#define SRGPT_KNOB(name, dflt) (dflt)
#endif

// Raise a kernel's dynamic-LDS limit once per (kernel, device).  hipFuncSetAttribute is per device, so the "done" state is a
// per-device bit behind an atomic (thread-safe, idempotent): `done` is the call site's own function-local atomic.
#include <atomic>
static inline int srgpt_ensure_dyn_lds(std::atomic<uint64_t>& done, const void* kfn, int bytes) {
  if (bytes <= 48 * 1024) return 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    srgpt_set_error("hipFuncSetAttribute(dynamic LDS %d B) failed: %s", bytes, hipGetErrorString(e));
    return SRGPT_ERR_LAUNCH;
  }
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

static inline hipStream_t as_stream(srgpt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t dtype_size(int dtype) { return dtype == SRGPT_BF16 ? 2 : 4; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
template <typename T>
__device__ __forceinline__ T from_f(float x);
template <>
__device__ __forceinline__ float from_f<float>(float x) { return x; }
template <>
__device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }  // RNE (v_cvt_pk_bf16_f32)

// round-trip through the storage type: mirrors PyTorch materialising an intermediate in `T`
template <typename T>
__device__ __forceinline__ float rnd(float x) { return to_f(from_f<T>(x)); }

__device__ __forceinline__ float bf16lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }


// Cross-lane sums / maxima on the VALU (DPP row operations + v_permlane16/32_swap) instead of ds_bpermute (what __shfl_xor
// compiles to on gfx950: an LDS-crossbar round trip of ~100+ cycles per step, and reductions are dependent chains of them).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// v_permlane16_swap a, b swaps the odd 16-lane rows of a with the even rows of b; v_permlane32_swap the upper half-wave of a with
// the lower half of b.  Fed two copies of v: a = the value of the lane's even row (lower half), b = of its odd row (upper half),
// in every lane -- op(a, b) is the pairwise reduction with the same operand order, hence the same bits, in both partners.
// (Inline asm: the builtins' second result comes back as a copy of the first with hipcc 7.2; the s_nops cover the VALU-write ->
// permlane-read and permlane-write -> VALU-read hazards the assembler cannot see around an asm block.)
template <typename Op>
__device__ __forceinline__ float rows_pair(float v, Op op) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return op(a, b);
}
template <typename Op>
__device__ __forceinline__ float halves_pair(float v, Op op) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return op(a, b);
}
// reduction over aligned groups of N lanes (N = 2, 4, ..., 64); every lane of a group ends with the same bits:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror, then the row / half-wave swaps
template <int N, typename Op>
__device__ __forceinline__ float lanes_reduce(float v, Op op) {
  if constexpr (N >= 2) v = op(v, dpp_mov<0xB1>(v));
  if constexpr (N >= 4) v = op(v, dpp_mov<0x4E>(v));
  if constexpr (N >= 8) v = op(v, dpp_mov<0x141>(v));
  if constexpr (N >= 16) v = op(v, dpp_mov<0x140>(v));
  if constexpr (N >= 32) v = rows_pair(v, op);
  if constexpr (N >= 64) v = halves_pair(v, op);
  return v;
}
template <int N>
__device__ __forceinline__ float lanes_sum(float v) { return lanes_reduce<N>(v, [](float a, float b) { return a + b; }); }
template <int N>
__device__ __forceinline__ float lanes_max(float v) { return lanes_reduce<N>(v, [](float a, float b) { return fmaxf(a, b); }); }
// whole-wave reductions (every lane gets the result): 4 DPP steps + 2 swaps, ~50 cycles -- the __shfl_xor butterfly they replace is
// six dependent ds_bpermute round trips (~700 cycles), which in the GEMV sat between a row's last FMA and the next row's loads
#ifndef SRGPT_WAVE_BPERMUTE  // (A/B builds only: the butterfly)
__device__ __forceinline__ float wave_sum(float v) { return lanes_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return lanes_max<64>(v); }
#else
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
#endif
// sum over the lanes l, l + S, l + 2S, ... of the wave (S = 1, 2, ..., 32 a power of two): rotations inside the 16-lane row, then
// the row / half-wave swaps; every lane ends with the sum of its residue class mod S
template <int S>
__device__ __forceinline__ float strided_sum(float v) {
  if constexpr (S <= 8) v += dpp_mov<0x128>(v);   // row_ror:8
  if constexpr (S <= 4) v += dpp_mov<0x124>(v);   // row_ror:4
  if constexpr (S <= 2) v += dpp_mov<0x122>(v);   // row_ror:2
  if constexpr (S <= 1) v += dpp_mov<0x121>(v);   // row_ror:1
  if constexpr (S <= 16) v = rows_pair(v, [](float a, float b) { return a + b; });
  if constexpr (S <= 32) v = halves_pair(v, [](float a, float b) { return a + b; });
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// vector of VEC elements of T occupying 16 bytes
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <>
struct Vec16<bf16_t> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  return 0.5f * x * (1.f + tanhf(k0 * (x + k1 * x * x * x)));
}
__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }

template <typename T>
__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case SRGPT_ACT_GELU_ERF: return gelu_erf(x);
    case SRGPT_ACT_GELU_TANH: return gelu_tanh(x);
    case SRGPT_ACT_SILU: return silu(x);
    case SRGPT_ACT_QUICK_GELU: return x / (1.f + __expf(-1.702f * x));  // x * sigmoid(1.702 x)
    default: return x;
  }
}

// ------------------------------------------------------------------------------------------------
// L2 prefetch blocks riding on a decode-step launch.  A batch-1 decode step is a chain of weight-streaming GEMVs separated by
// kernel boundaries: every GEMV pays a full memory latency before its first FMA, and the launch in front of it leaves HBM
// idle while it drains (or, for the attention kernel, for its whole duration).  Extra blocks appended to launch N pull the
// first rows launch N+1 will read into the L2 of the XCD its blocks will run on (block j of a launch runs on XCD j % 8 -- a
// wrong guess costs speed, never correctness): prefetch block p loads what GEMV block p will read first.  Being the LAST
// blocks of the grid they are dispatched as the compute blocks retire, i.e. into the drain.
// ------------------------------------------------------------------------------------------------
struct SrgptPrefetch {
  const char* base;      // weight matrix of the next GEMV (NULL: no prefetch blocks)
  long long row_bytes;   // stride between weight rows
  int prefix_bytes;      // leading bytes of each row to pull (multiple of 1024; the whole row if >= row_bytes)
  int n_units;           // work units of the next GEMV (a wave owns units unit0 + k * gemv_grid * 4)
  int unit_rows;         // rows per unit
  int umul, rstride;     // row of (unit u, r) = u * umul + r * rstride  (plain: 1, -; SwiGLU gate/up: 1, N; fp8 column pair: 2, 1)
  int gemv_grid;         // compute blocks the next GEMV launches
  int rounds;            // how many of a wave's units to pull
  int nblocks;           // prefetch blocks appended to this launch (= gemv_grid)
  int n_rows;            // rows of the matrix: the row index is clamped (fp8 column pairs of an odd N would touch row N)
  int batch;             // 1-KiB loads a wave keeps in flight (1: one at a time -- the trickle that leaves the host launch's own
                         // loads alone; 2 / 4 / 8: faster, at the price of queueing in front of them)
  int tile_bytes;        // > 0: TILE mode (round 6) -- the next launch is the batched MFMA product on a PACKED matrix whose block p
                         // streams the contiguous tile p (16 rows x K): prefetch block p pulls that tile, its waves interleaved
};

template <int NB>
__device__ __forceinline__ void srgpt_prefetch_rows(const SrgptPrefetch& pf, int p, int wave, int lane) {
  const int per_row = (int)((pf.prefix_bytes < pf.row_bytes ? (long long)pf.prefix_bytes : pf.row_bytes) >> 10);
  int done = 0;
  for (int u = p * 4 + wave; u < pf.n_units && done < pf.rounds; u += pf.gemv_grid * 4, ++done)
    for (int r = 0; r < pf.unit_rows; ++r) {
      const long long ri = min((long long)u * pf.umul + (long long)r * pf.rstride, (long long)pf.n_rows - 1);
      const char* row = pf.base + (size_t)ri * pf.row_bytes + lane * 16;
      for (int c0 = 0; c0 < per_row; c0 += NB) {
        u32x4 v[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) v[c] = *reinterpret_cast<const u32x4*>(row + ((size_t)min(c0 + c, per_row - 1) << 10));
#pragma unroll
        for (int c = 0; c < NB; ++c) asm volatile("" ::"v"(v[c]));  // keep the loads; the data is dropped
      }
    }
}
// tile mode: block p pulls tile p (+ k * gemv_grid), wave w the 1-KiB chunks w, w + nw, ...
template <int NB>
__device__ __forceinline__ void srgpt_prefetch_tiles(const SrgptPrefetch& pf, int p, int wave, int lane) {
  const int nw = (int)blockDim.x >> 6, chunks = pf.tile_bytes >> 10;
  int done = 0;
  for (int t = p; t < pf.n_units && done < pf.rounds; t += pf.gemv_grid, ++done) {
    const char* tb = pf.base + (size_t)t * pf.tile_bytes + lane * 16;
    for (int c0 = wave; c0 < chunks; c0 += nw * NB) {
      u32x4 v[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) v[c] = *reinterpret_cast<const u32x4*>(tb + ((size_t)min(c0 + c * nw, chunks - 1) << 10));
#pragma unroll
      for (int c = 0; c < NB; ++c) asm volatile("" ::"v"(v[c]));  // keep the loads; the data is dropped
    }
  }
}
__device__ __forceinline__ void srgpt_prefetch_block(const SrgptPrefetch& pf, int p, int wave, int lane) {
  if (pf.tile_bytes > 0) {
    switch (pf.batch) {
      case 8: srgpt_prefetch_tiles<8>(pf, p, wave, lane); break;
      case 4: srgpt_prefetch_tiles<4>(pf, p, wave, lane); break;
      case 2: srgpt_prefetch_tiles<2>(pf, p, wave, lane); break;
      default: srgpt_prefetch_tiles<1>(pf, p, wave, lane); break;
    }
    return;
  }
  switch (pf.batch) {
    case 8: srgpt_prefetch_rows<8>(pf, p, wave, lane); break;
    case 4: srgpt_prefetch_rows<4>(pf, p, wave, lane); break;
    case 2: srgpt_prefetch_rows<2>(pf, p, wave, lane); break;
    default: srgpt_prefetch_rows<1>(pf, p, wave, lane); break;
  }
}

extern "C" int srgpt_device_cus(void);
// descriptor for "the next launch is the batch-`batch` decode GEMV over W [N (2N if swiglu), K]" (bf16 rows, or fp8 bytes).
// Mirrors the grid / unit mapping of gemv.hip's and gemv_w8.hip's launchers for one row (the VALU kernels); anything else
// gets no prefetch: fp8 SwiGLU units of four rows, and 2+ rows (the skinny kernel -- its mapping, 4-row groups of block p's
// 16-row units, was measured: o_proj +1 % at 8 fp8 rows, -2 % at 4 bf16 rows per decode step, not kept; at 2 rows, which moved
// to the skinny kernel in round 3, the VALU mapping's prefetch costs 1.6 % per step: profiles/r03_skinny_min_batch.txt).
static inline SrgptPrefetch srgpt_prefetch_for_gemv(const void* W, int N, int K, int swiglu, int fp8, int batch, int rounds,
                                                    int prefix_bytes) {
  SrgptPrefetch pf{nullptr, 0, 0, 0, 1, 1, 0, 1, 0, 0, 1, 1, 0};
  const long long row_bytes = fp8 ? (long long)K : 2LL * K;
  if (!W || batch > 1 || rounds <= 0 || row_bytes % 1024 != 0 || (fp8 && swiglu)) return pf;
  const int cus = srgpt_device_cus();
  const int per_cu = (size_t)batch * K * 2 > 70 * 1024 ? 1 : 2;
  pf.base = reinterpret_cast<const char*>(W);
  pf.row_bytes = row_bytes;
  pf.prefix_bytes = prefix_bytes > 0 ? prefix_bytes : (int)row_bytes;
  pf.n_units = fp8 ? (N + 1) / 2 : N;
  pf.unit_rows = (fp8 || swiglu) ? 2 : 1;
  pf.umul = fp8 ? 2 : 1;
  pf.rstride = fp8 ? 1 : N;
  int grid = (pf.n_units + 3) / 4;
  if (grid > cus * per_cu) grid = cus * per_cu;
  pf.gemv_grid = grid;
  pf.rounds = rounds;
  pf.nblocks = grid;
  pf.n_rows = swiglu ? 2 * N : N;
  pf.batch = SRGPT_KNOB("SRGPT_DECODE_PREFETCH_BATCH", 2);  // measured 1 / 2 / 4 / 8: 3.016 / 2.948 / 2.986 / 3.000 ms per token (profiles/r03_decode_attention.txt)
  return pf;
}
// descriptor for "the next launch is the batched MFMA product (skinny.hip) over a PACKED matrix of 16-row granules, one tile per block"
// (o_proj / down_proj of the batched fp8 decode step): block p of that launch -- XCD p % 8, like prefetch block p -- streams tile p
static inline SrgptPrefetch srgpt_prefetch_for_packed_tiles(const void* Wp, int N, int K, int elem_bytes, int batch_loads) {
  SrgptPrefetch pf{nullptr, 0, 0, 0, 1, 1, 0, 1, 0, 0, 1, 1, 0};
  const long long tile = 16LL * K * elem_bytes;
  if (!Wp || batch_loads <= 0 || tile % 1024 != 0 || N % 16 != 0 || N / 16 > srgpt_device_cus()) return pf;
  pf.base = reinterpret_cast<const char*>(Wp);
  pf.tile_bytes = (int)tile;
  pf.n_units = N / 16;
  pf.gemv_grid = pf.nblocks = N / 16;
  pf.rounds = 1;
  pf.batch = batch_loads;
  return pf;
}
