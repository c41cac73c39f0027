// Decode-path weight-streaming GEMV (bf16: one row, fp32: up to 4; more rows -> skinny.hip): out[b, n] = W[n, :] . x[b, :]   (HBM-bound)
//
// Roofline: every weight byte is read exactly once per token (non-temporal 16-byte loads, one 1 KiB
// wave-instruction per row chunk); x lives in LDS; fp32 accumulate on the VALU (24 lane-ops per 16 B --
// ~12 % of VALU issue at HBM rate, so no MFMA reshaping).
//
// Work decomposition: a unit = one output row (plain) or one (gate row, up row) pair (SwiGLU); each wave owns
// units grid-strided; per unit the K range is walked in batches of 8 row chunks: 8 independent 1-KiB loads are
// issued back to back, then consumed in order with counted waits.  No cross-batch software pipeline: measured
// (scripts/experiments/ubench_stream.hip, ubench_gemv_ts.hip) a 3-deep register pipeline with 16+ loads in flight per wave
// was 2-3 us SLOWER per launch -- 8 waves/CU x 8 KiB already saturate HBM, and the independent waves of a CU
// drift apart so that some stream while others multiply.  Grid = 2 blocks per CU.
//
// Fusions (include/srgpt.h): RMSNorm prologue (LlamaRMSNorm), SwiGLU epilogue, residual add, fp32 logits.
// Rounding points mirror PyTorch's bf16 materialisation of each intermediate.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// Optional per-wave timestamping (scripts/experiments/ubench_gemv_ts.hip builds this file with -DSRGPT_GEMV_TS)
#ifdef SRGPT_GEMV_TS
__device__ long long srgpt_gemv_ts[8 * 8192];
extern "C" void* srgpt_gemv_ts_ptr() {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(srgpt_gemv_ts));
  return p;
}
#define SRGPT_TS(slot)                                                                             \
  do {                                                                                             \
    if (lane == 0) srgpt_gemv_ts[((int)blockIdx.x * 4 + wave) * 8 + (slot)] = wall_clock64();     \
  } while (0)
#else
#define SRGPT_TS(slot)
#endif

int srgpt_skinny_launch(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out,
                        int batch, int N, int K, int swiglu, int out_f32, const float* ss_in, float* ss_out, int packed,
                        hipStream_t s);  // skinny.hip
int srgpt_skinny_w8_launch(const void* x, const void* W8, const float* wscale, const void* norm_w, float norm_eps,
                           const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32,
                           const float* ss_in, float* ss_out, int packed, hipStream_t s);  // skinny.hip
int srgpt_w8_valu_max_batch();  // skinny.hip

// (compile-time variants: tuning / micro-benchmark builds only -- a product build that defines one is refused)
#if !defined(SRGPT_TUNING_KNOBS) && (defined(SRGPT_GEMV_TS) || defined(SRGPT_GEMV_REG_PIPE) || defined(SRGPT_GEMV_PIPE))
#error "gemv.hip: -DSRGPT_GEMV_* variants need -DSRGPT_TUNING_KNOBS (make TUNING=1, scripts/experiments/ubench_gemv_ts.hip)"
#endif
#ifndef SRGPT_GEMV_REG_PIPE
#define SRGPT_GEMV_REG_PIPE 0  // the same for the register-resident variant (o_proj: its weights are L2-prefetched; measured
                               // 3.036 vs 3.042 ms per token, o_proj 5.0 vs 5.45 us -- off)
#endif
#ifndef SRGPT_GEMV_PIPE
#define SRGPT_GEMV_PIPE 1  // 0: issue -> consume per batch, nothing in flight across the prologue / reductions (A/B builds)
#endif

namespace {

template <typename T>
struct WChunk;  // 16 bytes of weights -> VEC floats
template <>
struct WChunk<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void cvt(const u32x4& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = bf16lo(r[i]);
      f[2 * i + 1] = bf16hi(r[i]);
    }
  }
};
template <>
struct WChunk<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void cvt(const u32x4& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(r[i]);
  }
};

// UB = loads per batch (8; 7 for rows whose chunk iterations are a multiple of 7 and not of 8 -- K = 14336, down_proj: 28 iterations are
// 4 batches of 7, where batches of 8 spent every row's fourth batch half on clamped re-reads of the row's last chunk)
template <typename T, int B, bool SWIGLU, int NXMAX, int UB = 8>
__global__ __launch_bounds__(256, 2) void gemv_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                      const T* __restrict__ norm_w, float norm_eps,
                                                      const T* __restrict__ residual, void* __restrict__ out, int N,
                                                      int K, int out_f32) {
  constexpr int VEC = WChunk<T>::VEC;
  constexpr int R = SWIGLU ? 2 : 1;  // weight rows per unit
  constexpr int U = UB / R;          // K-chunks per row per batch: 8 loads (8 KiB per wave) in flight, then consumed
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);  // [B][K]
  __shared__ float red[16];
  constexpr int RES_MAXU = 4;  // residual elements of a wave's first 4 units are staged through LDS by the prologue
  __shared__ float res_s[B][4][RES_MAXU];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = K / VEC;          // 16-byte chunks per row
  const int nit = (nchunks + 63) >> 6;  // chunk iterations per row (64 lanes each)
  SRGPT_TS(0);

  // one batch = 8 independent 1-KiB wave loads (R rows x U chunks), issued back to back in consumption order, indices clamped,
  // never branched, so the compiler can place counted s_waitcnt vmcnt(N) in front of each consumer
  u32x4 w[R][U];
  auto issue = [&](int unit, int it0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)(unit + r * N) * K);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        w[r][j] = __builtin_nontemporal_load(p + min((it0 + j) * 64 + lane, nchunks - 1));
        // pin ISSUE order == CONSUMPTION order: left alone, the scheduler issued the first-consumed chunk last,
        // which turns the counted waits below into a full vmcnt(0) drain before the first FMA
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue: stage x (and RMSNorm it) into LDS ----
  // All global loads of the prologue (activation chunks AND norm gains) are issued up front, branch-free, so the
  // block pays one L2 latency.  NXMAX (2 or 8, picked by the host) = chunks per thread held in registers; unused
  // slots wrap around to valid chunks (a clamped address would hot-spot one cache line from every thread of the
  // grid); chunks beyond NXMAX*256 (B*K > 16384 elements) take the generic loop.
  {
    const int total = B * nchunks;
    const bool do_norm = norm_w != nullptr;
    Vec16<T> xr[NXMAX], gr[NXMAX];
    // residual elements this block will need: fetched with the prologue's loads (a load placed next to its use at
    // the end of a row gets sunk behind the weight stream by the compiler and exposes a full memory latency per row)
    // Round 5: the load is UNCONDITIONAL and its value is not touched until the LDS store below.  Written as
    // `if (residual) r = to_f(residual[i])` the compiler emitted branch -> load -> s_waitcnt vmcnt(0) -> convert at the very top of
    // the kernel: o_proj and down_proj spent a whole memory round trip (the row was just written by the launch before: an L2 miss)
    // before requesting their first activation or weight byte.
    const int rb = tid / (4 * RES_MAXU), rw = (tid / RES_MAXU) & 3, rk = tid % RES_MAXU;
    const int runit = (int)blockIdx.x * 4 + rw + rk * (int)gridDim.x * 4;
    const bool rok = !SWIGLU && residual != nullptr && tid < B * 4 * RES_MAXU && runit < N;
    const T res_raw = (residual != nullptr ? residual : x)[rok ? (size_t)rb * N + runit : 0];
#pragma unroll
    for (int j = 0; j < NXMAX; ++j) {
      const int c = (tid + 256 * j) % total;
      xr[j] = *reinterpret_cast<const Vec16<T>*>(x + (size_t)c * VEC);
      gr[j] = *reinterpret_cast<const Vec16<T>*>((do_norm ? norm_w : x) + (size_t)(c % nchunks) * VEC);
    }
#if SRGPT_GEMV_PIPE
    // the wave's first weight batch goes out BEHIND the prologue's own loads (in-order return: the statistics below wait for the
    // activations only) and its HBM latency overlaps the RMSNorm, the LDS staging and the barrier
    __builtin_amdgcn_sched_barrier(0);
    issue(min((int)blockIdx.x * 4 + wave, N - 1), 0);
    __builtin_amdgcn_sched_barrier(0);
#endif
    float ss[B];
#pragma unroll
    for (int b = 0; b < B; ++b) ss[b] = 0.f;
#pragma unroll
    for (int j = 0; j < NXMAX; ++j) {
      const int c = tid + 256 * j;
      const bool ok = c < total;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) sq += xr[j].get(i) * xr[j].get(i);
      const int b = (B == 1) ? 0 : (c % total) / nchunks;
#pragma unroll
      for (int bb = 0; bb < B; ++bb) ss[bb] += (ok && bb == b) ? sq : 0.f;
    }
    for (int c = tid + 256 * NXMAX; c < total; c += 256) {  // rare
      const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + (size_t)c * VEC);
      *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
      const int b = c / nchunks;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) sq += v.get(i) * v.get(i);
#pragma unroll
      for (int bb = 0; bb < B; ++bb)
        if (bb == b) ss[bb] += sq;
    }
    if (rok) res_s[rb][rw][rk] = to_f(res_raw);
    float rs[B];
#pragma unroll
    for (int b = 0; b < B; ++b) rs[b] = 1.f;
    if (do_norm) {
#pragma unroll
      for (int b = 0; b < B; ++b) rs[b] = rsqrtf(block_sum(ss[b], red) / (float)K + norm_eps);
    }
#pragma unroll
    for (int j = 0; j < NXMAX; ++j) {
      const int c = tid + 256 * j;
      if (c < total) {
        Vec16<T> v = xr[j];
        if (do_norm) {
          const int b = (B == 1) ? 0 : c / nchunks;
          float r = rs[0];
#pragma unroll
          for (int bb = 1; bb < B; ++bb)
            if (bb == b) r = rs[bb];
#pragma unroll
          for (int i = 0; i < VEC; ++i) v.set(i, gr[j].get(i) * rnd<T>(xr[j].get(i) * r));  // weight * h.to(dtype)
        }
        *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
      }
    }
    if (do_norm) {
      for (int c = tid + 256 * NXMAX; c < total; c += 256) {  // rare; each thread re-reads the chunks it wrote itself
        const int b = c / nchunks, kc = c - b * nchunks;
        Vec16<T> v = *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC);
        const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(norm_w + (size_t)kc * VEC);
        float r = rs[0];
#pragma unroll
        for (int bb = 1; bb < B; ++bb)
          if (bb == b) r = rs[bb];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v.set(i, g.get(i) * rnd<T>(v.get(i) * r));
        *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
      }
    }
    __syncthreads();
  }
  SRGPT_TS(2);

  int uk = 0;
  for (int unit = blockIdx.x * 4 + wave; unit < N; unit += gridDim.x * 4, ++uk) {
    float acc[R][B];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;

    for (int it0 = 0; it0 < nit; it0 += U) {
      // Waves drift apart naturally, so some stream while others multiply; 8 waves/CU keep 64 KiB in flight.
#if !SRGPT_GEMV_PIPE
      issue(unit, it0);
#endif
#pragma unroll
      for (int j = 0; j < U; ++j) {
        // branch-free: a chunk index past the row end is clamped for the loads and its weights are zeroed here.
        // (A per-chunk `if` made the compiler sink the chunk's global load INTO the branch, behind all the others,
        //  followed by s_waitcnt vmcnt(0): every batch was fully drained before its first FMA.)
        const int ch = (it0 + j) * 64 + lane;
        const bool valid = ch < nchunks;
        const int chc = valid ? ch : nchunks - 1;
        float wf[R][VEC];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          u32x4 wv = w[r][j];
#pragma unroll
          for (int q = 0; q < 4; ++q) wv[q] = valid ? wv[q] : 0u;
          WChunk<T>::cvt(wv, wf[r]);
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(xs + (size_t)b * K + (size_t)chc * VEC);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float xf = xv.get(i);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][b] = fmaf(wf[r][i], xf, acc[r][b]);
          }
        }
      }
#if SRGPT_GEMV_PIPE
      // the registers are free again: the next batch of this wave's stream (the row's next chunks, or the first chunks of its
      // next unit; past the last unit a valid row is re-read and dropped) goes out before the reduction and the store
      {
        const bool more = it0 + U < nit;
        issue(more ? unit : min(unit + (int)gridDim.x * 4, N - 1), more ? it0 + U : 0);
      }
#endif
    }

#pragma unroll
    for (int b = 0; b < B; ++b) {
      float a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = wave_sum(acc[r][b]);
      if (lane == 0) {
        if (SWIGLU) {
          const float g = rnd<T>(a[0]), u = rnd<T>(a[R - 1]);
          reinterpret_cast<T*>(out)[(size_t)b * N + unit] = from_f<T>(rnd<T>(silu(g)) * u);
        } else {
          float v = rnd<T>(a[0]);
          if (residual) v = rnd<T>((uk < RES_MAXU ? res_s[b][wave][uk] : to_f(residual[(size_t)b * N + unit])) + v);
          if (out_f32)
            reinterpret_cast<float*>(out)[(size_t)b * N + unit] = v;
          else
            reinterpret_cast<T*>(out)[(size_t)b * N + unit] = from_f<T>(v);
        }
      }
    }
  }
  SRGPT_TS(4);
}

// ------------------------------------------------------------------------------------------------
// Batch-1 bf16 variant with the activation row held in REGISTERS (no LDS, no block barrier).  Lane l only ever multiplies
// chunks l, l + 64, ... of the row, so a wave keeps NIT = ceil(K / 512) packed 16-byte chunks per lane (32 VGPRs at
// K = 4096, 112 at 14336) for all its rows; every wave holds the whole row across its lanes, so the RMSNorm sum of squares
// is one wave_sum -- no cross-wave reduction, no __syncthreads.  The prologue shrinks to NIT (+NIT gain) loads and a
// shuffle reduction, the inner loop loses its ds_read per chunk, and a K that is not a multiple of 4096 wastes no load
// slots (NIT is a template parameter: static register indexing, exact batch sizes).  Residual elements of the wave's first
// units are fetched up front by lanes 0..3 and broadcast with a shuffle.
// ------------------------------------------------------------------------------------------------
template <bool SWIGLU, bool NORM, int NIT, int UB = 8>
__global__ __launch_bounds__(256, 2) void gemv_reg_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                          const bf16_t* __restrict__ norm_w, float norm_eps,
                                                          const bf16_t* __restrict__ residual, void* __restrict__ out, int N,
                                                          int K, int out_f32) {
  typedef bf16_t T;
  constexpr int VEC = 8;
  constexpr int R = SWIGLU ? 2 : 1;
  constexpr int U = UB / R;
  constexpr int RES_MAXU = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = K / VEC;

  // one batch of the weight stream: the static chunks IT0 .. IT0 + U - 1 of the unit's R rows (issue order == consumption order)
  u32x4 w[R][U];
  auto issue = [&](int unit, auto it0_c) {
    constexpr int IT0 = decltype(it0_c)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)(unit + r * N) * K);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (IT0 + j < NIT) {  // static
          w[r][j] = __builtin_nontemporal_load(p + min((IT0 + j) * 64 + lane, nchunks - 1));
          __builtin_amdgcn_sched_barrier(0);  // issue order == consumption order (see gemv_kernel)
        }
      }
    }
  };

  // ---- prologue: the lane's chunks of x (and gains), straight to registers; nothing here is shared between waves ----
  u32x4 xp[NIT];
  unsigned int res_raw;  // bf16 bits of the residual element of unit `lane` (lanes 0..3); unconditional load, converted at its use
  {
    {
      const int ru = (int)blockIdx.x * 4 + wave + min(lane, RES_MAXU - 1) * (int)gridDim.x * 4;
      const bool has = !SWIGLU && residual != nullptr;  // (see gemv_kernel: a branch here cost a memory round trip per launch)
      res_raw = *reinterpret_cast<const unsigned short*>((has ? residual : x) + (has ? min(ru, N - 1) : 0));
    }
    u32x4 gr[NORM ? NIT : 1];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int c = min(j * 64 + lane, nchunks - 1);
      xp[j] = *reinterpret_cast<const u32x4*>(x + (size_t)c * VEC);
      if (NORM) gr[j] = *reinterpret_cast<const u32x4*>(norm_w + (size_t)c * VEC);
    }
#if SRGPT_GEMV_REG_PIPE
    // first weight batch behind the activation loads (see gemv_kernel): its latency overlaps theirs and the statistics
    __builtin_amdgcn_sched_barrier(0);
    issue(min((int)blockIdx.x * 4 + wave, N - 1), std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (NORM) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const bool ok = j * 64 + lane < nchunks;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = bf16lo(xp[j][q]), hi = bf16hi(xp[j][q]);
          sq += lo * lo + hi * hi;
        }
        ss += ok ? sq : 0.f;
      }
      const float r = rsqrtf(wave_sum(ss) / (float)K + norm_eps);
#pragma unroll
      for (int j = 0; j < NIT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // weight * hidden.to(dtype): two roundings, as LlamaRMSNorm materialises them
          bf16x2 p;
          p[0] = (bf16_t)(bf16lo(gr[j][q]) * rnd<T>(bf16lo(xp[j][q]) * r));
          p[1] = (bf16_t)(bf16hi(gr[j][q]) * rnd<T>(bf16hi(xp[j][q]) * r));
          xp[j][q] = __builtin_bit_cast(unsigned int, p);
        }
    }
  }

  int uk = 0;
  for (int unit = blockIdx.x * 4 + wave; unit < N; unit += gridDim.x * 4, ++uk) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    if constexpr (NIT > 16) {
      // long rows: the activations are loop-invariant, and left alone the compiler hoists their bf16 -> fp32 unpacking out of the
      // unit loop -- twice the registers for x (224 at K = 14336) and the kernel spills; opaque per unit, the packed form stays
#pragma unroll
      for (int j = 0; j < NIT; ++j) asm volatile("" : "+v"(xp[j]));
    }
    auto batch = [&](auto it0_c) {
      constexpr int it0 = decltype(it0_c)::value;
#if !SRGPT_GEMV_REG_PIPE
      issue(unit, it0_c);
#endif
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (it0 + j < NIT) {  // static
          const bool valid = (it0 + j) * 64 + lane < nchunks;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x0 = bf16lo(xp[it0 + j][q]), x1 = bf16hi(xp[it0 + j][q]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const unsigned int wq = valid ? w[r][j][q] : 0u;
              acc[r] = fmaf(bf16lo(wq), x0, acc[r]);
              acc[r] = fmaf(bf16hi(wq), x1, acc[r]);
            }
          }
        }
      }
#if SRGPT_GEMV_REG_PIPE
      // the next batch of the wave's stream goes out before the reduction and the store (past the last unit: a valid row, dropped)
      if constexpr (it0 + U < NIT) issue(unit, std::integral_constant<int, it0 + U>{});
      else issue(min(unit + (int)gridDim.x * 4, N - 1), std::integral_constant<int, 0>{});
#endif
    };
    batch(std::integral_constant<int, 0>{});
    if constexpr (U < NIT) batch(std::integral_constant<int, U>{});
    if constexpr (2 * U < NIT) batch(std::integral_constant<int, 2 * U>{});
    if constexpr (3 * U < NIT) batch(std::integral_constant<int, 3 * U>{});
    static_assert(4 * U >= NIT, "gemv_reg_kernel: at most 4 batches per row");
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = wave_sum(acc[r]);
    // lane uk of res_pre holds the residual element of this unit (uniform index: v_readlane, not a ds_bpermute)
    const float res_b = __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)res_raw, uk < RES_MAXU ? uk : 0) << 16);
    if (lane == 0) {
      if (SWIGLU) {
        const float g = rnd<T>(a[0]), u = rnd<T>(a[R - 1]);
        reinterpret_cast<T*>(out)[unit] = from_f<T>(rnd<T>(silu(g)) * u);
      } else {
        float v = rnd<T>(a[0]);
        if (residual) v = rnd<T>((uk < RES_MAXU ? res_b : to_f(residual[unit])) + v);
        if (out_f32)
          reinterpret_cast<float*>(out)[unit] = v;
        else
          reinterpret_cast<T*>(out)[unit] = from_f<T>(v);
      }
    }
  }
}

template <bool SWIGLU, bool NORM>
bool launch_gemv_reg(int nit, int grid, hipStream_t s, const void* x, const void* W, const void* norm_w, float eps,
                     const void* residual, void* out, int N, int K, int out_f32) {
#define SRGPT_REG_CASE(NITV)                                                                                                   \
  case NITV:                                                                                                                   \
    hipLaunchKernelGGL((gemv_reg_kernel<SWIGLU, NORM, NITV>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)W, \
                       (const bf16_t*)norm_w, eps, (const bf16_t*)residual, out, N, K, out_f32);                               \
    return true
  switch (nit) {
    SRGPT_REG_CASE(5);   // K = 2560
    SRGPT_REG_CASE(8);   // K = 4096
    SRGPT_REG_CASE(14);  // K = 6912
    default: break;
  }
  if (!SWIGLU && !NORM && nit == 28 && SRGPT_KNOB("SRGPT_GEMV_REG_LONG", 0)) {  // K = 14336 (tuning builds): 112 VGPRs of activations, batches of 7
    hipLaunchKernelGGL((gemv_reg_kernel<false, false, 28, 7>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)W,
                       (const bf16_t*)norm_w, eps, (const bf16_t*)residual, out, N, K, out_f32);
    return true;
  }
  return false;  // longer rows (K = 11008, 14336: 88-112 VGPRs of activations, spills when unrolled) stay on the LDS kernel
#undef SRGPT_REG_CASE
}

template <typename T, int B>
int launch_gemv(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out, int N,
                int K, int swiglu, int out_f32, hipStream_t s) {
  const size_t lds = (size_t)B * K * sizeof(T);
  SRGPT_CHECK(lds <= 150 * 1024, SRGPT_ERR_UNSUPPORTED, "srgpt_gemv: batch*K too large for LDS (%zu bytes)", lds);
  const int cus = srgpt_device_cus();
  const int env_per_cu = SRGPT_KNOB("SRGPT_GEMV_BLOCKS_PER_CU", 0);  // tuning knob
  // (round 5, measured and not kept: 3 / 4 / 5 blocks per CU for the short launches -- q/k/v, o_proj, where a wave owns only 2 - 3 rows,
  // i.e. 2 - 3 dependent memory round trips: 2.970 -> 2.998 / 3.009 / 3.017 ms per token; for all launches 3.037 / 3.049:
  // profiles/r05_decode_step_ab.txt)
  const int per_cu = lds > 70 * 1024 ? 1 : (env_per_cu > 0 ? env_per_cu : 2);
  int grid = (N + 3) / 4;
  if (grid > cus * per_cu) grid = cus * per_cu;
  if (grid < 1) grid = 1;
  const int chunks = B * (K / WChunk<T>::VEC);
  const int use_reg = SRGPT_KNOB("SRGPT_GEMV_REG", 1);  // A/B knob
  if (B == 1 && sizeof(T) == 2 && use_reg) {
    const int nit = (K / 8 + 63) / 64;
    // measured (scripts/experiments/ubench_gemv_c.hip): without the fused RMSNorm the register variant saves 0.6-0.8 us per launch
    // (o_proj 8.5 -> 7.9 us); with it every wave normalises the whole row redundantly and loses ~1 us -> LDS kernel
    bool ok = false;
    if (!norm_w)
      ok = swiglu ? launch_gemv_reg<true, false>(nit, grid, s, x, W, norm_w, eps, residual, out, N, K, out_f32)
                  : launch_gemv_reg<false, false>(nit, grid, s, x, W, norm_w, eps, residual, out, N, K, out_f32);
    if (ok) {
      SRGPT_LAUNCH_CHECK();
      return SRGPT_OK;
    }
  }
#define SRGPT_GEMV_LAUNCH(SW, NXV, ...)                                                                         \
  do {                                                                                                          \
    auto kfn = gemv_kernel<T, B, SW, NXV, ##__VA_ARGS__>;                                                       \
    static std::atomic<uint64_t> attr_done{0};                                                                  \
    SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)kfn, lds > 48 * 1024 ? 150 * 1024 : 0));             \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, (const T*)x, (const T*)W, (const T*)norm_w, eps,     \
                       (const T*)residual, out, N, K, out_f32);                                                 \
  } while (0)
  const int nit = (K / WChunk<T>::VEC + 63) / 64;
  if (swiglu) {
    if (chunks <= 512) SRGPT_GEMV_LAUNCH(true, 2); else SRGPT_GEMV_LAUNCH(true, 8);
  } else if (B == 1 && sizeof(T) == 2 && nit % 7 == 0 && nit % 8 != 0 && chunks > 512 && SRGPT_KNOB("SRGPT_GEMV_U7", 1)) {
    SRGPT_GEMV_LAUNCH(false, 8, 7);
  } else {
    if (chunks <= 512) SRGPT_GEMV_LAUNCH(false, 2); else SRGPT_GEMV_LAUNCH(false, 8);
  }
#undef SRGPT_GEMV_LAUNCH
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

// bf16 rows >= this go to the MFMA kernel of skinny.hip
inline int skinny_min_batch() { return SRGPT_KNOB("SRGPT_SKINNY_MIN_BATCH", 2); }  // measured (round 3, profiles/r03_skinny_min_batch.txt): VALU wins at 1 row, MFMA from 2

template <typename T>
int dispatch_b(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out,
               int batch, int N, int K, int swiglu, int out_f32, hipStream_t s) {
  const int skinny_min = skinny_min_batch();
  if (batch > 4 || (sizeof(T) == 2 && batch >= skinny_min)) {
    // bf16: rows go through the MFMA skinny kernel 16 at a time (skinny.hip); fp32 (parity dtype of the tiny models):
    // 4 rows at a time through the VALU kernel.  Each chunk streams the weights once.
    const bool mfma = sizeof(T) == 2;
    int step = mfma ? 16 : 4;
    if (!mfma)
      while (step > 1 && (size_t)step * K * sizeof(T) > 150 * 1024) --step;
    const size_t on = out_f32 ? sizeof(float) : sizeof(T);
    for (int b0 = 0; b0 < batch; b0 += step) {
      const int nb = batch - b0 < step ? batch - b0 : step;
      const void* xb = (const char*)x + (size_t)b0 * K * sizeof(T);
      const void* rb = residual ? (const char*)residual + (size_t)b0 * N * sizeof(T) : nullptr;
      void* ob = (char*)out + (size_t)b0 * N * on;
      // (a single-row tail of a longer batch -- 17, 33 rows -- stays on the kernel its other rows took, as srgpt_gemv_rowss does)
      if (mfma && (nb > 4 || nb >= skinny_min || b0 > 0))
        SRGPT_TRY(srgpt_skinny_launch(xb, W, norm_w, eps, rb, ob, nb, N, K, swiglu, out_f32, nullptr, nullptr, 0, s));
      else
        SRGPT_TRY((dispatch_b<T>(xb, W, norm_w, eps, rb, ob, nb, N, K, swiglu, out_f32, s)));
    }
    return SRGPT_OK;
  }
  switch (batch) {
    case 1: return launch_gemv<T, 1>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    case 2: return launch_gemv<T, 2>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    case 3: return launch_gemv<T, 3>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    case 4: return launch_gemv<T, 4>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    default:
      srgpt_set_error("srgpt_gemv: batch %d not supported", batch);
      return SRGPT_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int srgpt_gemv(const void* x, const void* W, const void* norm_w, float norm_eps, const void* residual,
                          void* out, int batch, int N, int K, int swiglu, int out_f32, int dtype,
                          srgpt_stream_t stream) {
  SRGPT_CHECK(x && W && out, SRGPT_ERR_ARG, "srgpt_gemv: null pointer");
  SRGPT_CHECK(N > 0 && K > 0 && batch > 0, SRGPT_ERR_ARG, "srgpt_gemv: bad shape");
  SRGPT_CHECK(dtype == SRGPT_F32 || dtype == SRGPT_BF16, SRGPT_ERR_ARG, "srgpt_gemv: bad dtype %d", dtype);
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(K % vec == 0, SRGPT_ERR_ARG, "srgpt_gemv: K=%d must be a multiple of %d", K, vec);
  SRGPT_CHECK(!(swiglu && (residual || out_f32)), SRGPT_ERR_ARG, "srgpt_gemv: swiglu excludes residual/out_f32");
  hipStream_t s = as_stream(stream);
  if (dtype == SRGPT_BF16)
    return dispatch_b<bf16_t>(x, W, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, s);
  return dispatch_b<float>(x, W, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, s);
}

// ------------------------------------------------------------------------------------------------
// The same products with the row-statistics hand-off of skinny.hip (ABI 8): bf16 activations, 2+ rows (the MFMA kernel).
// ------------------------------------------------------------------------------------------------
extern "C" int srgpt_gemv_rowss_supported(int batch, int dtype, int fp8) {
  if (dtype != SRGPT_BF16 || batch < 2) return 0;
  if (2 * srgpt_device_cus() > SRGPT_ROWSS_STRIDE) return 0;  // one slot per producer block (two 4-wave blocks per CU)
  return fp8 ? (batch > srgpt_w8_valu_max_batch() ? 1 : 0) : (batch >= skinny_min_batch() ? 1 : 0);
}

extern "C" int srgpt_gemv_rowss(const void* x, const void* W, const void* W8, const float* wscale, const void* norm_w,
                                float norm_eps, const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32,
                                const float* rowss_in, float* rowss_out, int packed_rows, srgpt_stream_t stream) {
  SRGPT_CHECK(x && (W || W8) && out, SRGPT_ERR_ARG, "srgpt_gemv_rowss: null pointer");
  SRGPT_CHECK(packed_rows == 0 || packed_rows == 4 || packed_rows == 8 || packed_rows == 16, SRGPT_ERR_ARG,
              "srgpt_gemv_rowss: packed_rows = %d (0: row-major; 4, 8 or 16 rows per granule)", packed_rows);
  SRGPT_CHECK(packed_rows == 0 || (N % packed_rows == 0 && K % (W8 ? 64 : 32) == 0), SRGPT_ERR_ARG,
              "srgpt_gemv_rowss: the packed layout needs N %% %d == 0 and K %% %d == 0 (N = %d, K = %d)", packed_rows, W8 ? 64 : 32, N, K);
  SRGPT_CHECK(!W8 || wscale, SRGPT_ERR_ARG, "srgpt_gemv_rowss: fp8 weights without row scales");
  SRGPT_CHECK(N > 0 && K > 0 && batch > 0 && K % 8 == 0, SRGPT_ERR_ARG, "srgpt_gemv_rowss: bad shape");
  SRGPT_CHECK(!(swiglu && (residual || out_f32)), SRGPT_ERR_ARG, "srgpt_gemv_rowss: swiglu excludes residual/out_f32");
  SRGPT_CHECK(srgpt_gemv_rowss_supported(batch, SRGPT_BF16, W8 != nullptr), SRGPT_ERR_UNSUPPORTED,
              "srgpt_gemv_rowss: %d row(s) of %s weights take a kernel without the statistics hand-off", batch, W8 ? "fp8" : "bf16");
  hipStream_t s = as_stream(stream);
  if (W8) return srgpt_skinny_w8_launch(x, W8, wscale, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, rowss_in, rowss_out, packed_rows, s);
  const size_t on = out_f32 ? sizeof(float) : 2;
  for (int b0 = 0; b0 < batch; b0 += 16) {
    const int nb = batch - b0 < 16 ? batch - b0 : 16;
    SRGPT_TRY(srgpt_skinny_launch((const char*)x + (size_t)b0 * K * 2, W, norm_w, norm_eps,
                                  residual ? (const char*)residual + (size_t)b0 * N * 2 : nullptr, (char*)out + (size_t)b0 * N * on, nb,
                                  N, K, swiglu, out_f32, rowss_in ? rowss_in + (size_t)b0 * SRGPT_ROWSS_STRIDE : nullptr,
                                  rowss_out ? rowss_out + (size_t)b0 * SRGPT_ROWSS_STRIDE : nullptr, packed_rows, s));
  }
  return SRGPT_OK;
}
