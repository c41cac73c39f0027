// Decode-path weight-streaming GEMV with fp8 weights for 1-2 batch rows (W8A16, BASELINE config 5):
//   out[b, n] = round_bf16( (sum_k x[b, k] * fp8(W8[n, k])) * wscale[n] )            (HBM-bound, half the bytes of bf16)
//
// Same structure as gemv.hip (wave per unit, 8 independent 1-KiB non-temporal wave loads per batch issued in consumption
// order, branch-free consumption with counted vmcnt, x in LDS, fused RMSNorm / SwiGLU / residual / fp32-logit epilogues,
// prologue loads issued up front) with these differences:
//   * a 16-byte chunk holds 16 weights, so a K = 4096 row is 4 wave loads: a unit is TWO output columns (plain: rows 2u,
//     2u+1 -> 8 loads per batch; SwiGLU: gate rows 2u, 2u+1 and up rows N+2u, N+2u+1 -> 16 loads = 16 KiB per batch:
//     at half the bytes per row the per-row latency dominates, fewer and larger rounds win here);
//   * weights are widened with v_cvt_pk_f32_fp8 (OCP e4m3fn on gfx950), 2 values per instruction; the two rows of a unit
//     share the fp32 copies of x;
//   * the per-row scale multiplies the reduced dot product.
// More than 2 rows go through the MFMA skinny kernel (skinny.hip, W8 variant).
#include <stdlib.h>

#include "common.h"

namespace {

typedef bf16_t T;

template <int B, bool SWIGLU, int NXMAX>
__global__ __launch_bounds__(256, 2) void gemv_w8_kernel(const T* __restrict__ x, const unsigned char* __restrict__ W,
                                                         const float* __restrict__ wscale, const T* __restrict__ norm_w,
                                                         float norm_eps, const T* __restrict__ residual,
                                                         void* __restrict__ out, int N, int K, int out_f32) {
  constexpr int VEC = 8;   // activation elements per 16-byte chunk
  constexpr int WVEC = 16;  // weight elements per 16-byte chunk
  constexpr int R = SWIGLU ? 4 : 2;  // weight rows per unit (SwiGLU: 2 gate + 2 up rows -> 2 outputs; 16 KiB per wave in flight)
  constexpr int U = 4;               // K-chunks per row per batch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);  // [B][K]
  __shared__ float red[16];
  constexpr int RES_MAXU = 4;  // residual elements of a wave's first 4 units are staged through LDS by the prologue
  __shared__ float res_s[B][4][2 * RES_MAXU];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = K / VEC;           // 16-byte activation chunks per row
  const int wchunks = K / WVEC;          // 16-byte weight chunks per row
  const int nit = (wchunks + 63) >> 6;   // weight-chunk iterations per row (64 lanes each)
  const int NUNIT = (N + 1) >> 1;

  // rows of a unit (wave-uniform: scalar addressing).  plain: rows 2u, 2u+1; SwiGLU: gate rows 2u, 2u+1 and up rows N+2u, N+2u+1.
  // The second row of an odd-N tail unit is clamped and never stored.
  auto unit_rows = [&](int unit, int* rows) {
    rows[0] = __builtin_amdgcn_readfirstlane(2 * unit);
    rows[1] = __builtin_amdgcn_readfirstlane(min(2 * unit + 1, N - 1));
    if (SWIGLU) {
      rows[R - 2] = __builtin_amdgcn_readfirstlane(N + 2 * unit);
      rows[R - 1] = __builtin_amdgcn_readfirstlane(N + min(2 * unit + 1, N - 1));
    }
  };
  // one batch of the weight stream: R rows x U chunks, issued in consumption order, indices clamped, never branched
  u32x4 w[R][U];
  auto issue = [&](int unit, int it0) {
    int rows[R];
    unit_rows(unit, rows);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)rows[r] * K);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        w[r][j] = __builtin_nontemporal_load(p + min((it0 + j) * 64 + lane, wchunks - 1));
        __builtin_amdgcn_sched_barrier(0);  // issue order == consumption order (see gemv.hip)
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
    constexpr int RSLOT = 2 * RES_MAXU;  // (unit, row-of-pair) slots per wave
    const int rb = tid / (4 * RSLOT), rw = (tid / RSLOT) & 3, rk = tid % RSLOT;
    const int runit = ((int)blockIdx.x * 4 + rw + (rk >> 1) * (int)gridDim.x * 4) * 2 + (rk & 1);  // output row
    const bool rok = !SWIGLU && residual != nullptr && tid < B * 4 * RSLOT && runit < N;
    // unconditional, untouched until the LDS store (gemv.hip: the branchy form cost a memory round trip at the top of the kernel)
    const T res_raw = (residual != nullptr ? residual : x)[rok ? (size_t)rb * N + runit : 0];
#pragma unroll
    for (int j = 0; j < NXMAX; ++j) {
      const int c = (tid + 256 * j) % total;
      xr[j] = *reinterpret_cast<const Vec16<T>*>(x + (size_t)c * VEC);
      gr[j] = *reinterpret_cast<const Vec16<T>*>((do_norm ? norm_w : x) + (size_t)(c % nchunks) * VEC);
    }
    // first weight batch behind the prologue's own loads: its HBM latency overlaps the RMSNorm, the LDS staging and the barrier
    __builtin_amdgcn_sched_barrier(0);
    issue(min((int)blockIdx.x * 4 + wave, NUNIT - 1), 0);
    __builtin_amdgcn_sched_barrier(0);
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

  int uk = 0;
  for (int unit = blockIdx.x * 4 + wave; unit < NUNIT; unit += gridDim.x * 4, ++uk) {
    int rows[R];
    unit_rows(unit, rows);
    // the row scales are requested BEFORE the weight stream (scalar loads): placed at their use they would queue behind
    // the weight loads and expose a memory latency per unit (same trap as the residual element, see gemv.hip)
    float sc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sc[r] = wscale[rows[r]];
    float acc[R][B];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;

    for (int it0 = 0; it0 < nit; it0 += U) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int ch = (it0 + j) * 64 + lane;
        const bool valid = ch < wchunks;
        const int chc = valid ? ch : wchunks - 1;
        float xf[B][WVEC];
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const u32x4 x0 = *reinterpret_cast<const u32x4*>(xs + (size_t)b * K + (size_t)chc * WVEC);
          const u32x4 x1 = *reinterpret_cast<const u32x4*>(xs + (size_t)b * K + (size_t)chc * WVEC + 8);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            xf[b][2 * q] = bf16lo(x0[q]);
            xf[b][2 * q + 1] = bf16hi(x0[q]);
            xf[b][8 + 2 * q] = bf16lo(x1[q]);
            xf[b][8 + 2 * q + 1] = bf16hi(x1[q]);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int wq = valid ? (int)w[r][j][q] : 0;  // fp8 code 0 = +0.0
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(wq, false);
            const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8(wq, true);
#pragma unroll
            for (int b = 0; b < B; ++b) {
              acc[r][b] = fmaf(lo[0], xf[b][4 * q], acc[r][b]);
              acc[r][b] = fmaf(lo[1], xf[b][4 * q + 1], acc[r][b]);
              acc[r][b] = fmaf(hi[0], xf[b][4 * q + 2], acc[r][b]);
              acc[r][b] = fmaf(hi[1], xf[b][4 * q + 3], acc[r][b]);
            }
          }
        }
      }
      // the next batch of the wave's stream goes out before the reduction and the store (past the last unit: a valid unit, dropped)
      {
        const bool more = it0 + U < nit;
        issue(more ? unit : min(unit + (int)gridDim.x * 4, NUNIT - 1), more ? it0 + U : 0);
      }
    }

#pragma unroll
    for (int b = 0; b < B; ++b) {
      float a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = wave_sum(acc[r][b]) * sc[r];
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int n = 2 * unit + r;
          if (n < N) {
            if (SWIGLU) {
              const float g = rnd<T>(a[r]), u = rnd<T>(a[R - 2 + r]);
              reinterpret_cast<T*>(out)[(size_t)b * N + n] = from_f<T>(rnd<T>(silu(g)) * u);
            } else {
              float v = rnd<T>(a[r]);
              if (residual)
                v = rnd<T>((uk < RES_MAXU ? res_s[b][wave][2 * uk + r] : to_f(residual[(size_t)b * N + n])) + v);
              if (out_f32)
                reinterpret_cast<float*>(out)[(size_t)b * N + n] = v;
              else
                reinterpret_cast<T*>(out)[(size_t)b * N + n] = from_f<T>(v);
            }
          }
        }
      }
    }
  }
}

template <int B>
int launch_gemv_w8(const void* x, const void* W, const float* wscale, const void* norm_w, float eps, const void* residual,
                   void* out, int N, int K, int swiglu, int out_f32, hipStream_t s) {
  const size_t lds = (size_t)B * K * sizeof(T);
  SRGPT_CHECK(lds <= 150 * 1024, SRGPT_ERR_UNSUPPORTED, "srgpt_gemv_w8: batch*K too large for LDS (%zu bytes)", lds);
  const int cus = srgpt_device_cus();
  const int per_cu = lds > 70 * 1024 ? 1 : 2;
  const int nunit = (N + 1) / 2;
  int grid = (nunit + 3) / 4;
  if (grid > cus * per_cu) grid = cus * per_cu;
  if (grid < 1) grid = 1;
  const int chunks = B * (K / 8);
#define SRGPT_W8_LAUNCH(SW, NXV)                                                                               \
  do {                                                                                                         \
    auto kfn = gemv_w8_kernel<B, SW, NXV>;                                                                     \
    static std::atomic<uint64_t> attr_done{0};                                                                 \
    SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)kfn, lds > 48 * 1024 ? 150 * 1024 : 0));            \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, (const T*)x, (const unsigned char*)W, wscale,       \
                       (const T*)norm_w, eps, (const T*)residual, out, N, K, out_f32);                         \
  } while (0)
  if (swiglu) {
    if (chunks <= 512) SRGPT_W8_LAUNCH(true, 2); else SRGPT_W8_LAUNCH(true, 8);
  } else {
    if (chunks <= 512) SRGPT_W8_LAUNCH(false, 2); else SRGPT_W8_LAUNCH(false, 8);
  }
#undef SRGPT_W8_LAUNCH
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

}  // namespace

// host entry used by srgpt_gemv_w8 (skinny.hip) for one row (two under the tuning knob); K must be a multiple of 16 here
int srgpt_gemv_w8_valu(const void* x, const void* W8, const float* wscale, const void* norm_w, float eps,
                       const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32, hipStream_t s) {
  if (batch == 1) return launch_gemv_w8<1>(x, W8, wscale, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
  return launch_gemv_w8<2>(x, W8, wscale, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
}
