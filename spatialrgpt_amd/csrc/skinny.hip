// Batched decode-path weight-streaming product for 5..16 rows: out[b, n] = W[n, :] . x[b, :]   (bf16, HBM-bound)
//
// Same contract and fusions as the GEMV (gemv.hip: RMSNorm prologue, SwiGLU epilogue, residual add, fp32 logits), but
// above 4 rows the VALU formulation runs out of issue slots and LDS bandwidth (B LDS reads + 8 B FMAs per 16 weight
// bytes), so the 16-row batch becomes the M side of v_mfma_f32_16x16x32_bf16 and 16 weight rows the N side.
//
// Roofline: weight bytes read exactly once (non-temporal, each wave-instruction = 2 rows x 512 contiguous bytes).
// MFMA operand fragments need lane (r = l & 15, g = l >> 4) to hold 16 bytes of ROW r -- loading them straight from
// HBM puts 16 different rows in one instruction (64-byte pieces; measured slow, scripts/ubench_stream.hip), so each
// wave stages its 16 x 256 weight tile through a PRIVATE 8 KiB LDS region: 8 coalesced global loads -> 8 ds_write_b128
// -> 8 conflict-free ds_read_b128 fragments (row stride padded to 528 B).  The matching [rows][256] slice of the
// (normalised) activations goes through a second private region the same way (it comes from L2).  Nothing in the K loop
// is shared between waves, so there is NO block barrier in it (LDS ops of one wave execute in order) and the 8-12 waves
// of a CU drift apart like the GEMV's do: some stream while others multiply.
//
// Work decomposition: a unit = 16 output columns (SwiGLU: 16 gate rows + the 16 matching up rows = 2 sub-units);
// block b owns units b, b + grid, ...; up to 4 sub-units per pass keep their 16x16 fp32 accumulators in registers
// (4 VGPRs each).  The 4 waves of a block split K in interleaved 256-wide slices, so every block uses all its waves
// even when N/16 is only one tile per CU (o_proj / down_proj); their partial sums are added in a fixed order through
// LDS at the end of the pass (deterministic).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

int srgpt_gemv_w8_valu(const void* x, const void* W8, const float* wscale, const void* norm_w, float eps,
                       const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32, hipStream_t s);  // gemv_w8.hip

namespace {

constexpr int WSK = 256;               // k per wave slice
constexpr int WROWB = WSK * 2 + 16;    // bytes per staged row (padded: fragment reads hit 64 distinct banks)
constexpr int WSTAGEB = 16 * WROWB;    // weight stage per wave
constexpr int MAXSU = 4;               // sub-units (16-row weight tiles) accumulated per pass

// NI = x rows staged per wave / 2: 2 (batch <= 4), 4 (batch <= 8) or 8 (batch <= 16).  NW = waves per block (the K split): 4, or 8 when
// there are no more units than CUs so that one block per CU still keeps 8 waves streaming.
// W8: the weights are OCP fp8 e4m3fn bytes with one fp32 scale per weight row (W8A16): a stage is 8 loads of 8 bytes per lane
// (2 rows x 256 bytes each), widened to bf16 (exact) on the way into LDS; the row scale multiplies the fp32 dot product.
template <bool SWIGLU, int NI, int NW, bool W8>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void skinny_kernel(const bf16_t* __restrict__ x, const void* __restrict__ Wv,
                                                                          const float* __restrict__ wscale,
                                                                          const bf16_t* __restrict__ norm_w, float norm_eps,
                                                                          const bf16_t* __restrict__ residual, void* __restrict__ out,
                                                                          int B, int N, int K, int out_f32) {
  constexpr int R = SWIGLU ? 2 : 1;
  constexpr int MAXU = MAXSU / R;
  constexpr int XSTAGEB = 2 * NI * WROWB;
  constexpr int NT = 64 * NW;
  using WReg = typename std::conditional<W8, u32x2, u32x4>::type;  // one staged weight load per lane
  constexpr int WEB = W8 ? 1 : 2;                                  // bytes per weight element
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // per wave: [x stage | w stage]; reused for the reduction
  __shared__ float rs_s[16];
  __shared__ float ss_s[16][NW / 2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* xst = smem + wave * (XSTAGEB + WSTAGEB);
  unsigned char* wst = xst + XSTAGEB;
  const int nsl = (K + WSK - 1) / WSK;  // K slices; wave w takes w, w + NW, ...
  const int NU = (N + 15) >> 4;
  const int grid = gridDim.x;
  const int npass = (NU + grid * MAXU - 1) / (grid * MAXU);
  const bool do_norm = norm_w != nullptr;
  const int lrow = lane >> 5, lchunk = lane & 31;  // staging loads: lane -> (row parity, 16-byte chunk of the 512-byte row piece)

  // activation slice: NI loads, each 2 rows x 512 contiguous bytes (rows past the batch re-read row B-1: their outputs
  // are never stored), plus the 512 bytes of RMSNorm gains of the slice
  u32x4 xr[NI];
  u32x4 gr = {0u, 0u, 0u, 0u};
  auto load_x = [&](int sl) {
    int kg = min(sl * WSK + lchunk * 8, K - 8);
    asm volatile("" : "+v"(kg));  // keep the row products out of loop-invariant registers (see issue_w)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const unsigned off = (unsigned)min(2 * j + lrow, B - 1) * (unsigned)K + (unsigned)kg;
      xr[j] = *reinterpret_cast<const u32x4*>(x + off);
    }
    if (do_norm) gr = *reinterpret_cast<const u32x4*>(norm_w + kg);
  };
  auto stage_x = [&](int sl) {
    const bool kvalid = sl * WSK + lchunk * 8 < K;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      u32x4 v = xr[j];
      if (do_norm) {
        const float rsj = rs_s[2 * j + lrow];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // weight * hidden.to(dtype): two roundings, like the GEMV prologue
          const float lo = bf16lo(gr[q]) * rnd<bf16_t>(bf16lo(v[q]) * rsj);
          const float hi = bf16hi(gr[q]) * rnd<bf16_t>(bf16hi(v[q]) * rsj);
          bf16x2 p;
          p[0] = (bf16_t)lo;
          p[1] = (bf16_t)hi;
          v[q] = __builtin_bit_cast(unsigned int, p);
        }
      }
      if (!kvalid) v = u32x4{0u, 0u, 0u, 0u};  // k past K contributes zeros (the weight loads there are clamped)
      *reinterpret_cast<u32x4*>(xst + (2 * j + lrow) * WROWB + lchunk * 16) = v;
    }
  };

  // weight stage of sub-unit su of (pass, slice): 8 loads, each 2 rows x 512 contiguous bytes
  auto issue_w = [&](WReg* w, int pass, int sl, int su) {
    const int unit = (pass * MAXU + su / R) * grid + (int)blockIdx.x;
    // uniform 64-bit base of the unit's first row + a 32-bit per-lane offset.  The per-lane part is made opaque per call:
    // left visible, LICM hoists the 8 x MAXSU row products out of the K loop and holds them in ~64 VGPRs.
    const unsigned char* base = reinterpret_cast<const unsigned char*>(Wv) + ((size_t)unit * 16 + (size_t)(su % R) * N) * K * WEB;
    const int rmax = N - 1 - unit * 16;  // last valid row of the unit (>= 15 except in the last unit)
    int kg = min(sl * WSK + lchunk * 8, K - 8);
    asm volatile("" : "+v"(kg));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned off = ((unsigned)min(2 * j + lrow, rmax) * (unsigned)K + (unsigned)kg) * WEB;
      w[j] = __builtin_nontemporal_load(reinterpret_cast<const WReg*>(base + off));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // 8 fp8 -> 8 bf16 (every e4m3 value is exactly representable: the high half of the fp32 conversion is the bf16)
  auto widen = [&](const WReg& r) -> u32x4 {
    if constexpr (W8) {
      u32x4 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[h], false);
        const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[h], true);
        o[2 * h] = __builtin_amdgcn_perm(__float_as_uint(lo[1]), __float_as_uint(lo[0]), 0x07060302u);
        o[2 * h + 1] = __builtin_amdgcn_perm(__float_as_uint(hi[1]), __float_as_uint(hi[0]), 0x07060302u);
      }
      return o;
    } else {
      return r;
    }
  };

  const int cnt = wave < nsl ? (nsl - wave + NW - 1) / NW : 0;  // slices of this wave
  if (cnt > 0) load_x(wave);

  // ---- RMSNorm statistics of every batch row (LlamaRMSNorm: fp32 mean of squares over K) ----
  auto rms_stats = [&]() {
    constexpr int HT = NT / 2;                      // threads per row parity
    constexpr int UF = 16 / NI;                     // 16 loads in flight per thread
    const int half = tid / HT, kc = tid % HT;       // thread -> rows {2i + half}, chunks kc, kc + HT, ...
    float ss[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) ss[i] = 0.f;
    const int nch = K >> 3;
    for (int c0 = 0; c0 < nch; c0 += UF * HT) {
      u32x4 v[UF][NI];
#pragma unroll
      for (int u = 0; u < UF; ++u)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int b = min(2 * i + half, B - 1);
          const int c = min(c0 + u * HT + kc, nch - 1);
          v[u][i] = *reinterpret_cast<const u32x4*>(x + (size_t)b * K + (size_t)c * 8);
        }
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const bool ok = c0 + u * HT + kc < nch;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          float s = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = bf16lo(v[u][i][q]), hi = bf16hi(v[u][i][q]);
            s += lo * lo + hi * hi;
          }
          ss[i] += ok ? s : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float t = wave_sum(ss[i]);
      if (lane == 0) ss_s[2 * i + half][wave % (NW / 2)] = t;
    }
    __syncthreads();
    if (tid < 2 * NI) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW / 2; ++k) t += ss_s[tid][k];
      rs_s[tid] = rsqrtf(t / (float)K + norm_eps);
    }
    __syncthreads();
  };

  // one pass over K for NSU sub-units
  auto run_pass = [&](auto nsu_c, int pass, int nu) {
    constexpr int NSU = decltype(nsu_c)::value;
    f32x4 acc[NSU];
#pragma unroll
    for (int su = 0; su < NSU; ++su) acc[su] = f32x4{0.f, 0.f, 0.f, 0.f};
    // register ring of DEPTH weight stages with static slot indices (a trip = DEPTH slices x NSU stages, a multiple of
    // DEPTH): the loads of stage t+DEPTH-1 are issued before stage t is multiplied.  bf16: 2 x 8 KiB, fp8: 2 x 4 KiB.
    // epilogue operands that do not depend on the products -- row scales (W8) and residual elements -- are requested now:
    // fetched at their use they would each add a memory round trip after the last barrier of the pass
    constexpr int EIT0 = ((NSU / R) * 256 + NT - 1) / NT;  // epilogue iterations per thread
    constexpr int EIT = EIT0 > 0 ? EIT0 : 1;               // (odd NSU never occurs with SwiGLU; keep the type valid)
    float pre_sc[EIT][R], pre_res[EIT];
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
      const int e = tid + i * NT;
      const int u = e >> 8, l2 = e & 63, q = (e >> 6) & 3;
      const int b = min(4 * (l2 >> 4) + q, B - 1);
      const int unit = (pass * MAXU + u) * grid + (int)blockIdx.x;
      const int n = min(unit * 16 + (l2 & 15), N - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) pre_sc[i][r] = W8 ? wscale[n + r * N] : 1.f;
      pre_res[i] = (!SWIGLU && residual) ? (float)residual[(size_t)b * N + n] : 0.f;
    }
    constexpr int DEPTH = 2;  // measured: a 4-deep ring of 4 KiB fp8 stages is slower than 2-deep (3.61 vs 3.85 TB/s on gate/up)
    WReg wb[DEPTH][8];
#pragma unroll
    for (int f = 0; f < DEPTH - 1; ++f)
      if (f / NSU < cnt) issue_w(wb[f % DEPTH], pass, wave + NW * (f / NSU), f % NSU);
    for (int i = 0; i < cnt; i += DEPTH) {
#pragma unroll
      for (int h = 0; h < DEPTH; ++h) {
        const int sl = wave + NW * (i + h);
        if (i + h < cnt) {
#pragma unroll
          for (int su = 0; su < NSU; ++su) {
            const int cur = (h * NSU + su) % DEPTH;
            const int fn = h * NSU + su + DEPTH - 1;  // stage to prefetch, relative to this trip
            if (i + fn / NSU < cnt) issue_w(wb[fn % DEPTH], pass, wave + NW * (i + fn / NSU), fn % NSU);
            if (su == 0) {
              stage_x(sl);
              if (i + h + 1 < cnt) load_x(sl + NW);
              else if (pass + 1 < npass) load_x(wave);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(wst + (2 * j + lrow) * WROWB + lchunk * 16) = widen(wb[cur][j]);
            __builtin_amdgcn_wave_barrier();
            // x fragments are re-read per sub-unit rather than held: 32 VGPRs buy nothing, LDS has the headroom
            const int xrow = lane & (2 * NI - 1);  // rows past the staged ones alias valid rows: their outputs are never stored
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xst + xrow * WROWB + (4 * s + (lane >> 4)) * 16);
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wst + (lane & 15) * WROWB + (4 * s + (lane >> 4)) * 16);
              // D[batch row][weight row] += x[batch row][k] * W[weight row][k]
              acc[su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf, acc[su], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    }

    // ---- cross-wave reduction (fixed order) + epilogue ----
    __syncthreads();
    float* redf = reinterpret_cast<float*>(smem);  // [NW waves][MAXSU][64 lanes][4]
#pragma unroll
    for (int su = 0; su < NSU; ++su) *reinterpret_cast<f32x4*>(redf + ((wave * MAXSU + su) * 64 + lane) * 4) = acc[su];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
      const int e = tid + i * NT;
      if (e >= (NSU / R) * 256) break;
      const int u = e >> 8, l2 = e & 63, q = (e >> 6) & 3;
      const int b = 4 * (l2 >> 4) + q;
      const int unit = (pass * MAXU + u) * grid + (int)blockIdx.x;
      const int n = unit * 16 + (l2 & 15);
      float a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) t += redf[((wv * MAXSU + u * R + r) * 64 + l2) * 4 + q];
        if (W8) t *= pre_sc[i][r];
        a[r] = t;
      }
      if (b < B && n < N) {
        if (SWIGLU) {
          const float g = rnd<bf16_t>(a[0]), up = rnd<bf16_t>(a[R - 1]);
          reinterpret_cast<bf16_t*>(out)[(size_t)b * N + n] = (bf16_t)(rnd<bf16_t>(silu(g)) * up);
        } else {
          float v = rnd<bf16_t>(a[0]);
          if (residual) v = rnd<bf16_t>(pre_res[i] + v);
          if (out_f32)
            reinterpret_cast<float*>(out)[(size_t)b * N + n] = v;
          else
            reinterpret_cast<bf16_t*>(out)[(size_t)b * N + n] = (bf16_t)v;
        }
      }
    }
    __syncthreads();  // the reduction buffer aliases the wave-private stages of the next pass
  };

  if (do_norm) rms_stats();
  for (int pass = 0; pass < npass; ++pass) {
    int nu = 0;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) nu += ((pass * MAXU + u) * grid + (int)blockIdx.x < NU) ? 1 : 0;
    if (nu == 0) break;  // uniform per block; later passes are empty too
    switch (nu * R) {
      case 1: run_pass(std::integral_constant<int, 1>{}, pass, nu); break;
      case 2: run_pass(std::integral_constant<int, 2>{}, pass, nu); break;
      case 3: run_pass(std::integral_constant<int, 3>{}, pass, nu); break;
      default: run_pass(std::integral_constant<int, 4>{}, pass, nu); break;
    }
  }
}

template <bool SWIGLU, int NI, int NW, bool W8>
int launch_skinny(const void* x, const void* W, const float* wscale, const void* norm_w, float eps, const void* residual,
                  void* out, int batch, int N, int K, int out_f32, int grid, hipStream_t s) {
  constexpr int lds = NW * (2 * NI * WROWB + WSTAGEB);
  static_assert(lds >= NW * MAXSU * 64 * 4 * 4, "reduction buffer must fit");
  static_assert(lds <= 160 * 1024, "LDS");
  auto kfn = skinny_kernel<SWIGLU, NI, NW, W8>;
  static std::atomic<uint64_t> attr_done{0};
  SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)kfn, lds));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * NW), lds, s, (const bf16_t*)x, W, wscale, (const bf16_t*)norm_w, eps,
                     (const bf16_t*)residual, out, batch, N, K, out_f32);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

template <bool SWIGLU, int NI, bool W8>
int launch_skinny_nw(const void* x, const void* W, const float* wscale, const void* norm_w, float eps, const void* residual,
                     void* out, int batch, int N, int K, int out_f32, hipStream_t s) {
  const int cus = srgpt_device_cus();
  const int NU = (N + 15) / 16;
  if (NU <= cus) return launch_skinny<SWIGLU, NI, 8, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, NU, s);
  // balanced grid: every block gets the same number of units whenever N allows (e.g. 896 SwiGLU units -> 448 blocks x 2)
  const int maxgrid = cus * 2;
  const int per = (NU + maxgrid - 1) / maxgrid;  // units per block (more than MAXSU/R -> several passes in the kernel)
  const int grid = (NU + per - 1) / per;
  return launch_skinny<SWIGLU, NI, 4, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, grid, s);
}

template <bool W8>
int skinny_dispatch(const void* x, const void* W, const float* wscale, const void* norm_w, float eps, const void* residual,
                    void* out, int batch, int N, int K, int swiglu, int out_f32, hipStream_t s) {
  SRGPT_CHECK(batch >= 1 && batch <= 16, SRGPT_ERR_ARG, "skinny: batch %d outside 1..16", batch);
  SRGPT_CHECK(K % 8 == 0 && K >= 8, SRGPT_ERR_ARG, "skinny: K=%d must be a multiple of 8", K);
  if (batch <= 4)
    return swiglu ? launch_skinny_nw<true, 2, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, s)
                  : launch_skinny_nw<false, 2, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, s);
  if (batch <= 8)
    return swiglu ? launch_skinny_nw<true, 4, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, s)
                  : launch_skinny_nw<false, 4, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, s);
  return swiglu ? launch_skinny_nw<true, 8, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, s)
                : launch_skinny_nw<false, 8, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, s);
}

}  // namespace

// host entry used by srgpt_gemv (gemv.hip) for batches of up to 16 rows, bf16 weights
int srgpt_skinny_launch(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out,
                        int batch, int N, int K, int swiglu, int out_f32, hipStream_t s) {
  return skinny_dispatch<false>(x, W, nullptr, norm_w, eps, residual, out, batch, N, K, swiglu, out_f32, s);
}

// Decode-path product with fp8 (OCP e4m3fn) weights and one fp32 scale per weight row, bf16 activations (W8A16):
// out[b, n] = bf16( (sum_k x[b, k] * fp8(W8[n, k])) * wscale[n] ), same fusions as srgpt_gemv.  Any batch size
// (16 rows per weight pass).
extern "C" int srgpt_gemv_w8(const void* x, const void* W8, const float* wscale, const void* norm_w, float norm_eps,
                             const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32,
                             srgpt_stream_t stream) {
  SRGPT_CHECK(x && W8 && wscale && out, SRGPT_ERR_ARG, "srgpt_gemv_w8: null pointer");
  SRGPT_CHECK(N > 0 && K > 0 && batch > 0, SRGPT_ERR_ARG, "srgpt_gemv_w8: bad shape");
  SRGPT_CHECK(K % 8 == 0, SRGPT_ERR_ARG, "srgpt_gemv_w8: K=%d must be a multiple of 8", K);
  SRGPT_CHECK(!(swiglu && (residual || out_f32)), SRGPT_ERR_ARG, "srgpt_gemv_w8: swiglu excludes residual/out_f32");
  hipStream_t s = as_stream(stream);
  const int valu_max = SRGPT_KNOB("SRGPT_W8_VALU_MAX_BATCH", 2);  // tuning knob
  if (batch <= valu_max && batch <= 2 && K % 16 == 0)  // 1-2 rows: VALU kernel (gemv_w8.hip), like the bf16 path
    return srgpt_gemv_w8_valu(x, W8, wscale, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, s);
  const size_t on = out_f32 ? sizeof(float) : 2;
  for (int b0 = 0; b0 < batch; b0 += 16) {
    const int nb = batch - b0 < 16 ? batch - b0 : 16;
    SRGPT_TRY(skinny_dispatch<true>((const char*)x + (size_t)b0 * K * 2, W8, wscale, norm_w, norm_eps,
                                    residual ? (const char*)residual + (size_t)b0 * N * 2 : nullptr,
                                    (char*)out + (size_t)b0 * N * on, nb, N, K, swiglu, out_f32, s));
  }
  return SRGPT_OK;
}
