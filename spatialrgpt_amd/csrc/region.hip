// Region extractor kernels (SpatialRGPT specific, reference: llava/model/region_extractor/base_extractor.py)
// plus the layout kernels either side of it (projector space-to-depth, adaptive average pool, patch im2col).
//
// Masked region pooling is HBM-bound: the refined feature map (108*108 x C bf16 = 26.9 MB at C = 1152) is
// read exactly once for ALL M masks.  Two launches (round 3; three before), no float atomics, every sum in a fixed order:
//   1. region_resample_kernel : grid (position slabs, masks), one thread per feature position: bilinear resample of the mask
//      to the feature grid (PyTorch's upsample_bilinear2d index math, align_corners = False, no antialias), rounded to the
//      feature dtype -> v[M][L]; the block's fixed-order partial sum -> psum[M][slabs].  Also re-arms launch 2's tickets.
//   2. region_pool_kernel     : grid (row slabs x channel slabs).  Prologue: denominators from psum (fixed order, + 1e-8, the
//      feature dtype's roundings), the slab's L1-normalised weights rnd(v / denorm) into LDS.  Main loop: every thread
//      streams 16-byte channel chunks of its rows, 8 independent loads in flight, M x 8 fp32 accumulators; in-block
//      fixed-order reduce (cross-lane inside the wave, LDS across waves); the slab's partial is published write-through and an
//      arrival ticket per channel slab drawn: the block that arrives last sums the row slabs' partials IN SLAB ORDER, rounds
//      and stores (the result does not depend on which block came last).
#include "common.h"

namespace {

constexpr int RP_CL = 8;                  // 16-byte channel chunks per block (one 128-byte line per row)
constexpr int RP_RG = 30;                 // row groups per block (8 * 30 = 240 of 256 threads)
constexpr int RP_RPT = 10;                // rows per thread: independent 16-byte loads in flight
constexpr int RP_ROWS = RP_RG * RP_RPT;   // 300 feature rows per block: 39 x 18 = 702 blocks at 108 x 108 x 1152 -- all resident at
                                          // 3 blocks per CU (168 registers: at 128 the accumulators spilled and the reduce crawled)
constexpr int RP_POS = 1024;              // positions per block of the resample launch
constexpr int RP_MAXM = 16;               // masks per launch
constexpr int RP_FB = 16;                 // partials the last arriver keeps in flight (25 cost registers: 3 blocks per CU instead of 4)

struct Idx {
  int i0, i1;
  float l0, l1;
};
// PyTorch area_pixel_compute_source_index + guard_index_and_lambda (ATen/native/UpSample.h)
__device__ __forceinline__ Idx src_index(float rscale, int dst, int n_in) {
  float src = rscale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Idx r;
  r.i0 = min((int)src, n_in - 1);
  r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
  r.l0 = 1.f - r.l1;
  r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
  return r;
}

// RAW = true: `masks` are the caller's raw uint8 [M, rh, rw] masks and (ys, xs) the cv2.INTER_NEAREST source
// index of every row / column of the mh x mw processor-size mask the reference would have built on the host
// (mm_utils.py:477-532: cv2.resize(..., INTER_NEAREST), then the processor with rescale 1.0 -> float(uint8)); the bilinear taps
// read THROUGH the tables, so the nearest resize, the float conversion and the resample are one pass over the raw bytes.
template <typename T, typename MT, bool RAW>
__global__ __launch_bounds__(RP_POS) void region_resample_kernel(const MT* __restrict__ masks, float* __restrict__ v,
                                                                 float* __restrict__ psum, int* __restrict__ tickets,
                                                                 int n_tickets, int mh, int mw, int fw, float rscale_h,
                                                                 float rscale_w, const int* __restrict__ ys,
                                                                 const int* __restrict__ xs, int rh, int rw) {
  __shared__ float red[16];
  const int m = blockIdx.y, slab = blockIdx.x;
  const MT* mk = masks + (size_t)m * (RAW ? (size_t)rh * rw : (size_t)mh * mw);
  const int L = fw * fw;
  if (m == 0 && slab == 0)
    for (int i = threadIdx.x; i < n_tickets; i += blockDim.x) tickets[i] = 0;  // launch 2 runs behind this one on the stream
  auto at = [&](int yy, int xx) -> float {
    if constexpr (RAW)
      return (float)mk[(size_t)ys[yy] * rw + xs[xx]];  // float(uint8), as the processor with rescale_factor 1.0 produces
    else
      return to_f(mk[(size_t)yy * mw + xx]);
  };
  const int l = slab * RP_POS + threadIdx.x;
  float val = 0.f;
  if (l < L) {
    const int oy = l / fw, ox = l - oy * fw;
    const Idx y = src_index(rscale_h, oy, mh), x = src_index(rscale_w, ox, mw);
    const float v00 = at(y.i0, x.i0), v01 = at(y.i0, x.i1);
    const float v10 = at(y.i1, x.i0), v11 = at(y.i1, x.i1);
    // explicit fused steps: every instantiation (float / bf16 / raw uint8 masks) rounds the same way, whatever the
    // compiler's contraction choices would have been
    const float top = fmaf(x.l1, v01, x.l0 * v00), bot = fmaf(x.l1, v11, x.l0 * v10);
    val = rnd<T>(fmaf(y.l1, bot, y.l0 * top));  // .to(x.dtype)
    v[(size_t)m * L + l] = val;
  }
  const float t = block_sum(val, red);
  if (threadIdx.x == 0) psum[(size_t)m * gridDim.x + slab] = t;
}

#ifdef SRGPT_TUNING_KNOBS
// phase stamps (tuning build; scripts/experiments/ubench_region_stamps.py): block (0, 0) -> slots 0..15, the last arriver of channel slab 0 -> 16..
__device__ unsigned long long srgpt_region_stamps[32];
#define RP_STAMP(i) do { if (stamp_base >= 0 && threadIdx.x == 0) srgpt_region_stamps[stamp_base + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RP_STAMP(i) do { } while (0)
#endif

template <typename T, int MM>
__global__ __launch_bounds__(256, MM <= 8 ? 3 : 2) void region_pool_kernel(const T* __restrict__ feat, const float* __restrict__ v,
                                                          const float* __restrict__ psum, int n_psum,
                                                          float* __restrict__ partial, int* __restrict__ tickets,
                                                          T* __restrict__ out, int M, int L, int C) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int CW = RP_CL * VEC;  // channels per block
  constexpr int RP_UN = MM <= 8 ? RP_RPT : RP_RPT / 2, RP_NB = RP_RPT / RP_UN;  // 16 masks: 128 accumulators, two batches of 5 rows
  // one LDS array, two lives: the slab's weights [MM][RP_ROWS] during the FMAs, then the reduce staging [16][MM * VEC][RP_CL]
  constexpr int LDS_BIG = (MM * RP_ROWS > 16 * MM * VEC * RP_CL) ? MM * RP_ROWS : 16 * MM * VEC * RP_CL;
  __shared__ float lds_big[LDS_BIG];
  // gfx950 only (the library's one target, Makefile ARCH): the 16-mask instance declares ~72 KB of static LDS, which fits the 160 KB of
  // a CDNA4 CU and no 64 KB-LDS part; say so at compile time instead of failing inside the assembler of another ARCH
  static_assert(sizeof(float) * (LDS_BIG + MM + MM * 32 + MM * CW) + sizeof(int) <= 160 * 1024,
                "region_pool_kernel: static LDS exceeds the 160 KB of a gfx950 CU");
  float (*wsm)[RP_ROWS] = reinterpret_cast<float (*)[RP_ROWS]>(lds_big);
  float* rstage = lds_big;
  __shared__ float den[MM];
  __shared__ float psl[MM][32];
  __shared__ float red[MM][CW];
  __shared__ int last;
  // channel slab = the FAST grid index: the blocks that run together cover whole feature rows (contiguous 2304-byte rows, open
  // DRAM pages used in full) instead of one 128-byte line out of every row
  const int slab = blockIdx.y, l0 = slab * RP_ROWS, nslab = gridDim.y, cslab = blockIdx.x;
  const int nrows = min(RP_ROWS, L - l0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = tid % RP_CL, rg = tid / RP_CL;
  const int chunk = cslab * RP_CL + cl;
  const bool active = rg < RP_RG && chunk * VEC < C;
#ifdef SRGPT_TUNING_KNOBS
  int stamp_base = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : -1;
#endif
  RP_STAMP(0);
  // ---- the feature rows of the first batch go out before the prologue touches anything else (row index clamped, never branched)
  const T* fbase = feat + (size_t)(active ? chunk : 0) * VEC;
  Vec16<T> f[RP_UN];
  auto issue = [&](int batch) {
#pragma unroll
    for (int u = 0; u < RP_UN; ++u) {
      const int r = min(l0 + (batch * RP_UN + u) * RP_RG + min(rg, RP_RG - 1), L - 1);
      f[u] = *reinterpret_cast<const Vec16<T>*>(fbase + (size_t)r * C);
    }
  };
  issue(0);
  RP_STAMP(1);
  // ---- prologue: mask.sum() + 1e-8 in the feature dtype (fixed slab order), then the slab's normalised weights ----
  // (the resample launch's per-slab sums are fetched by 32 threads per mask at once and added in slab order from LDS: a chain of
  //  n_psum dependent loads in one thread was ~6 us of every block's prologue)
  float sden = 0.f;
  for (int i0 = 0; i0 < n_psum; i0 += 32) {
    for (int idx = tid; idx < MM * 32; idx += 256) {
      const int m = idx >> 5, i = i0 + (idx & 31);
      psl[m][idx & 31] = (m < M && i < n_psum) ? psum[(size_t)m * n_psum + i] : 0.f;
    }
    __syncthreads();
    if (tid < MM) {
      const int n = min(32, n_psum - i0);
      for (int i = 0; i < n; ++i) sden += psl[tid][i];
    }
    __syncthreads();
  }
  if (tid < MM) den[tid] = rnd<T>(rnd<T>(sden) + 1e-8f);
  __syncthreads();
  RP_STAMP(2);
  {  // mask / denorm in the feature dtype; the thread's v values are requested together (clamped, unconditional), then divided
    constexpr int NW = (MM * RP_ROWS + 255) / 256;
    float vv[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int i = min(tid + 256 * q, MM * RP_ROWS - 1);
      const int m = i / RP_ROWS, r = i - m * RP_ROWS;
      vv[q] = v[(size_t)min(m, M - 1) * L + min(l0 + r, L - 1)];
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int i = tid + 256 * q;
      if (i < MM * RP_ROWS) {
        const int m = i / RP_ROWS, r = i - m * RP_ROWS;
        wsm[m][r] = (m < M && r < nrows) ? rnd<T>(vv[q] / den[m]) : 0.f;
      }
    }
  }
  __syncthreads();
  RP_STAMP(3);
  float acc[MM][VEC];
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[m][i] = 0.f;
#pragma unroll
  for (int b = 0; b < RP_NB; ++b) {
#pragma unroll
    for (int u = 0; u < RP_UN; ++u) {
      const int r = (b * RP_UN + u) * RP_RG + min(rg, RP_RG - 1);  // rows past the slab's end carry weight 0
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const float wv = wsm[m][r];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[m][i] = fmaf(wv, f[u].get(i), acc[m][i]);
      }
    }
    if (b + 1 < RP_NB) issue(b + 1);
  }
  RP_STAMP(4);
  // ---- fixed-order reduce over the block's 30 row groups: the two row groups of a 16-lane row on the VALU (one DPP rotate), the
  //      16 (wave, row) partials through LDS in ONE round -- the staging array takes over the weights' LDS (dead by now).
  //      (The same sums as cross-lane v_permlane swaps took 12k cycles for 64 values; two masks per LDS round 6.4k.)
  constexpr int NVAL = MM * VEC;                 // values per thread
  __syncthreads();  // every wave is done reading the weights: the array becomes the reduce staging
  {
    const int row16 = lane >> 4, l16 = lane & 15;
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float a = (rg < RP_RG) ? acc[m][i] : 0.f;
        a += dpp_mov<0x128>(a);  // row_ror:8: lanes l and l + 8 of a 16-lane row = the same channel chunk, two row groups
        if (l16 < RP_CL) rstage[((wave * 4 + row16) * NVAL + m * VEC + i) * RP_CL + l16] = a;
      }
  }
  __syncthreads();
  for (int o = tid; o < NVAL * RP_CL; o += 256) {
    const int j = o / RP_CL, c8 = o - j * RP_CL;  // value j = (mask j / VEC, element j % VEC) of channel chunk c8
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += rstage[(q * NVAL + j) * RP_CL + c8];  // fixed (wave, row) order
    red[j / VEC][c8 * VEC + (j % VEC)] = t;
  }
  __syncthreads();
  RP_STAMP(5);
  const int c0 = cslab * CW;
  // partials go out WRITE-THROUGH in 16-byte pieces (sc1 buffer stores: a 4-byte write-through store is one fabric write each,
  // ~6x the time per byte) -- whichever block of this channel slab arrives last reads every slab's partial with sc1 loads
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(partial, 0, (int)((size_t)nslab * M * C * sizeof(float)), 0x00020000);
  constexpr int CW4 = CW / 4;
  for (int j = tid; j < M * CW4; j += 256) {
    const int m = j / CW4, c = 4 * (j - m * CW4);
    if (c0 + c < C) {  // C % 4 == 0: a group of 4 channels is entirely in or out
      u32x4 t;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        t[q] = __float_as_uint(red[m][c + q]);
      __builtin_amdgcn_raw_buffer_store_b128(t, prs, (int)((((size_t)slab * M + m) * C + c0 + c) * sizeof(float)), 0, 16);
    }
  }
  // ---- arrival ticket of the channel slab; the last arriver sums the row slabs in slab order and stores ----
  RP_STAMP(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  RP_STAMP(7);
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(tickets + cslab, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == nslab - 1) ? 1 : 0;
  }
  __syncthreads();
  RP_STAMP(8);
  if (!last) return;
#ifdef SRGPT_TUNING_KNOBS
  if (cslab == 0) stamp_base = 16;
#endif
  RP_STAMP(0);
  for (int j = tid; j < M * CW4; j += 256) {
    const int m = j / CW4, c = 4 * (j - m * CW4);
    if (c0 + c < C) {
      // slab order, RP_FB loads in flight at a time (a chain of dependent L2-bypassing loads cost ~1 us per slab)
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < nslab; s0 += RP_FB) {
        u32x4 pv[RP_FB];
#pragma unroll
        for (int q = 0; q < RP_FB; ++q)
          pv[q] = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)((((size_t)min(s0 + q, nslab - 1) * M + m) * C + c0 + c) * sizeof(float)),
                                                        0, 16);
#pragma unroll
        for (int q = 0; q < RP_FB; ++q)
          if (s0 + q < nslab) {
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += __uint_as_float(pv[q][e]);
          }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(size_t)m * C + c0 + c + e] = from_f<T>(t[e]);
    }
  }
  RP_STAMP(1);
  if (tid == 0) __hip_atomic_store(tickets + cslab, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// Round 4: the pooling launch on the matrix pipe (bf16 features).  out[16 masks][C] = Wn[16][L] . F[L][C] is a skinny GEMM whose
// contraction runs over POSITIONS -- the strided axis of the channels-last map -- so the B operand of v_mfma_f32_16x16x32_bf16 (lane
// (channel, k group) holds 8 consecutive positions of ONE channel) needs a transpose.  A wave streams 32 positions x 64 channels
// per chunk (4 KB, four coalesced 8 x 128-byte loads), parks the chunk row-major in a PRIVATE 4.5-KB LDS region (row stride 144 B)
// and reads its fragments back with the transposing LDS read (ds_read_b64_tr_b16: a 16-lane group addresses a [4 positions] x
// [16 channels] block, 8 bytes per lane, and every lane receives one channel's 4 positions -- two reads per fragment).  The A
// operand -- the slab's normalised weights rnd(v / denorm), exact in bf16 -- comes from the block's weight array (row stride
// ROWS + 16 elements: conflict-free ds_read_b128).  Four accumulators (16 masks x 16 channels each) per wave, no cross-lane
// reduction, 16 masks for the price of 8.
// Requests go out in the order their consumers run (mask sums, the slab's resampled values, then ALL feature rows of the wave):
// loads return in order, so the prologue waits for the two small sets only while the features stream in behind them.  Denominators
// are summed per thread (16 threads per mask, every one of them the same slab order: no exchange, no barrier).
// Partial publication, arrival ticket and the slab-order final sum are the VALU kernel's; the VALU kernel stays for fp32 features
// (the 2e-6 KATs).  108 x 108 x 1152, 8 / 16 masks: 18.6 / 31.9 us -> 11.2 / 11.3 us (profiles/r04_region_pooling.txt, which also
// holds the ablations: without the ticket + last-arriver tail 7.8 us, without the feature loads 11.3 us -- the three dependent
// memory round trips of the deterministic cross-block sum, not the 26.9-MB stream, are what is left).
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int CH>  // 32-position chunks per wave: a block owns 4 * CH * 32 positions x 64 channels
__global__ __launch_bounds__(256, 2) void region_pool_mfma_kernel(const bf16_t* __restrict__ feat, const float* __restrict__ v,
                                                                  const float* __restrict__ psum, int n_psum,
                                                                  float* __restrict__ partial, int* __restrict__ tickets,
                                                                  bf16_t* __restrict__ out, int M, int L, int C) {
  constexpr int ROWS = 4 * CH * 32;
  constexpr int WLD = ROWS + 16;  // weight row stride (elements): = 16 mod 32 -> the 16-lane groups of ds_read_b128 cover all banks
  constexpr int TLD = 72;         // transposed-staging row stride (elements) = 144 bytes
  constexpr int CW = 64;
  constexpr int FB = 24;  // partials the last arriver keeps in flight: the 23 row slabs of 108 x 108 at CH = 4 in one round trip
  __shared__ __attribute__((aligned(16))) bf16_t wlds[16 * WLD];
  __shared__ __attribute__((aligned(16))) bf16_t tlds[4][32 * TLD];
  __shared__ __attribute__((aligned(16))) float red[4][16][CW];
  __shared__ int last;
  const int slab = blockIdx.y, l0 = slab * ROWS, nslab = gridDim.y, cslab = blockIdx.x, c0 = cslab * CW;
  const int nrows = min(ROWS, L - l0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SRGPT_TUNING_KNOBS
  int stamp_base = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : -1;
#endif
  RP_STAMP(0);
  // ---- requests in the order their consumers run: mask sums, the slab's resampled values, then every feature row of the wave.
  // Loads return in order, so the prologue waits only for the two small sets while the features stream in behind them.
  const int wm = tid >> 4, wr = tid & 15, wmc = min(wm, M - 1);  // weights: 16 threads per mask, positions wr + 16 j
  float ps[16], vv[8 * CH];
#pragma unroll
  for (int q = 0; q < 16; ++q) ps[q] = psum[(size_t)wmc * n_psum + min(q, n_psum - 1)];
#pragma unroll
  for (int j = 0; j < 8 * CH; ++j) vv[j] = v[(size_t)wmc * L + min(l0 + wr + 16 * j, L - 1)];
  const int lrow = lane >> 3, cch = min(c0 + (lane & 7) * 8, C - 8);
  u32x4 f[CH][4];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(l0 + (wave * CH + ch) * 32 + 8 * i + lrow, L - 1);
      f[ch][i] = *reinterpret_cast<const u32x4*>(feat + (size_t)r * C + cch);
    }
  RP_STAMP(1);
  // ---- mask.sum() + 1e-8 in the feature dtype: the psum slabs in slab order (every thread of the mask's group, no exchange) ----
  float sden = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q)
    if (q < n_psum) sden += ps[q];
  for (int i0 = 16; i0 < n_psum; i0 += 16) {  // maps beyond 128 x 128
#pragma unroll
    for (int q = 0; q < 16; ++q) ps[q] = psum[(size_t)wmc * n_psum + min(i0 + q, n_psum - 1)];
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (i0 + q < n_psum) sden += ps[q];
  }
  const float den = rnd<bf16_t>(rnd<bf16_t>(sden) + 1e-8f);
  RP_STAMP(2);
  // ---- the slab's normalised weights rnd(v / denorm), bf16-exact; masks past M and rows past the slab weigh 0 ----
#pragma unroll
  for (int j = 0; j < 8 * CH; ++j) {
    const int r = wr + 16 * j;
    wlds[wm * WLD + r] = (bf16_t)((wm < M && r < nrows) ? rnd<bf16_t>(vv[j] / den) : 0.f);
  }
  __syncthreads();
  RP_STAMP(3);
  // ---- main: per chunk, park row-major, transpose-read the B fragments (ds_read_b64_tr_b16: a 16-lane group reads a
  // [4 positions][16 channels] block and every lane receives one channel's 4 positions), four MFMAs ----
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16_t* tl = tlds[wave];
  const int n16 = lane & 15, g4 = lane >> 4;
  const bf16_t* trp = tl + (8 * g4 + (n16 >> 2)) * TLD + 4 * (n16 & 3);
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tl + (8 * i + lrow) * TLD + (lane & 7) * 8) = f[ch][i];
    __builtin_amdgcn_wave_barrier();
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(wlds + n16 * WLD + (wave * CH + ch) * 32 + g4 * 8);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(trp + 16 * t));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(trp + 16 * t + 4 * TLD));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, both), acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  RP_STAMP(4);
  // ---- the four waves' partial sums, fixed wave order ----
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][4 * g4 + q][16 * t + n16] = acc[t][q];
  __syncthreads();
  RP_STAMP(5);
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(partial, 0, (int)((size_t)nslab * M * C * sizeof(float)), 0x00020000);
  {
    const int m = tid >> 4, c = 4 * (tid & 15);
    if (m < M && c0 + c < C) {  // C % 8 == 0: a group of 4 channels is entirely in or out
      u32x4 t;
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = __float_as_uint(((red[0][m][c + q] + red[1][m][c + q]) + red[2][m][c + q]) + red[3][m][c + q]);
      __builtin_amdgcn_raw_buffer_store_b128(t, prs, (int)((((size_t)slab * M + m) * C + c0 + c) * sizeof(float)), 0, 16);
    }
  }
  RP_STAMP(6);
  // ---- arrival ticket of the channel slab; the last arriver sums the row slabs in slab order and stores ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  RP_STAMP(7);
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(tickets + cslab, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == nslab - 1) ? 1 : 0;
  }
  __syncthreads();
  RP_STAMP(8);
  if (!last) return;
#ifdef SRGPT_TUNING_KNOBS
  if (cslab == 0) stamp_base = 16;
#endif
  RP_STAMP(0);
  {
    const int m = tid >> 4, c = 4 * (tid & 15);
    if (m < M && c0 + c < C) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < nslab; s0 += FB) {
        u32x4 pv[FB];
#pragma unroll
        for (int q = 0; q < FB; ++q)
          pv[q] = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)((((size_t)min(s0 + q, nslab - 1) * M + m) * C + c0 + c) * sizeof(float)),
                                                        0, 16);
#pragma unroll
        for (int q = 0; q < FB; ++q)
          if (s0 + q < nslab) {
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += __uint_as_float(pv[q][e]);
          }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(size_t)m * C + c0 + c + e] = (bf16_t)t[e];
    }
  }
  RP_STAMP(1);
  if (tid == 0) __hip_atomic_store(tickets + cslab, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// AdaptiveAvgPool2d(out_w) on a channels-last [n, in_w, in_w, C] map -> [n, out_w*out_w, C]
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int in_w, int out_w, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int op = blockIdx.x, img = blockIdx.y;
  const int oy = op / out_w, ox = op - oy * out_w;
  const int y0 = (oy * in_w) / out_w, y1 = ((oy + 1) * in_w + out_w - 1) / out_w;
  const int x0 = (ox * in_w) / out_w, x1 = ((ox + 1) * in_w + out_w - 1) / out_w;
  const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
  const T* xi = x + (size_t)img * in_w * in_w * C;
  for (int c = threadIdx.x; c < C / VEC; c += blockDim.x) {
    float s[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(xi + ((size_t)yy * in_w + xx) * C + (size_t)c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] += v.get(i);
      }
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, s[i] * inv);
    *reinterpret_cast<Vec16<T>*>(y + ((size_t)img * out_w * out_w + op) * C + (size_t)c * VEC) = o;
  }
}

// DownSampleBlock: out token o = bc * hb + br, channels [x(2br,2bc) | x(2br,2bc+1) | x(2br+1,2bc) | x(2br+1,2bc+1)],
// x(r,c) = input token r*g + c, zero outside the g x g grid (verified against the reference).
template <typename T>
__global__ void s2d_kernel(const T* __restrict__ x, T* __restrict__ y, int g, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int hb = (g + 1) >> 1;
  const int o = blockIdx.x, img = blockIdx.y;
  const int br = o % hb, bc = o / hb;
  const int cpq = C / VEC;  // chunks per quadrant
  for (int c = threadIdx.x; c < 4 * cpq; c += blockDim.x) {
    const int quad = c / cpq, cc = c - quad * cpq;
    const int r = 2 * br + (quad >> 1), col = 2 * bc + (quad & 1);
    u32x4 v = {0, 0, 0, 0};
    if (r < g && col < g)
      v = *reinterpret_cast<const u32x4*>(x + ((size_t)img * g * g + (size_t)r * g + col) * C + (size_t)cc * VEC);
    *reinterpret_cast<u32x4*>(y + ((size_t)img * hb * hb + o) * 4 * C + (size_t)c * VEC) = v;
  }
}

// patch im2col for Conv2d(3, C, k = s = p, 'valid'): row (img, py, px) <- pixels ordered (c, ky, kx), zero pad to kp
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ img, T* __restrict__ out, int S, int p, int kp) {
  const int gw = S / p;
  const int patch = blockIdx.x, n = blockIdx.y;
  const int py = patch / gw, px = patch - py * gw;
  T* o = out + ((size_t)n * gw * gw + patch) * kp;
  const int kk = 3 * p * p;
  for (int i = threadIdx.x; i < kp; i += blockDim.x) {
    T v = from_f<T>(0.f);
    if (i < kk) {
      const int c = i / (p * p), rem = i - c * p * p, ky = rem / p, kx = rem - ky * p;
      v = img[(((size_t)n * 3 + c) * S + (py * p + ky)) * S + (px * p + kx)];
    }
    o[i] = v;
  }
}

// CLIP embeddings: [cls | patches] + position embedding, one block per output token
template <typename T>
__global__ void vit_assemble_cls_kernel(const T* __restrict__ patches, const T* __restrict__ cls, const T* __restrict__ pos,
                                        T* __restrict__ x, int gg, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int t = blockIdx.x, img = blockIdx.y;
  const T* src = (t == 0) ? cls : patches + ((size_t)img * gg + (t - 1)) * C;
  const T* pe = pos + (size_t)t * C;
  T* dst = x + ((size_t)img * (gg + 1) + t) * C;
  for (int c = threadIdx.x; c < C / VEC; c += blockDim.x) {
    const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(src + c * VEC);
    const Vec16<T> b = *reinterpret_cast<const Vec16<T>*>(pe + c * VEC);
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, a.get(i) + b.get(i));
    *reinterpret_cast<Vec16<T>*>(dst + c * VEC) = o;
  }
}

}  // namespace

// workspace layout (floats): v [M, L] | psum [M, ceil(L / 1024)] | partial [row slabs, M, C] | tickets (ints) [channel slabs]
struct RegionWs {
  size_t v, psum, partial, tickets, total;
  int n_psum, nslab, nslab_cap, ncslab_max;
};
// 32-position chunks per wave of region_pool_mfma_kernel by map size: 4 (512 positions per block: 23 row slabs x 18 channel slabs at
// 108^2 x 1152) / 2 / 1 -- measured per size in profiles/r04_region_pooling.txt (108^2: 26.3 / 15.2 / 11.2 us at 1 / 2 / 4;
// 27^2: 5.2 / 6.0 / 7.5)
static inline int region_mfma_chunks(size_t L) { return L >= 8192 ? 4 : (L >= 2048 ? 2 : 1); }
static RegionWs region_ws(int M, int fw, int C) {
  RegionWs r;
  const size_t L = (size_t)fw * fw;
  r.n_psum = (int)((L + RP_POS - 1) / RP_POS);
  r.nslab = (int)((L + RP_ROWS - 1) / RP_ROWS);
  r.ncslab_max = (C / 4 + RP_CL - 1) / RP_CL;  // fp32 features: 4 channels per 16-byte chunk (bf16 needs half as many)
  r.v = 0;
  r.psum = r.v + (size_t)M * L;
  r.partial = (r.psum + (size_t)M * r.n_psum + 3) & ~(size_t)3;  // 16-byte pieces
  const size_t nslab_mfma = (L + 128 * region_mfma_chunks(L) - 1) / (128 * region_mfma_chunks(L));  // row slabs of the MFMA pooling kernel
  r.nslab_cap = (int)(nslab_mfma > (size_t)r.nslab ? nslab_mfma : (size_t)r.nslab);
  r.tickets = r.partial + (size_t)r.nslab_cap * M * C;
  r.total = r.tickets + (size_t)r.ncslab_max;
  return r;
}

#ifdef SRGPT_TUNING_KNOBS
extern "C" int srgpt_region_debug_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(srgpt_region_stamps), sizeof(unsigned long long) * (n < 32 ? n : 32));
}
#endif

extern "C" int64_t srgpt_region_pool_ws_floats(int M, int fw, int C) {
  if (M <= 0 || fw <= 0 || C <= 0) return -1;
  return (int64_t)region_ws(M, fw, C).total;
}

static int region_pool_impl(const void* feat, const void* masks, void* out, float* ws, int M, int mh, int mw, int fw, int C,
                            float rscale_h, float rscale_w, int mask_dtype, int dtype, const int* ys, const int* xs, int rh, int rw,
                            srgpt_stream_t stream) {
  const bool raw = ys != nullptr;
  SRGPT_CHECK(raw || mask_dtype == SRGPT_BF16 || mask_dtype == SRGPT_F32, SRGPT_ERR_ARG, "srgpt_region_pool: bad mask dtype");
  SRGPT_CHECK(feat && masks && out && ws, SRGPT_ERR_ARG, "srgpt_region_pool: null pointer");
  SRGPT_CHECK(M > 0 && M <= RP_MAXM, SRGPT_ERR_ARG, "srgpt_region_pool: M=%d must be in 1..%d per call", M, RP_MAXM);
  SRGPT_CHECK(mh > 0 && mw > 0 && fw > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_region_pool: bad shape");
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(C % vec == 0, SRGPT_ERR_ARG, "srgpt_region_pool: C=%d must be a multiple of %d", C, vec);
  SRGPT_CHECK(dtype == SRGPT_BF16 || dtype == SRGPT_F32, SRGPT_ERR_ARG, "srgpt_region_pool: bad dtype");
  hipStream_t s = as_stream(stream);
  const int L = fw * fw;
  const RegionWs lay = region_ws(M, fw, C);
  float* v = ws + lay.v;
  float* psum = ws + lay.psum;
  float* partial = ws + lay.partial;
  int* tickets = reinterpret_cast<int*>(ws + lay.tickets);
  const int ncslab = cdiv(C / vec, RP_CL);
  dim3 rgrid(lay.n_psum, M), pgrid(ncslab, lay.nslab);
#define RW(TT, MT, RAWV)                                                                                                       \
  hipLaunchKernelGGL((region_resample_kernel<TT, MT, RAWV>), rgrid, dim3(RP_POS), 0, s, (const MT*)masks, v, psum, tickets,    \
                     lay.ncslab_max, mh, mw, fw, rscale_h, rscale_w, ys, xs, rh, rw)
#define RP(TT, MMV)                                                                                                    \
  hipLaunchKernelGGL((region_pool_kernel<TT, MMV>), pgrid, dim3(256), 0, s, (const TT*)feat, v, psum, lay.n_psum, partial, \
                     tickets, (TT*)out, M, L, C)
  if (dtype == SRGPT_BF16) {
    if (raw) RW(bf16_t, unsigned char, true);
    else if (mask_dtype == SRGPT_BF16) RW(bf16_t, bf16_t, false);
    else RW(bf16_t, float, false);
    int ch = region_mfma_chunks((size_t)L);
#ifdef SRGPT_TUNING_KNOBS  // A/B (scripts/experiments/ab_region_pool.sh): 0 = the VALU kernel on bf16 features, 1 / 2 / 4 = that many chunks per wave
    const int mode = SRGPT_KNOB("SRGPT_REGION_MFMA", 3);
    if ((mode == 1 || mode == 2 || mode == 4) && cdiv(L, 128 * mode) <= lay.nslab_cap) ch = mode;  // if the partials fit
    if (mode == 0) {
      if (M <= 8) RP(bf16_t, 8); else RP(bf16_t, 16);
    } else
#endif
    {
      const dim3 mgrid(ncslab, cdiv(L, 4 * ch * 32));
#define RPM(CHV)                                                                                                              \
  hipLaunchKernelGGL((region_pool_mfma_kernel<CHV>), mgrid, dim3(256), 0, s, (const bf16_t*)feat, v, psum, lay.n_psum, partial, \
                     tickets, (bf16_t*)out, M, L, C)
      if (ch == 4) RPM(4); else if (ch == 2) RPM(2); else RPM(1);
#undef RPM
    }
  } else {
    if (raw) RW(float, unsigned char, true);
    else if (mask_dtype == SRGPT_BF16) RW(float, bf16_t, false);
    else RW(float, float, false);
    if (M <= 8) RP(float, 8); else RP(float, 16);
  }
#undef RW
#undef RP
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_region_pool(const void* feat, const void* masks, void* out, float* ws, int M, int mh, int mw,
                                 int fw, int C, float rscale_h, float rscale_w, int mask_dtype, int dtype,
                                 srgpt_stream_t stream) {
  return region_pool_impl(feat, masks, out, ws, M, mh, mw, fw, C, rscale_h, rscale_w, mask_dtype, dtype, nullptr, nullptr, 0, 0,
                          stream);
}

extern "C" int srgpt_region_pool_u8(const void* feat, const void* masks_u8, const int* ys, const int* xs, void* out, float* ws,
                                    int M, int rh, int rw, int mh, int mw, int fw, int C, float rscale_h, float rscale_w,
                                    int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(ys && xs && rh > 0 && rw > 0, SRGPT_ERR_ARG, "srgpt_region_pool_u8: null index tables / bad raw size");
  return region_pool_impl(feat, masks_u8, out, ws, M, mh, mw, fw, C, rscale_h, rscale_w, SRGPT_F32, dtype, ys, xs, rh, rw, stream);
}

#define RDISPATCH(dtype, ...)                      \
  if ((dtype) == SRGPT_BF16) {                     \
    using T = bf16_t;                              \
    __VA_ARGS__;                                   \
  } else if ((dtype) == SRGPT_F32) {               \
    using T = float;                               \
    __VA_ARGS__;                                   \
  } else {                                         \
    srgpt_set_error("bad dtype %d", (int)(dtype)); \
    return SRGPT_ERR_ARG;                          \
  }

extern "C" int srgpt_avgpool(const void* x, void* y, int n_img, int in_w, int out_w, int C, int dtype,
                             srgpt_stream_t stream) {
  SRGPT_CHECK(x && y && n_img > 0 && in_w > 0 && out_w > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_avgpool: bad args");
  SRGPT_CHECK(C % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_avgpool: C not a 16-byte multiple");
  RDISPATCH(dtype, hipLaunchKernelGGL(avgpool_kernel<T>, dim3(out_w * out_w, n_img), dim3(256), 0, as_stream(stream),
                                      (const T*)x, (T*)y, in_w, out_w, C));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_s2d(const void* x, void* y, int n_img, int g, int C, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(x && y && n_img > 0 && g > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_s2d: bad args");
  SRGPT_CHECK(C % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_s2d: C not a 16-byte multiple");
  const int hb = (g + 1) / 2;
  RDISPATCH(dtype, hipLaunchKernelGGL(s2d_kernel<T>, dim3(hb * hb, n_img), dim3(256), 0, as_stream(stream), (const T*)x,
                                      (T*)y, g, C));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_im2col(const void* images, void* out, int n_img, int S, int patch, int kp, int dtype,
                            srgpt_stream_t stream) {
  SRGPT_CHECK(images && out && n_img > 0 && S > 0 && patch > 0, SRGPT_ERR_ARG, "srgpt_im2col: bad args");
  SRGPT_CHECK(kp >= 3 * patch * patch, SRGPT_ERR_ARG, "srgpt_im2col: kp < 3*patch*patch");
  const int gw = S / patch;
  RDISPATCH(dtype, hipLaunchKernelGGL(im2col_kernel<T>, dim3(gw * gw, n_img), dim3(256), 0, as_stream(stream),
                                      (const T*)images, (T*)out, S, patch, kp));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_vit_assemble_cls(const void* patches, const void* cls_emb, const void* pos_emb, void* x, int n_img,
                                      int gg, int C, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(patches && cls_emb && pos_emb && x && n_img > 0 && gg > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_vit_assemble_cls: bad args");
  SRGPT_CHECK(C % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_vit_assemble_cls: C not a 16-byte multiple");
  RDISPATCH(dtype, hipLaunchKernelGGL(vit_assemble_cls_kernel<T>, dim3(gg + 1, n_img), dim3(128), 0, as_stream(stream),
                                      (const T*)patches, (const T*)cls_emb, (const T*)pos_emb, (T*)x, gg, C));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
