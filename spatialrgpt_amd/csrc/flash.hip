// MFMA flash attention (bf16) -- prefill / ViT.  v_mfma_f32_16x16x32_bf16, online fp32 softmax.
//
// Both products are issued "swapped" (S^T = K Q^T, O^T = V^T P^T) so the softmax row of a query lives in ONE lane column
// (col = lane & 15): row max / sum need two cross-lane steps, the rescale factor is lane-local, and P never leaves registers --
// the PV contraction simply runs over the keys in the order the QK^T accumulator already holds them.
//
// Round 4 (profiles/r04_flash_attention.txt; same products in the same order as the round-3 kernel, results differ only by the
// softmax exponent being one fma + exp2 instead of mul / sub / mul / exp2):
//   * V is parked ROW-major like K (16-byte writes; the round-3 form wrote the transposed tile two bytes at a time, 24 ds_write_b16
//     per thread and tile) and the A operand of O^T += V^T P^T -- 8 keys of one d per lane -- is read back with the transposing
//     LDS read (ds_read_b64_tr_b16: a 16-lane group addresses a [4 keys][16 d] block, every lane receives one d's 4 keys); row
//     stride HDP + 16 elements = 32 B x odd, so the 8 rows a 32-lane half touches cover all 64 banks;
//   * two LDS buffers and two register sets: tile t + 1 is parked and tile t + 3 requested at the top of step t, ahead of the
//     products of tile t -- a request is two whole steps old when its registers are needed, ONE barrier per tile instead of two;
//   * the kernel was VALU-bound, not MFMA- or LDS-bound (PMC: 3 waves per SIMD each 35 % VALU-active, 360 VALU instructions per
//     key tile against 24 MFMAs): K / V rows through buffer descriptors (no 64-bit address arithmetic, no zeroing selects), the
//     softmax on the raw products, masks only on tiles that contain a masked position, and -- the largest single item -- this file
//     is compiled with MFMA accumulators in arch VGPRs (Makefile): in the default AGPR form the per-tile rescale of the output
//     accumulators cost 24 v_accvgpr_read + 24 v_accvgpr_write per wave and tile.
// This file holds the kernel and its launcher only; the C entry point (srgpt_attention) is in attn.hip.
#include <type_traits>

#include "common.h"
#include "flash.h"

namespace {

constexpr int QBLK = 64, KVBLK = 64;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifdef SRGPT_TUNING_KNOBS
// phase stamps of block (0, 0, 0), wave 0, key tile 5 (scripts/experiments/ubench_flash_stamps.py)
__device__ unsigned long long srgpt_flash_stamps[16];
#define FL_STAMP(i) do { if (stamp_on && t == 5 && threadIdx.x == 0) srgpt_flash_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FL_STAMP(i) do { } while (0)
#endif

template <int HDP, bool CAUSAL>
__global__ __launch_bounds__(256) void flash_bf16_kernel(AttnArgs a) {
  constexpr int NS = HDP / 8;                 // 16-byte slots per K / V row
  constexpr int SWM = (NS % 8 == 0) ? 7 : 3;  // swizzle mask of the K tile (keeps a slot inside its aligned group)
  constexpr int NKS = HDP / 32;               // k-steps of QK^T
  constexpr int ND = HDP / 16;                // 16-wide d sub-tiles of the output
  constexpr int VLD = HDP + 16;               // V row stride (elements)
  constexpr int NLD = NS / 4;                 // 16-byte loads per thread and operand for one tile
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][KVBLK * HDP];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][KVBLK * VLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QBLK;
  const int hk = h / (a.Hq / a.Hkv);
  const int klen = a.kv_len ? min(a.kv_len[b], a.Tk) : a.Tk;
  const int qrow = q0 + wave * 16 + lq;  // query row owned by this lane column
  const int coff = a.Tk - a.Tq;          // causal offset: key s visible iff s <= t + coff

  int kend = klen;
  if (CAUSAL) kend = min(kend, q0 + QBLK + coff);  // keys beyond the last query of the block are masked
  const int ntiles = (kend + KVBLK - 1) / KVBLK;
  const bf16_t* kb = a.k + b * a.k_bs + hk * a.k_hs;
  const bf16_t* vb = a.v + b * a.v_bs + hk * a.v_hs;

  // K / V rows come through buffer descriptors that end at the last valid row of this (batch, head) slice: rows past the
  // sequence read as zeros without a compare or a select, columns past D (the padded slots of a 72-wide head) get an offset
  // that is out of range by construction, and an address is ONE 32-bit add per load and tile (the flat-pointer form spent ~75
  // VALU instructions per tile on 64-bit row * stride arithmetic and 24 on zeroing selects).
  const int klc = max(klen, 1);
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(kb), 0, (int)(((int64_t)(klc - 1) * a.k_ts + a.D) * sizeof(bf16_t)), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(vb), 0, (int)(((int64_t)(klc - 1) * a.v_ts + a.D) * sizeof(bf16_t)), 0x00020000);
  int kvo[NLD], vvo[NLD];  // byte offsets of this thread's loads in tile 0
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = tid + 256 * i;
    const int key = idx / NS, c = idx - key * NS;
    const bool colok = c * 8 < a.D;
    kvo[i] = colok ? (int)((key * a.k_ts + c * 8) * sizeof(bf16_t)) : (int)0x80000000;
    vvo[i] = colok ? (int)((key * a.v_ts + c * 8) * sizeof(bf16_t)) : (int)0x80000000;
  }
  const int ktile = (int)(KVBLK * a.k_ts * sizeof(bf16_t)), vtile = (int)(KVBLK * a.v_ts * sizeof(bf16_t));
  u32x4 kreg[2][NLD], vreg[2][NLD];
  auto fetch = [&](auto SET, int t) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      kreg[S][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[i] + t * ktile, 0, 0);
      vreg[S][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvo[i] + t * vtile, 0, 0);
    }
  };
  auto park = [&](auto SET) {  // registers of set S -> LDS buffer S
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i;
      const int key = idx / NS, c = idx - key * NS;
      *reinterpret_cast<u32x4*>(Ks[S] + key * HDP + ((c ^ (key & SWM)) << 3)) = kreg[S][i];
      *reinterpret_cast<u32x4*>(Vs[S] + key * VLD + c * 8) = vreg[S][i];
    }
  };

  // Q fragments (B operand of S^T = K Q^T): lane (n = q, kgroup g) holds Q[q][ks*32 + 8g .. +8]
  bf16x8 qf[NKS];
  if (ntiles > 0) {
    fetch(std::integral_constant<int, 0>{}, 0);
    fetch(std::integral_constant<int, 1>{}, 1);
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int d = ks * 32 + g * 8;
    if (qrow < a.Tq && d < a.D)
      qf[ks] = *reinterpret_cast<const bf16x8*>(a.q + b * a.q_bs + (int64_t)qrow * a.q_ts + h * a.q_hs + d);
    else
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[ks][i] = (bf16_t)0.f;
  }
  if (ntiles > 0) {
    park(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 0>{}, 2);
  }

  f32x4 o[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;             // running maximum of the RAW products, running sum of exp(scale * (x - m))
  const float c2 = a.scale * 1.44269504088896340736f;

#ifdef SRGPT_TUNING_KNOBS
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
#endif
  auto step = [&](auto CUR, int t) {
    constexpr int S = decltype(CUR)::value;
    const int k0 = t * KVBLK;
    FL_STAMP(0);
    __syncthreads();  // tile t parked by every wave; every wave done with tile t - 1 (the other buffer)
    FL_STAMP(1);
    if (t + 1 < ntiles) park(std::integral_constant<int, S ^ 1>{});
    fetch(std::integral_constant<int, S ^ 1>{}, t + 3);
    FL_STAMP(2);
    const bf16_t* Kt = Ks[S];
    const bf16_t* Vt = Vs[S];

    // ---- S^T = K Q^T : p[s][r] = raw product of (key = k0 + 16 s + 4 g + r, query = qrow) ----
    // The softmax runs on the RAW products: with scale > 0 the row maximum commutes with the scaling, and
    // exp(scale * (x - max)) = exp2(fma(x, c, -max * c)), c = scale * log2(e): one fma + one v_exp per score instead of
    // mul / compare / select / max / sub / mul / exp.  Masks (keys past the row's length, the causal diagonal) are applied only
    // on the tiles that contain a masked position -- a wave-uniform branch; the interior tiles carry none of it.
    float p[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int key = 16 * s + lq;  // A-operand row of this lane
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kt + key * HDP + (((ks * 4 + g) ^ (key & SWM)) << 3));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) p[s][r] = acc[r];
    }
    FL_STAMP(3);
    bool edge = k0 + KVBLK > klen;
    if (CAUSAL) edge = edge || (k0 + KVBLK - 1 > q0 + wave * 16 + coff);  // the wave's FIRST query does not see the tile's last key
    if (edge) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kidx = k0 + 16 * s + 4 * g + r;
          bool ok = kidx < klen;
          if (CAUSAL) ok = ok && (kidx <= qrow + coff);
          if (!ok) p[s][r] = -INFINITY;
        }
    }
    float mx = fmaxf(fmaxf(p[0][0], p[0][1]), fmaxf(p[0][2], p[0][3]));
#pragma unroll
    for (int s = 1; s < 4; ++s) mx = fmaxf(fmaxf(mx, fmaxf(p[s][0], p[s][1])), fmaxf(p[s][2], p[s][3]));
    // over the four 16-lane rows (the 4 g groups of a query column): v_permlane16/32_swap, not ds_bpermute (common.h)
    mx = rows_pair(mx, [](float a, float b) { return fmaxf(a, b); });
    mx = halves_pair(mx, [](float a, float b) { return fmaxf(a, b); });
    const float m_new = fmaxf(m, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f((m - m_use) * c2);  // m = -inf -> 0
    const float mc = -m_use * c2;
    float rs = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[s][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[s][r], c2, mc));
        rs += p[s][r];
      }
    rs = rows_pair(rs, [](float a, float b) { return a + b; });
    rs = halves_pair(rs, [](float a, float b) { return a + b; });
    l = l * alpha + rs;
    m = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // the running maximum of some row of the wave moved
#pragma unroll
      for (int i = 0; i < ND; ++i) o[i] *= alpha;
    }

    // ---- O^T += V^T P^T ; contraction index 8g+i <-> key 32j + 4g + i (i<4), 32j + 16 + 4g + (i-4) ----
    FL_STAMP(4);
    const bf16_t* vp = Vt + (4 * g + (lq >> 2)) * VLD + 4 * (lq & 3);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (bf16_t)p[2 * j][r];
        pf[4 + r] = (bf16_t)p[2 * j + 1][r];
      }
#pragma unroll
      for (int ds = 0; ds < ND; ++ds) {
        const bf16_t* vq = vp + 32 * j * VLD + 16 * ds;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vq));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vq + 16 * VLD));
        const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[ds] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[ds], 0, 0, 0);
      }
    }
    FL_STAMP(5);
  };
  for (int t = 0; t < ntiles; t += 2) {
    step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) step(std::integral_constant<int, 1>{}, t + 1);
  }

  // ---- epilogue: O[b, q, h, d], d = 16 ds + 4 g + r ----
  if (qrow < a.Tq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* orow = a.o + (((int64_t)b * a.Tq + qrow) * a.Hq + h) * a.D;
#pragma unroll
    for (int ds = 0; ds < ND; ++ds) {
      const int d = ds * 16 + 4 * g;
      if (d < a.D) {  // D % 8 == 0 -> a group of 4 is entirely in or out
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[ds][r] * inv);
        *reinterpret_cast<bf16x4*>(orow + d) = w;
      }
    }
  }
}

}  // namespace

int64_t srgpt_flash_slice_span_limit() { return (int64_t)1 << 31; }

// launches the MFMA kernel; the caller has checked dtype, alignment, D <= 128, scale > 0 and the 32-bit span of a (batch, head) slice
void srgpt_flash_bf16_launch(const AttnArgs& a, int B, bool causal, hipStream_t s) {
  dim3 grid(cdiv(a.Tq, QBLK), a.Hq, B);
  const int hdp = (a.D + 31) / 32 * 32;
#define LF(H)                                                                  \
  if (causal)                                                                  \
    hipLaunchKernelGGL((flash_bf16_kernel<H, true>), grid, dim3(256), 0, s, a); \
  else                                                                         \
    hipLaunchKernelGGL((flash_bf16_kernel<H, false>), grid, dim3(256), 0, s, a)
  switch (hdp) {
    case 32: LF(32); break;
    case 64: LF(64); break;
    case 96: LF(96); break;
    default: LF(128); break;
  }
#undef LF
}

#ifdef SRGPT_TUNING_KNOBS
extern "C" int srgpt_flash_debug_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(srgpt_flash_stamps), sizeof(unsigned long long) * (n < 16 ? n : 16));
}
#endif
