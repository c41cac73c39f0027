// fp8 x fp8 GEMM on the fp8 matrix pipe (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales) -- the opt-in W8A8 prefill of
// BASELINE configs[4] ("fp8 weights on CDNA4 fp8 MFMA"); the default fp8 mode stays W8A16 (gemm256.hip<W8>: exact weights x bf16
// activations).  C[M,N] = epilogue((A8[M,K] @ W8[N,K]^T) * ascale[m] * wscale[n]): both operands OCP e4m3fn bytes, K-contiguous,
// one power-of-two fp32 scale per ROW of each operand (weights: per output feature, quantised at load; activations: per token,
// srgpt_quant_rows_e4m3 below).  Products of two e4m3 values and their fp32 sums are exact in the MFMA's accumulator up to fp32
// rounding of the running sum, the scales are powers of two: the result equals the fp32 GEMM of the dequantised operands to fp32
// rounding -- what the activation quantisation costs against W8A16 is stated (and tested) at the model level, DESIGN.md section 4.
//
// The kernel is gemm256.hip's schedule with a K tile of 128 BYTES per row instead of 64 bf16 (the same 128-byte LDS rows, the same
// LDS-DMA instructions, barriers, counted waits and phase stagger -- see the header there; the 16-byte-slot swizzle differs): per
// K tile the matrix pipe does the same number of cycles (16 x v_mfma 16x16x128 at 32 cycles per phase and wave) on twice the K.
//   * waves 2 (M) x 4 (N); a wave owns 128 x 64 of C = 8 x 4 tiles of 16 x 16 (128 accumulator VGPRs)
//   * operand fragment of one MFMA: lane (r = lane & 15, g = lane >> 4) holds k = 32 g .. 32 g + 31 of row r = 16-byte slots 2g and
//     2g + 1 of the row: two ds_read_b128 at slot ^ f(row), f chosen for the instruction's 16-lane service groups (see below)
//   * phase P of a K tile = rows 64 P .. 64 P + 63 of the wave's A panel (4 fragments) x the 4 W fragments of the tile
// Requirements (checked by the launcher): K % 128 == 0, K >= 256, 16-byte aligned rows.
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

int srgpt_splitk_reduce_bf16(const Epilogue& e, hipStream_t s);  // gemm.hip

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int F_BM = 256, F_BN = 256, F_BK = 128;  // F_BK in fp8 elements = bytes
constexpr int F_OPER = F_BM * F_BK;                // bytes of one operand tile (32 KiB)
constexpr int F_BUF = 2 * F_OPER;                  // one K-tile buffer: A then W (64 KiB)
constexpr int F_LDS = 2 * F_BUF;                   // 128 KiB
constexpr int F_UNIT = 0x7f7f7f7f;                 // E8M0 block scale 2^0 in every byte

#define F_BARRIER()                      \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define F_VMCNT(N)                                            \
  do {                                                        \
    __builtin_amdgcn_sched_barrier(0);                        \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

__global__ __launch_bounds__(512, 2) void gemm_f8_256_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W,
                                                             const float* __restrict__ ascale, int K, int lda, Epilogue e, int gx,
                                                             int gy) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- block -> tile: XCD-contiguous runs (bijective for any grid size), 8-row bands inside a run ----
  int by, bx;
  {
    const int nwg = gx * gy, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int band = wg / (8 * gx), idx = wg - band * 8 * gx;
    const int hb = min(8, gy - band * 8);
    by = band * 8 + idx % hb;
    bx = idx / hb;
  }
  const int m0 = by * F_BM, n0 = bx * F_BN;

  const int nk_all = K / F_BK;
  const int kt0 = e.splits > 1 ? (int)blockIdx.y * e.tiles_per_split : 0;
  const int nk = e.splits > 1 ? min(nk_all, kt0 + e.tiles_per_split) : nk_all;

  // ---- DMA sources: one wave-instruction stages 8 rows x 128 B; lane -> row lane >> 3 of the group, physical slot lane & 7,
  //      which holds logical chunk slot ^ f(row), f(row) = (((row >> 1) & 7) + 6) & 7; every group of this wave has the parity of
  //      the wave, so (row >> 1) & 7 = ((wave & 1) << 2) | (row-in-group >> 1) ----
  const int lr = lane >> 3, lc = (lane & 7) ^ ((((((wave & 1) << 2) | (lr >> 1))) + 6) & 7);  // slot ^ f(row), f below
  // slot 0: W groups w, w+8 | slot 1: W groups 16+w, 24+w | slot 2: A groups w, 16+w | slot 3: A groups 8+w, 24+w
  const unsigned char* pw[4];
  const unsigned char* pa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gw_ = wave + 8 * i;
    const int ga_ = (i & 1) * 16 + (i >> 1) * 8 + wave;
    pw[i] = W + (size_t)min(n0 + gw_ * 8 + lr, e.N - 1) * K + lc * 16;
    pa[i] = A + (size_t)min(m0 + ga_ * 8 + lr, e.M - 1) * lda + lc * 16;
  }
  auto dma = [&](const void* src, int lds_off) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + lds_off), 16, 0, 0);
  };
  // stage slot `sl` (compile-time) of K tile kt into buffer kt & 1; which = 0 / 1: first / second instruction of the slot, 2: both
  auto stage = [&](auto sl_c, int kt, int which = 2) {
    constexpr int sl = decltype(sl_c)::value;
    const int k0 = kt * F_BK;
    const int boff = (kt & 1) * F_BUF;
    if constexpr (sl == 0) {
      if (which != 1) dma(pw[0] + k0, boff + F_OPER + (wave + 0) * 1024);
      if (which != 0) dma(pw[1] + k0, boff + F_OPER + (wave + 8) * 1024);
    } else if constexpr (sl == 1) {
      if (which != 1) dma(pw[2] + k0, boff + F_OPER + (wave + 16) * 1024);
      if (which != 0) dma(pw[3] + k0, boff + F_OPER + (wave + 24) * 1024);
    } else if constexpr (sl == 2) {
      if (which != 1) dma(pa[0] + k0, boff + (wave + 0) * 1024);
      if (which != 0) dma(pa[1] + k0, boff + (wave + 16) * 1024);
    } else {
      if (which != 1) dma(pa[2] + k0, boff + (wave + 8) * 1024);
      if (which != 0) dma(pa[3] + k0, boff + (wave + 24) * 1024);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;

  // ---- fragment addresses inside a 16-row tile (tile rows are multiples of 16: the swizzle depends on the lane only) ----
  // f(row) = (((row >> 1) & 7) + 6) & 7 -- not gemm256's (row >> 1) & 7: ds_read_b128 is serviced in four groups of 16 lanes,
  // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32, and here a group mixes rows 0-3 / 12-15 of one k quarter with
  // rows 4-11 of the next: with the plain swizzle both land on the same slots (PMC: SQ_LDS_BANK_CONFLICT = 50 % of the LDS
  // cycles); f sends row pairs 2-5 to slots 0-3 and pairs 0, 1, 6, 7 to slots 4-7, each set closed under ^ 2 (and ^ 4, ^ 6)
  const int r16 = lane & 15, g = lane >> 4, sw = ((r16 >> 1) + 6) & 7;
  const int fo_lo = r16 * 128 + (((2 * g) ^ sw) << 4);      // k = 32 g .. 32 g + 15
  const int fo_hi = r16 * 128 + (((2 * g + 1) ^ sw) << 4);  // k = 32 g + 16 .. 32 g + 31
  const int a_base = wr * 128 * 128;                        // + (4 P + mi) * 16 rows
  const int w_base = F_OPER + wc * 64 * 128;                // + jn * 16 rows
  auto frag = [&](const char* p) -> i32x8 {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(p + fo_lo);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(p + fo_hi);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 fw[4], fa[4];

  // ---- prologue: tile kt0 complete, W + A rows 0..63 / 128..191 of tile kt0+1 (the rest is staged in phase 0 of tile kt0) ----
  stage(S0{}, kt0);
  stage(S1{}, kt0);
  stage(S2{}, kt0);
  stage(S3{}, kt0);
  if (kt0 + 1 < nk) {
    stage(S0{}, kt0 + 1);
    stage(S1{}, kt0 + 1);
    stage(S2{}, kt0 + 1);
    F_VMCNT(6);
  } else {
    F_VMCNT(0);
  }
  F_BARRIER();
  // the late half runs one phase behind the early half; the two waves of a SIMD are in different halves
  const bool late = wr == 1;
  if (late) F_BARRIER();

  auto phase = [&](auto p_c, int kt, bool more1, bool more2) {
    constexpr int P = decltype(p_c)::value;
    const char* buf = lds + (kt & 1) * F_BUF;
    if constexpr (P == 0) {
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) fw[jn] = frag(buf + w_base + jn * 16 * 128);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) fa[mi] = frag(buf + a_base + (4 * P + mi) * 16 * 128);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P == 0) {
      if (more1) F_VMCNT(6); else F_VMCNT(0);
    } else {
      if (more2) F_VMCNT(2); else F_VMCNT(0);
    }
    // fragments land before the barrier: whoever passes it may overwrite what this phase read
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    F_BARRIER();
    // -------- multiply part: 16 MFMAs on 16 accumulators, this phase's DMA instructions in their shadow --------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int jn = 0; jn < 4; ++jn)
        acc[4 * P + mi][jn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[mi], fw[jn], acc[4 * P + mi][jn], 0, 0, 0, F_UNIT,
                                                                               0, F_UNIT);
      // anchor: the MFMA intrinsic has no side effects, and without a use here the compiler sinks all 32 of a K tile below the
      // phase's closing barrier (seen in the ISA) -- the schedule above exists to keep them between the barriers
      asm volatile("" : "+v"(acc[4 * P + mi][0]), "+v"(acc[4 * P + mi][1]), "+v"(acc[4 * P + mi][2]), "+v"(acc[4 * P + mi][3]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (P == 0) {
        if (more1 && (mi == 0 || mi == 2)) stage(S3{}, kt + 1, mi >> 1);
      } else {
        if (more2) {
          if (mi == 0) stage(S0{}, kt + 2, 2);
          if (mi == 1) stage(S1{}, kt + 2, 2);
          if (mi == 2) stage(S2{}, kt + 2, 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    F_BARRIER();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  for (int kt = kt0; kt < nk; ++kt) {
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    phase(P0{}, kt, more1, more2);
    phase(P1{}, kt, more1, more2);
  }
  if (!late) F_BARRIER();  // pairs with the last barrier of the late half

  // ---- epilogue.  D layout of the 16 x 16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg.  The arithmetic and rounding
  //      points of epilogue_store() (gemm_epilogue.h), with the loads batched: the column scale and bias once per column (4 per
  //      lane), the row scales and the 16 residual values of a 16-row strip issued together before anything consumes them (one
  //      memory latency per strip; a load + wait per ELEMENT cost 90 us per tile).  The stores stay one 2-byte element each:
  //      a quad transpose to 8-byte stores measured no gain -- what is left of the epilogue is the write of the tile itself
  //      (profiles/r02_w8a8_gemm.txt) ----
  if (e.splits > 1) {
    float* slab = e.partial + (size_t)blockIdx.y * e.M * e.N;
#pragma clang loop unroll(full)
    for (int i = 0; i < 8; ++i) {
      const int mb = m0 + wr * 128 + i * 16 + 4 * g;
      float as[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) as[q] = ascale[min(mb + q, e.M - 1)];
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int n = n0 + wc * 64 + jn * 16 + r16;
#pragma unroll
        for (int q = 0; q < 4; ++q)  // power-of-two row scale: exact; the column scale follows in the slab reduction
          if (mb + q < e.M && n < e.N) slab[(size_t)(mb + q) * e.N + n] = acc[i][jn][q] * as[q];
      }
    }
    return;
  }
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(e.bias);
  const bool has_res = e.residual != nullptr;
  // no residual: the loads below still run, from one valid address, so that nothing loaded is consumed behind a branch -- hipcc
  // answers a load consumed under a condition with s_waitcnt vmcnt(0) in front of every use, and stores count in vmcnt: the
  // first version of this epilogue waited for each of its 128 stores to be acknowledged before issuing the next
  const bf16_t* rp = has_res ? reinterpret_cast<const bf16_t*>(e.residual) : reinterpret_cast<const bf16_t*>(ascale);
  float sc[4], bs[4];
  int ncl[4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) {
    ncl[jn] = min(n0 + wc * 64 + jn * 16 + r16, e.N - 1);
    sc[jn] = e.wscale[ncl[jn]];
  }
  {
    const bf16_t* bp = bias ? bias : reinterpret_cast<const bf16_t*>(ascale);
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) bs[jn] = to_f(bp[bias ? ncl[jn] : 0]);
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) bs[jn] = bias ? bs[jn] : 0.f;
  }
  float as[2][4], res[2][4][4];
  auto load_strip = [&](int i, float (&as_)[4], float (&res_)[4][4]) {
    const int mb = m0 + wr * 128 + i * 16 + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mc = min(mb + q, e.M - 1);
      as_[q] = ascale[mc];
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) res_[jn][q] = to_f(rp[has_res ? (size_t)mc * e.N + ncl[jn] : (size_t)0]);
    }
  };
  load_strip(0, as[0], res[0]);
#pragma clang loop unroll(full)
  for (int i = 0; i < 8; ++i) {
    const int mb = m0 + wr * 128 + i * 16 + 4 * g;
    if (i + 1 < 8) load_strip(i + 1, as[(i + 1) & 1], res[(i + 1) & 1]);  // in flight across this strip's stores
    float v[4][4];
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float y = rnd<bf16_t>(acc[i][jn][q] * as[i & 1][q] * sc[jn] + bs[jn]);  // the Linear's output is materialised in bf16
        v[jn][q] = has_res ? rnd<bf16_t>(y + res[i & 1][jn][q]) : y;
      }
    if (e.out_f32) {
      float* c = reinterpret_cast<float*>(e.C);
#pragma unroll
      for (int jn = 0; jn < 4; ++jn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mb + q, n = n0 + wc * 64 + jn * 16 + r16;
          if (m < e.M && n < e.N) c[(size_t)m * e.ldc + n] = v[jn][q];
        }
    } else {
      bf16_t* c = reinterpret_cast<bf16_t*>(e.C);
#pragma unroll
      for (int jn = 0; jn < 4; ++jn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mb + q, n = n0 + wc * 64 + jn * 16 + r16;
          if (m < e.M && n < e.N) c[(size_t)m * e.ldc + n] = from_f<bf16_t>(v[jn][q]);
        }
    }
  }
}

// Per-row (per-token) fp8 quantisation of the activations: scale[m] = the smallest power of two with max|x[m,:]| / scale <= 448
// (448 = 0.875 * 2^9; frexp / ldexp, no division -- the same rule as the weights, spatialrgpt_amd/ops.py quantize_fp8_rows),
// q[m,k] = e4m3fn(x[m,k] / scale) round-to-nearest-even (v_cvt_pk_fp8_f32; the quotient is exact and <= 448, so the conversion
// never saturates).  One block per row; the row is read twice (the second pass hits L2).
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* __restrict__ x, int ldx, unsigned char* __restrict__ q,
                                                         float* __restrict__ scale, int K) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + (size_t)row * ldx);
  const int nc = K >> 3;
  float amax = 0.f;
  for (int c = tid; c < nc; c += 256) {
    const bf16x8 v = xr[c];
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf((float)v[i]));
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int ex;
  const float mant = frexpf(fmaxf(amax, 0x1p-100f), &ex);  // amax = mant * 2^ex, mant in [0.5, 1)
  const int k = ex - 9 + (mant > 0.875f ? 1 : 0);
  const float inv = ldexpf(1.f, -k);
  if (tid == 0) scale[row] = ldexpf(1.f, k);
  u32x2* qr = reinterpret_cast<u32x2*>(q + (size_t)row * K);
  for (int c = tid; c < nc; c += 256) {
    const bf16x8 v = xr[c];
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[0] * inv, (float)v[1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[2] * inv, (float)v[3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[4] * inv, (float)v[5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[6] * inv, (float)v[7] * inv, hi, true);
    qr[c] = u32x2{(unsigned int)lo, (unsigned int)hi};
  }
}

// The same quantisation with the PRODUCER of the rows fused in (W8A8 prefill: the RMSNorm in front of q/k/v and gate/up, the
// SwiGLU in front of down_proj), so the bf16 intermediate is neither written nor read: MODE 1 = LlamaRMSNorm (rmsnorm_kernel's
// arithmetic: the same per-thread chunk order and block reduction for the mean of squares, weight * (x * r).to(bf16)),
// MODE 2 = silu_mul_kernel's bf16(bf16(silu(gate)) * up) on a [gate | up] row.  The produced row lives in registers (a thread
// owns chunks tid, tid + 256, ... : NV of them), its maximum and its codes come from there -- bit-identical to the two launches.
template <int MODE, int NV>
__global__ __launch_bounds__(256) void quant_rows_fused_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ nw,
                                                               float eps, unsigned char* __restrict__ q, float* __restrict__ scale,
                                                               int K) {
  __shared__ float red[16];
  __shared__ float redm[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + (size_t)row * ldx);
  const int nc = K >> 3;
  bf16x8 val[NV];
  if (MODE == 1) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = tid + 256 * v;
      if (c < nc) {
        val[v] = xr[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (float)val[v][i] * (float)val[v][i];
      }
    }
    const float r = rsqrtf(block_sum(s, red) / (float)K + eps);
    const bf16x8* gr = reinterpret_cast<const bf16x8*>(nw);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = tid + 256 * v;
      if (c < nc) {
        const bf16x8 g = gr[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) val[v][i] = (bf16_t)((float)g[i] * rnd<bf16_t>((float)val[v][i] * r));
      }
    }
  } else {
    const bf16x8* ur = reinterpret_cast<const bf16x8*>(x + (size_t)row * ldx + K);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = tid + 256 * v;
      if (c < nc) {
        const bf16x8 g = xr[c], u = ur[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) val[v][i] = (bf16_t)(rnd<bf16_t>(silu((float)g[i])) * (float)u[i]);
      }
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v)
    if (tid + 256 * v < nc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf((float)val[v][i]));
    }
  amax = wave_max(amax);
  if ((tid & 63) == 0) redm[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  int ex;
  const float mant = frexpf(fmaxf(amax, 0x1p-100f), &ex);
  const int k = ex - 9 + (mant > 0.875f ? 1 : 0);
  const float inv = ldexpf(1.f, -k);
  if (tid == 0) scale[row] = ldexpf(1.f, k);
  u32x2* qr = reinterpret_cast<u32x2*>(q + (size_t)row * K);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = tid + 256 * v;
    if (c < nc) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)val[v][0] * inv, (float)val[v][1] * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)val[v][2] * inv, (float)val[v][3] * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)val[v][4] * inv, (float)val[v][5] * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)val[v][6] * inv, (float)val[v][7] * inv, hi, true);
      qr[c] = u32x2{(unsigned int)lo, (unsigned int)hi};
    }
  }
}

template <int MODE>
int launch_quant_fused(const void* x, int ldx, const void* nw, float eps, void* q, float* scale, int M, int K, hipStream_t s) {
  const int nv = (K / 8 + 255) / 256;  // chunks per thread
#define SRGPT_QF(NV) hipLaunchKernelGGL((quant_rows_fused_kernel<MODE, NV>), dim3(M), dim3(256), 0, s, (const bf16_t*)x, ldx, \
                                        (const bf16_t*)nw, eps, (unsigned char*)q, scale, K)
  if (nv <= 2) SRGPT_QF(2);
  else if (nv <= 4) SRGPT_QF(4);
  else if (nv <= 8) SRGPT_QF(8);
  else {
    srgpt_set_error("fused row quantisation: K=%d exceeds 16384 columns", K);
    return SRGPT_ERR_UNSUPPORTED;
  }
#undef SRGPT_QF
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

}  // namespace

extern "C" {

int srgpt_quant_rows_e4m3(const void* x, void* q, float* scale, int M, int K, int ldx, srgpt_stream_t stream) {
  SRGPT_CHECK(x && q && scale, SRGPT_ERR_ARG, "srgpt_quant_rows_e4m3: null pointer");
  SRGPT_CHECK(M > 0 && K > 0 && ldx >= K, SRGPT_ERR_ARG, "srgpt_quant_rows_e4m3: bad shape M=%d K=%d ldx=%d", M, K, ldx);
  SRGPT_CHECK(K % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 8) == 0, SRGPT_ERR_UNSUPPORTED,
              "srgpt_quant_rows_e4m3: K and ldx must be multiples of 8 and the rows 16-byte aligned");
  hipLaunchKernelGGL(quant_rows_kernel, dim3(M), dim3(256), 0, as_stream(stream), (const bf16_t*)x, ldx, (unsigned char*)q, scale,
                     K);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

int srgpt_quant_rows_e4m3_rmsnorm(const void* x, const void* norm_w, float eps, void* q, float* scale, int M, int K, int ldx,
                                  srgpt_stream_t stream) {
  SRGPT_CHECK(x && norm_w && q && scale, SRGPT_ERR_ARG, "srgpt_quant_rows_e4m3_rmsnorm: null pointer");
  SRGPT_CHECK(M > 0 && K > 0 && ldx >= K, SRGPT_ERR_ARG, "srgpt_quant_rows_e4m3_rmsnorm: bad shape M=%d K=%d ldx=%d", M, K, ldx);
  SRGPT_CHECK(K % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)norm_w % 16) == 0 && ((uintptr_t)q % 8) == 0,
              SRGPT_ERR_UNSUPPORTED, "srgpt_quant_rows_e4m3_rmsnorm: K and ldx must be multiples of 8 and the rows 16-byte aligned");
  return launch_quant_fused<1>(x, ldx, norm_w, eps, q, scale, M, K, as_stream(stream));
}

int srgpt_quant_rows_e4m3_swiglu(const void* gate_up, void* q, float* scale, int M, int inter, srgpt_stream_t stream) {
  SRGPT_CHECK(gate_up && q && scale, SRGPT_ERR_ARG, "srgpt_quant_rows_e4m3_swiglu: null pointer");
  SRGPT_CHECK(M > 0 && inter > 0, SRGPT_ERR_ARG, "srgpt_quant_rows_e4m3_swiglu: bad shape M=%d inter=%d", M, inter);
  SRGPT_CHECK(inter % 8 == 0 && ((uintptr_t)gate_up % 16) == 0 && ((uintptr_t)q % 8) == 0, SRGPT_ERR_UNSUPPORTED,
              "srgpt_quant_rows_e4m3_swiglu: inter must be a multiple of 8 and the rows 16-byte aligned");
  return launch_quant_fused<2>(gate_up, 2 * inter, nullptr, 0.f, q, scale, M, inter, as_stream(stream));
}

// C = ((A8 @ W8^T) * ascale[m] * wscale[n] + bias[n]) + residual in bf16 (or fp32 if out_f32), no activation (the LLM's Linears
// have none); K splits (deterministic slabs) when the 256 x 256
// tiles do not fill the chip -- the rule of srgpt_gemm_w8.
int srgpt_gemm_w8a8(const void* A8, const float* ascale, const void* W8, const float* wscale, const void* bias,
                    const void* residual, void* C, int M, int N, int K, int lda, int ldc, int out_f32, void* ws,
                    int64_t ws_bytes, srgpt_stream_t stream) {
  SRGPT_CHECK(A8 && ascale && W8 && wscale && C, SRGPT_ERR_ARG, "srgpt_gemm_w8a8: null pointer");
  SRGPT_CHECK(M > 0 && N > 0 && K > 0, SRGPT_ERR_ARG, "srgpt_gemm_w8a8: bad shape M=%d N=%d K=%d", M, N, K);
  SRGPT_CHECK(lda >= K, SRGPT_ERR_ARG, "srgpt_gemm_w8a8: lda < K");
  SRGPT_CHECK(ldc >= N, SRGPT_ERR_ARG, "srgpt_gemm_w8a8: ldc < N");
  SRGPT_CHECK(K % F_BK == 0 && K >= 2 * F_BK && lda % 16 == 0 && ((uintptr_t)A8 % 16) == 0 && ((uintptr_t)W8 % 16) == 0,
              SRGPT_ERR_UNSUPPORTED, "srgpt_gemm_w8a8: K must be a multiple of 128 (>= 256) and the rows 16-byte aligned (K=%d lda=%d)",
              K, lda);
  Epilogue e{bias, residual, C, M, N, ldc, SRGPT_ACT_NONE, 0, 0, out_f32, SRGPT_OUT_PLAIN, 0, nullptr, 1, 0, wscale};
  hipStream_t s = as_stream(stream);
  const int cus = srgpt_device_cus();
  const int nk = K / F_BK;
  const int gx = cdiv(N, F_BN), gy = cdiv(M, F_BM);
  const long tiles = (long)gx * gy;
  int sp = 1;
  if (tiles < cus && ws) {
    sp = (int)(cus / tiles);
    if (sp > 4) sp = 4;
    while (sp > 1 && (nk / sp < 4 || (int64_t)sp * M * N * 4 > ws_bytes)) --sp;
  }
  if (sp > 1) {
    e.partial = reinterpret_cast<float*>(ws);
    e.tiles_per_split = cdiv(nk, sp);
    e.splits = cdiv(nk, e.tiles_per_split);
  }
  static std::atomic<uint64_t> attr_done{0};
  SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)gemm_f8_256_kernel, F_LDS));
  hipLaunchKernelGGL(gemm_f8_256_kernel, dim3(gx * gy, e.splits > 1 ? e.splits : 1), dim3(512), F_LDS, s,
                     (const unsigned char*)A8, (const unsigned char*)W8, ascale, K, lda, e, gx, gy);
  SRGPT_LAUNCH_CHECK();
  if (e.splits > 1) SRGPT_TRY(srgpt_splitk_reduce_bf16(e, s));
  return SRGPT_OK;
}
}
