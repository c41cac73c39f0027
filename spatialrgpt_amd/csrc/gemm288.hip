// bf16 MFMA GEMM for the bs = 1 prefill products: ALL rows of a 225 .. 272-row activation (T = 259 at BASELINE configs[1]) in one
// block.  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), both operands K-contiguous, 272 x 128 x 64 block tile, 8 waves.
//
// Why a kernel of its own: at M = 259 the product sits on the ridge (259 flop per weight byte against ~312 for the chip), so it
// has to stream W at HBM rate AND keep the matrix pipe fed.  The 96 x 128 tiles of gemm.hip read every W tile three times (once
// per row tile; L2 absorbs most of it) and move 17.8 B through the global -> LDS path per kFLOP; here W crosses that path once
// (11 B / kFLOP) and a block's requests are two K tiles ahead.  profiles/r04_gemm288.txt has the build-up with numbers (9 waves x
// 32 rows -> this form; lockstep -> staggered; the variants that lost) and the ablations that name the two ceilings: the L2 -> LDS
// path (717 MB per gate/up launch, 71 us alone) and the matrix pipe (49 us alone at the sustained clock).
//   * waves 4 (M) x 2 (N): a wave owns 64 x 64 of C = 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 (64 accumulator VGPRs, 8 A + 8 W
//     fragment reads per K tile); rows 256 .. 271 (3 of them exist at T = 259) are one extra 16 x 16 tile per wave
//     (v_mfma_f32_16x16x32_bf16 on W rows 16 w .. 16 w + 15) -- a ninth wave would put three waves on one SIMD
//   * LDS: THREE stages x (A 272 x 64 + W 128 x 64) bf16 = 150 KiB, rows of 128 B in 16-byte slots, slot ^ ((row >> 1) & 7)
//     (gemm256's conflict-free ds_read_b128 layout), filled by LDS-DMA (global_load_lds_dwordx4: one wave-instruction = 8 rows
//     x 128 B).  Per K tile a wave requests 4 A groups + 2 W groups; waves 0 and 1 also carry the two tail groups (a wave-uniform
//     branch picks their wait immediate: 7 instead of 6)
//   * the wave halves run half a K tile apart: a wave alternates a LOAD part (its 16 + 4 fragments into registers) and a MULTIPLY
//     part (16 + 2 MFMAs, the requests of a later tile between them), one workgroup barrier after each; waves 4 - 7 take one extra
//     barrier first, so one wave of every SIMD multiplies while the other reads.  The protocol (which part requests which tile,
//     where each share is waited for, why a stage is free when it is overwritten) is spelled out at the loop
//   * split-K over blockIdx.y for the narrow products (deterministic fp32 slabs, reduced by the split-K kernels of gemm.hip, which
//     also carry the norm / RoPE that follows); SwiGLU mode for the stacked gate / up weight (srgpt_gemm_swiglu): the block's 128 W
//     rows are 64 gate rows and the 64 matching up rows, the activation is the epilogue
// Requirements (checked by the callers in gemm.hip): K % 64 == 0, at least 4 K tiles, M <= 272 per row tile.
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int T_BM = 272, T_BN = 128, T_BK = 64, T_NS = 3;
constexpr int T_MAIN = 256;                  // rows of the eight 32-row wave tiles; rows 256 .. 271 are the 16-row tail
constexpr int T_A = T_BM * T_BK * 2;         // 34 KiB
constexpr int T_W = T_BN * T_BK * 2;         // 16 KiB
constexpr int T_STAGE = T_A + T_W;           // 50 KiB
constexpr int T_LDS = T_NS * T_STAGE;        // 150 KiB

#define T_BARRIER()                       \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)
#define T_VMCNT(N)                                            \
  do {                                                        \
    __builtin_amdgcn_sched_barrier(0);                        \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

__global__ __launch_bounds__(512, 1) void gemm_bf16_288_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, int K,
                                                               int lda, Epilogue e) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // SwiGLU mode: the block's 128 W rows are 64 gate rows and the 64 up rows of the same outputs (tile rows 64 .. 127)
  const int swi = e.swiglu_inter;
  const int n0 = blockIdx.x * (swi ? T_BN / 2 : T_BN), m0 = blockIdx.z * T_BM;

  const int nk_all = K / T_BK;
  const int kt0 = e.splits > 1 ? (int)blockIdx.y * e.tiles_per_split : 0;
  const int nk = e.splits > 1 ? min(nk_all, kt0 + e.tiles_per_split) : nk_all;

  // ---- requests: lane -> row lane >> 3 of an 8-row group, physical slot lane & 7 = logical chunk ^ ((row >> 1) & 7); every group
  //      of a wave has g & 1 == wave & 1, so the chunk a lane fetches is the same for all of them.  Per K tile a wave requests A
  //      groups wave + 8 i (i < 4) and W groups wave + 8 i (i < 2); the 16-row tail (rows 256 .. 271; 3 of them exist at T = 259)
  //      is two more A groups on waves 0 and 1: those two have 7 requests per tile in flight, the others 6 -- a wave-uniform
  //      branch picks the wait immediate ----
  const int lr = lane >> 3, lc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
  const bool has_tail = m0 + T_MAIN < e.M;  // uniform
  const bool tail_wave = has_tail && wave < 2;
  const bf16_t* pa[4];
  const bf16_t* pw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) pa[i] = A + (size_t)min(m0 + (wave + 8 * i) * 8 + lr, e.M - 1) * lda + lc * 8;  // A groups 0 .. 31
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // W groups 0 .. 15
    const int r = (wave + 8 * i) * 8 + lr;  // row of the W tile
    const int wrow = swi ? (r < 64 ? n0 + r : swi + n0 + r - 64) : min(n0 + r, e.N - 1);
    pw[i] = W + (size_t)wrow * K + lc * 8;
  }
  const bf16_t* ptl = A + (size_t)min(m0 + T_MAIN + wave * 8 + lr, e.M - 1) * lda + lc * 8;                    // A group 32 + wave
  auto dma = [&](const void* s_, int lds_off) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,
                                     (__attribute__((address_space(3))) void*)(lds + lds_off), 16, 0, 0);
  };
  auto request = [&](int kt) {  // the wave's groups of K tile kt into stage kt % 3
    const int k0 = kt * T_BK;
    const int soff = (kt % T_NS) * T_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(pw[i] + k0, soff + T_A + (wave + 8 * i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(pa[i] + k0, soff + (wave + 8 * i) * 1024);
    if (tail_wave) dma(ptl + k0, soff + (32 + wave) * 1024);
  };
  auto request_part = [&](int kt, int part) {  // the same requests in three portions (issued between the MFMAs of a multiply part)
    const int k0 = kt * T_BK;
    const int soff = (kt % T_NS) * T_STAGE;
    if (part == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) dma(pw[i] + k0, soff + T_A + (wave + 8 * i) * 1024);
    } else if (part == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) dma(pa[i] + k0, soff + (wave + 8 * i) * 1024);
    } else {
#pragma unroll
      for (int i = 2; i < 4; ++i) dma(pa[i] + k0, soff + (wave + 8 * i) * 1024);
      if (tail_wave) dma(ptl + k0, soff + (32 + wave) * 1024);
    }
  };
  // tail fragments for v_mfma_f32_16x16x32_bf16: lane (row lane & 15, k group lane >> 4), 16 bytes: chunk 4 s + (lane >> 4);
  // wave w multiplies the 16 tail rows with W rows 16 w .. 16 w + 15 of the block's column tile
  const int l15 = lane & 15, g4 = lane >> 4;
  const int trow = wave * 16 + l15;
  int toff[2], taoff[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    toff[s2] = T_A + trow * 128 + (((4 * s2 + g4) ^ ((trow >> 1) & 7)) << 4);
    taoff[s2] = (T_MAIN + l15) * 128 + (((4 * s2 + g4) ^ ((l15 >> 1) & 7)) << 4);
  }

  // ---- fragment addresses (gemm256): row = tile row + (lane & 31), slot = (2 ks + (lane >> 5)) ^ ((row >> 1) & 7) ----
  const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + ((hi ^ (sw & 1)) << 4) + ((ks ^ (sw >> 1)) << 5);
  const int wm = wave >> 1, wn = wave & 1;  // waves 4 (M) x 2 (N): 64 x 64 of C each = 2 x 2 MFMA tiles (8 A + 8 W fragment reads per K tile)
  const int a_base = wm * 64 * 128, w_base = T_A + wn * 64 * 128;

  f32x16 acc[4];
  {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = z;
  }
  f32x4 acct = {0.f, 0.f, 0.f, 0.f};

  // Two wave halves half a K tile apart (gemm256's idea on this tile): a wave alternates a LOAD part (all 16 + 4 fragments of the
  // K tile into registers) and a MULTIPLY part (its 16 + 2 MFMAs, the requests of a later tile between them), one workgroup
  // barrier after each.  Waves 4 - 7 ("late"; wave w + 4 shares a SIMD with wave w) take one extra barrier first, so while one
  // wave of a SIMD multiplies the other reads -- the matrix pipe does not idle through the barrier + LDS latency at the top of
  // every tile (the lockstep form kept it 54 % busy with NO memory traffic at all: profiles/r04_gemm288.txt).
  // Parts are numbered by barrier interval p.  Early: LOAD(t) at p = 2t, MULT(t) at 2t + 1; late: LOAD(t) at 2t + 1, MULT(t) at
  // 2t + 2.  Stage t % 3 is read last at p = 2t + 1, so tile t + 3 may be requested into it from p = 2t + 2 on:
  //   early MULT(t) (p = 2t + 1) requests its share of tile t + 2 (stage of t - 1, free since 2t);  lead 3 parts
  //   late  MULT(t) (p = 2t + 2) requests its share of tile t + 3 (stage of t, free since 2t + 2); lead 4 parts
  // and a tile must have landed before p = 2T: early waves wait for their share of T at the end of MULT(T - 1) (p = 2T - 1, the
  // share of T + 1 just requested stays in flight), late waves in LOAD(T - 1) (p = 2T - 1, their share of T + 1 stays in flight).
  const bool late = wave >= 4;
#define T_WAIT_SHARE(more)              \
do {                                  \
  if (!(more)) T_VMCNT(0);            \
  else if (tail_wave) T_VMCNT(7);     \
  else T_VMCNT(6);                    \
} while (0)
  request(kt0);
  if (kt0 + 1 < nk) request(kt0 + 1);
  if (late && kt0 + 2 < nk) request(kt0 + 2);
  if (!late) {
    T_WAIT_SHARE(kt0 + 1 < nk);
  } else {  // late waves carry no tail group: 6 requests per tile
    const int later = nk - 1 - kt0;
    if (later >= 2) T_VMCNT(12);
    else if (later == 1) T_VMCNT(6);
    else T_VMCNT(0);
  }
  T_BARRIER();
  if (late) T_BARRIER();
  for (int kt = kt0; kt < nk; ++kt) {
    // ---- LOAD part ----
    const char* buf = lds + (kt % T_NS) * T_STAGE;
    bf16x8 fa[4][2], fw[4][2], fta[2], ftw[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[ks][i] = *reinterpret_cast<const bf16x8*>(buf + a_base + i * 32 * 128 + koff[ks]);
        fw[ks][i] = *reinterpret_cast<const bf16x8*>(buf + w_base + i * 32 * 128 + koff[ks]);
      }
    if (has_tail) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        fta[s2] = *reinterpret_cast<const bf16x8*>(buf + taoff[s2]);
        ftw[s2] = *reinterpret_cast<const bf16x8*>(buf + toff[s2]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (late) T_WAIT_SHARE(kt + 2 < nk);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments land before the barrier: whoever passes it may overwrite the stage
    T_BARRIER();
    // ---- MULTIPLY part ----
    const int rq = late ? kt + 3 : kt + 2;
    const bool do_rq = rq < nk;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[2 * mi + jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mi], fw[ks][jn], acc[2 * mi + jn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (do_rq && ks < 3) request_part(rq, ks);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_tail) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fta[s2], ftw[s2], acct, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!late) T_WAIT_SHARE(kt + 2 < nk);
    T_BARRIER();
  }
  if (!late) T_BARRIER();  // pairs with the last barrier of the late half
#undef T_WAIT_SHARE

  // ---- epilogue.  32x32 D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
  if (swi) {
    // gate lives in the waves with wn = 0 (tail: waves 0 - 3), up in their partners: the bf16-rounded products change hands through
    // the LDS the K tiles no longer need (every wave has passed the last barrier: all stages are read and all requests have landed),
    // then each wave finishes half of the pair's tiles with silu_mul_kernel's arithmetic: rnd(rnd(silu(g)) * u)
    float* xw = reinterpret_cast<float*>(lds) + (size_t)wave * (4 * 16 + 4) * 64;  // [tile][reg][lane] + the tail's [reg][lane]
    T_BARRIER();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) xw[(t * 16 + r) * 64 + lane] = rnd<bf16_t>(acc[t][r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) xw[(64 + r) * 64 + lane] = rnd<bf16_t>(acct[r]);
    T_BARRIER();
    const float* xp = reinterpret_cast<const float*>(lds) + (size_t)(wave ^ 1) * (4 * 16 + 4) * 64;  // the partner (wm, wn ^ 1)
    bf16_t* C = reinterpret_cast<bf16_t*>(e.C);
    auto finish = [&](auto t_c, auto mine_is_gate) {  // static tile index: a run-time index would move the accumulators to scratch
      constexpr int t = decltype(t_c)::value;
      constexpr bool own_gate = decltype(mine_is_gate)::value;
      const int n = n0 + (t & 1) * 32 + l31;
      const int mb = m0 + wm * 64 + (t >> 1) * 32 + 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float own = rnd<bf16_t>(acc[t][r]), oth = xp[(t * 16 + r) * 64 + lane];
        const float gv = own_gate ? own : oth, uv = own_gate ? oth : own;
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < e.M) C[(size_t)m * e.ldc + n] = (bf16_t)(rnd<bf16_t>(silu(gv)) * uv);
      }
    };
    if (wn == 0) {  // wn = 0 finishes tiles 0, 1 of the pair, wn = 1 tiles 2, 3
      finish(std::integral_constant<int, 0>{}, std::true_type{});
      finish(std::integral_constant<int, 1>{}, std::true_type{});
    } else {
      finish(std::integral_constant<int, 2>{}, std::false_type{});
      finish(std::integral_constant<int, 3>{}, std::false_type{});
    }
    if (has_tail && wave < 4) {  // tail columns 16 w .. 16 w + 15: gate in wave w, up in wave w + 4
      const float* xt = reinterpret_cast<const float*>(lds) + (size_t)(wave + 4) * (4 * 16 + 4) * 64;
      const int n = n0 + wave * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + T_MAIN + 4 * g4 + r;
        if (m < e.M) C[(size_t)m * e.ldc + n] = (bf16_t)(rnd<bf16_t>(silu(rnd<bf16_t>(acct[r]))) * xt[(64 + r) * 64 + lane]);
      }
    }
    return;
  }
  float* slab = e.splits > 1 ? e.partial + (size_t)blockIdx.y * e.M * e.N : nullptr;
#pragma clang loop unroll(full)
  for (int t = 0; t < 4; ++t) {
    const f32x16 a = acc[t];
    const int n = n0 + wn * 64 + (t & 1) * 32 + l31;
    const int mb = m0 + wm * 64 + (t >> 1) * 32 + 4 * hi;
    if (slab) {
#pragma clang loop unroll(full)
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < e.M && n < e.N) slab[(size_t)m * e.N + n] = a[r];
      }
    } else {
      epilogue_tile32<bf16_t>(e, mb, n, a);
    }
  }
  // tail: 16x16 D layout: row = 4 * (lane >> 4) + reg, col = lane & 15
  if (has_tail) {
    const int n = n0 + wave * 16 + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + T_MAIN + 4 * g4 + r;
      if (slab) {
        if (m < e.M && n < e.N) slab[(size_t)m * e.N + n] = acct[r];
      } else {
        epilogue_store<bf16_t>(e, m, n, acct[r]);
      }
    }
  }
}

}  // namespace

// e.splits / e.tiles_per_split / e.partial are set by the caller (srgpt_gemm) when it wants split-K; the deterministic slab
// reduction (splitk_reduce_kernel in gemm.hip) follows there.
int srgpt_gemm288_launch(const void* A, const void* W, int K, int lda, const Epilogue& e, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)gemm_bf16_288_kernel, T_LDS));
  hipLaunchKernelGGL(gemm_bf16_288_kernel,
                     dim3(e.swiglu_inter ? e.swiglu_inter / (T_BN / 2) : cdiv(e.N, T_BN), e.splits > 1 ? e.splits : 1, cdiv(e.M, T_BM)),
                     dim3(512), T_LDS, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
