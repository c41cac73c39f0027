// bf16 MFMA GEMM, 256 x 256 x 64 block tile, 8 waves -- the kernel for shapes that fill the chip with 256^2 tiles
// (batched ViT / batched prefill / large square shapes).  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), both operands K-contiguous.
//
// Structure (MI355X: 1 block = 512 threads per CU, 2 waves per SIMD, 128 KiB of the 160 KiB LDS):
//   * waves 2 (M) x 4 (N); a wave owns 128 x 64 of C = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator VGPRs)
//   * LDS: two K-tile buffers x (A 256x64 + W 256x64) bf16, rows of 128 B in 16-byte slots, slot ^ ((row >> 1) & 7)
//     (the conflict-free ds_read_b128 fragment layout of gemm.hip), filled by LDS-DMA (global_load_lds_dwordx4: one
//     wave-instruction = 8 rows x 128 B = 1 KiB, full cache lines; the swizzle permutes which 16-byte chunk a lane fetches)
//   * a K tile is consumed in 2 PHASES, phase P = m-tiles 2P, 2P+1 of the wave (64 rows) x both n-tiles x K = 64 -> 16 MFMAs
//     on four accumulators (512 matrix-pipe cycles).  Measured (profiles/r02_gemm256_*.txt; the ablations were template arguments of this kernel while it was built): with 8-MFMA clusters on TWO
//     accumulators the MFMAs alone ran at 44 instead of 32 cycles each -- a dependent 32x32x16 MFMA two issue slots behind its
//     producer stalls.  The W fragments of the whole K tile are read in phase 0 and kept in 32 VGPRs, the A fragments of the
//     phase's two m-tiles in 32 more.  Every LDS region therefore has ONE reading phase (W and A rows 0,1: phase 0; A rows 2,3:
//     phase 1) and the next-but-one K tile is DMA'd into the buffer that is still being multiplied:
//         phase 0 of tile t: A rows of m-tiles 2,3 of tile t+1 (2 DMA instructions per thread)
//         phase 1 of tile t: W (lo, hi) and A rows of m-tiles 0,1 of tile t+2 (6 instructions)
//   * the two waves of a SIMD run ONE PHASE APART (waves 4-7 start one barrier late): while one wave issues its 16 MFMAs the
//     other reads fragments, so the matrix pipe always has a wave on it; the DMA instructions are issued BETWEEN the MFMAs of
//     the multiply part (in the matrix pipe's shadow), s_setprio 1 around the cluster
//   * DMA is never drained in the main loop: two counted waits per K tile -- vmcnt(6) in phase 0 (A rows 2,3 of THIS tile have
//     landed; the 6 of the previous phase 1 stay in flight) and vmcnt(2) in phase 1 (W + A rows 0,1 of the NEXT tile) -- each
//     placed one full phase (a workgroup barrier both wave groups have passed) before the first ds_read of that data; fragment
//     reads retire (lgkmcnt 0) BEFORE the barrier of their load part, so a region is free once both halves passed it; raw
//     s_barrier only (a __syncthreads() would carry vmcnt(0)).
//   * blocks are renumbered so that each XCD (private 4 MiB L2) owns a contiguous run of tiles, walked in 8-row bands.
// Requirements (checked by the launcher; everything else stays on gemm.hip's kernels): K % 64 == 0, K >= 256.
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_OPER = G_BM * G_BK * 2;   // bytes of one operand tile (32 KiB)
constexpr int G_BUF = 2 * G_OPER;         // one K-tile buffer: A then W (64 KiB)
constexpr int G_LDS = 2 * G_BUF;          // 128 KiB

#define G_BARRIER()                          \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)
#define G_VMCNT(N)                                                   \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)

// W8: the weight operand is OCP e4m3fn bytes [N, K] with one fp32 (power-of-two) scale per row (BASELINE config 5): the W tile is
// DMA'd as bytes (16 KiB per K tile, 16 rows x 64 B per wave-instruction, 16-byte chunk ^ ((row >> 2) & 3) -> conflict-free
// ds_read_b128), one 16-byte read = 16 consecutive k of a row = the B operands of TWO MFMAs after v_cvt_scalef32_pk_bf16_fp8
// (scale 1), so the A operand takes its 16-byte chunks in the matching order (chunk 4j + 2 hi + e for MFMA 2j + e);
// the row scale multiplies the accumulator in the epilogue (exact: the bf16 value of code * 2^k is code * 2^k).
// PERSIST (round 6): one block per CU walks a list of tiles (its XCD's run, strided by the blocks of that XCD).  At K = 1152 (the ViT
// and deconvolution products: 18 K tiles) a tile's K loop is ~9 us and its block launch + two-K-tile prologue + epilogue ~5 us
// more (profiles/r05_vit_batched_gemm.txt: 0.63 - 0.79 PF/s against 1.1 at K = 4096): here the prologue DMAs of tile t + 1 are
// issued right after tile t's last barrier -- every fragment read has retired, both K-tile buffers are free -- and land while tile
// t's epilogue stores go out; no block dispatch between tiles.
template <bool W8, bool PERSIST>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256_kernel(const bf16_t* __restrict__ A, const void* __restrict__ Wv,
                                                               int K, int lda, Epilogue e, int gx, int gy) {
  const bf16_t* W = reinterpret_cast<const bf16_t*>(Wv);
  const unsigned char* W8p = reinterpret_cast<const unsigned char*>(Wv);
  constexpr int NW = W8 ? 1 : 2;  // DMA instructions per thread for one half (128 rows) of the W tile
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- block -> tile: XCD-contiguous runs (bijective for any grid size), 8-row bands inside a run ----
  // PERSIST: the block walks tiles run_start + (bid >> 3) + j * (blocks per XCD) of its XCD's run (the grid is a multiple of 8)
  const int nwg = gx * gy, bid = blockIdx.x;
  const int xcd = bid & 7, runq = nwg >> 3, runr = nwg & 7;
  const int run_start = xcd < runr ? xcd * (runq + 1) : runr * (runq + 1) + (xcd - runr) * runq;
  const int run_len = runq + (xcd < runr ? 1 : 0);
  const int tstep = PERSIST ? (int)(gridDim.x >> 3) : run_len;  // (not persistent: one tile)
  int tpos = bid >> 3;                                          // position inside the run
  int m0, n0;
  auto tile_origin = [&](int pos) {
    const int wg = run_start + pos;
    const int band = wg / (8 * gx), idx = wg - band * 8 * gx;
    const int hb = min(8, gy - band * 8);
    m0 = (band * 8 + idx % hb) * G_BM;
    n0 = (idx / hb) * G_BN;
  };
  tile_origin(tpos);

  const int nk_all = K / G_BK;
  const int kt0 = e.splits > 1 ? (int)blockIdx.y * e.tiles_per_split : 0;
  const int nk = e.splits > 1 ? min(nk_all, kt0 + e.tiles_per_split) : nk_all;

  // ---- DMA source pointers: this lane's row of each 8-row group it stages, at its (swizzled) 16-byte chunk ----
  // group g = rows 8g .. 8g+7 of an operand tile; every group of this wave has g & 1 == wave & 1, so the chunk a lane
  // fetches (physical slot ^ ((row >> 1) & 7), row = 8g + lr) is the same for all of them
  const int lr = lane >> 3, lc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
  // slot 0: W groups w, w+8 | slot 1: W groups 16+w, 24+w | slot 2: A groups w, 16+w | slot 3: A groups 8+w, 24+w
  const bf16_t* pw[4];
  const bf16_t* pa[4];
  const unsigned char* pw8[2];  // fp8 W: group g = rows 16g .. 16g+15 (64 B each); this wave stages groups w and 8 + w
  auto set_sources = [&]() {  // for the tile at (m0, n0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gw_ = wave + 8 * i;                                  // W: 0..7, 8..15, 16..23, 24..31
      const int ga_ = (i & 1) * 16 + (i >> 1) * 8 + wave;            // A: w, 16+w, 8+w, 24+w
      pw[i] = W + (size_t)min(n0 + gw_ * 8 + lr, e.N - 1) * K + lc * 8;
      pa[i] = A + (size_t)min(m0 + ga_ * 8 + lr, e.M - 1) * lda + lc * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)  // lane -> row lane >> 2 of the group, physical chunk lane & 3; (row >> 2) & 3 = (lane >> 4) & 3
      pw8[i] = W8p + (size_t)min(n0 + (wave + 8 * i) * 16 + (lane >> 2), e.N - 1) * K + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  };
  set_sources();
  auto dma = [&](const void* src, int lds_off) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + lds_off), 16, 0, 0);
  };
  // stage slot `sl` (compile-time) of K tile kt into buffer kt & 1
  // which = 0 / 1: the first / second DMA instruction of the slot, 2: both
  auto stage = [&](auto sl_c, int kt, int which = 2) {
    constexpr int sl = decltype(sl_c)::value;
    const int k0 = kt * G_BK;
    const int boff = (kt & 1) * G_BUF;
    if constexpr (sl == 0 && W8) {
      if (which != 1) dma(pw8[0] + k0, boff + G_OPER + (wave + 0) * 1024);
    } else if constexpr (sl == 1 && W8) {
      if (which != 1) dma(pw8[1] + k0, boff + G_OPER + (wave + 8) * 1024);
    } else if constexpr (sl == 0) {
      if (which != 1) dma(pw[0] + k0, boff + G_OPER + (wave + 0) * 1024);
      if (which != 0) dma(pw[1] + k0, boff + G_OPER + (wave + 8) * 1024);
    } else if constexpr (sl == 1) {
      if (which != 1) dma(pw[2] + k0, boff + G_OPER + (wave + 16) * 1024);
      if (which != 0) dma(pw[3] + k0, boff + G_OPER + (wave + 24) * 1024);
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

  // ---- fragment addresses: row = tile row + (lane & 31), 16-byte slot = (2 ks + (lane >> 5)) ^ ((row >> 1) & 7);
  //      tile rows are multiples of 32, so the swizzle depends on the lane only ----
  const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
  const int frag0 = l31 * 128 + ((hi ^ (sw & 1)) << 4);   // + ((ks ^ (sw >> 1)) << 5) per k-step
  int koff[4];  // A (and bf16 W) fragment offset of MFMA ks within its 32-row tile
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if constexpr (W8) {  // chunk 4j + 2 hi + e for MFMA ks = 2j + e: the k order of one 16-byte fp8 read of the W row
      const int slot = 4 * (ks >> 1) + 2 * hi + (ks & 1);
      koff[ks] = l31 * 128 + ((slot ^ sw) << 4);
    } else {
      koff[ks] = frag0 + ((ks ^ (sw >> 1)) << 5);
    }
  }
  const int a_base = wr * 128 * 128;                       // + p * 32 * 128 per m-tile
  const int w_base = G_OPER + wc * 64 * (W8 ? 64 : 128);   // + jn * 32 rows per n-tile
  int koff8[2];  // fp8 W: 16-byte chunk (2j + hi) ^ ((row >> 2) & 3) of the 64-byte row
#pragma unroll
  for (int j = 0; j < 2; ++j) koff8[j] = l31 * 64 + (((2 * j + hi) ^ ((l31 >> 2) & 3)) << 4);

  f32x16 acc[4][2];
  bf16x8 fw[2][4], fa[2][4];
  u32x4 raw8[2][2];  // W8: the tile's fp8 W fragments as read from LDS
  const bool late = wr == 1;  // the late half runs one phase behind the early half; the two waves of a SIMD are in different halves

  // ---- prologue requests of a tile: K tile kt0 complete, W + A rows 0,1 of K tile kt0+1 (its A rows 2,3 are staged in phase 0 of kt0) ----
  auto prologue_requests = [&]() {
    stage(S0{}, kt0);
    stage(S1{}, kt0);
    stage(S2{}, kt0);
    stage(S3{}, kt0);
    if (kt0 + 1 < nk) {
      stage(S0{}, kt0 + 1);
      stage(S1{}, kt0 + 1);
      stage(S2{}, kt0 + 1);
    }
  };
  prologue_requests();
  bool first_tile = true;

  // Phase P (0 / 1) of K tile kt = m-tiles 2P, 2P+1 of the wave x both n-tiles x K = 64: 16 MFMAs on FOUR accumulators (the same
  // accumulator comes round every 4th MFMA: back-to-back dependent 32x32x16 MFMAs at distance 2 stall ~8 cycles each).
  //   load part:     fragment reads (P0: W of the tile + A m-tiles 0,1; P1: A m-tiles 2,3), counted DMA wait, lgkmcnt(0), barrier
  //   multiply part: 16 MFMAs with this phase's DMA instructions issued in their shadow, barrier
  // DMA plan (issue order per thread): P0 of tile t: A rows 2,3 of tile t+1 (2 instr) | P1 of tile t: W lo, W hi, A rows 0,1 of
  // tile t+2 (6 instr) into the buffer tile t occupies -- all of its readers retired their reads (lgkmcnt(0) BEFORE the
  // barrier of their P0 load part) at least one workgroup barrier earlier.  Waits: P0 needs A rows 2,3 of THIS tile one
  // phase later -> vmcnt(6) (the 6 of the previous P1 may fly); P1 needs W + A rows 0,1 of the NEXT tile one phase later
  // -> vmcnt(2) (the 2 of this tile's P0 may fly).
  auto phase = [&](auto p_c, int kt, bool more1, bool more2) {
#define G_WAIT(N) G_VMCNT(N)
    constexpr int P = decltype(p_c)::value;
    const char* buf = lds + (kt & 1) * G_BUF;
    {
      if constexpr (P == 0 && !W8) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            fw[jn][ks] = *reinterpret_cast<const bf16x8*>(buf + w_base + jn * 32 * 128 + koff[ks]);
      }
      if constexpr (P == 0 && W8) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
          for (int j = 0; j < 2; ++j) raw8[jn][j] = *reinterpret_cast<const u32x4*>(buf + w_base + jn * 32 * 64 + koff8[j]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          fa[mi][ks] = *reinterpret_cast<const bf16x8*>(buf + a_base + (2 * P + mi) * 32 * 128 + koff[ks]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P == 0) {
      if (more1) {
        if constexpr (W8) G_WAIT(4); else G_WAIT(6);
      } else {
        G_WAIT(0);
      }
    } else {
      if (more2) G_WAIT(2); else G_WAIT(0);
    }
    // fragments land before the barrier: whoever passes it may overwrite what this phase read
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (P == 0 && W8) {  // 16 fp8 -> two bf16x8 MFMA operands (k ascending: byte 0 of word 0 is the lowest k)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {  // v_cvt_scalef32_pk_bf16_fp8: two fp8 -> one packed bf16x2 register (scale 1, exact)
            u32x4 o;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
              const unsigned int raw = raw8[jn][j][2 * hf + qq];
              o[2 * qq] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw, 1.0f, false));
              o[2 * qq + 1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw, 1.0f, true));
            }
            fw[jn][2 * j + hf] = __builtin_bit_cast(bf16x8, o);
          }
    }
    G_BARRIER();
    // -------- multiply part --------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[2 * P + mi][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ks], fw[jn][ks], acc[2 * P + mi][jn], 0, 0, 0);
      // DMA in the matrix pipe's shadow: P0 one instruction after MFMA 4 and 12; P1 two after MFMA 4, 8, 12
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (P == 0) {
        if (more1 && (ks == 0 || ks == 2)) stage(S3{}, kt + 1, ks >> 1);
      } else {
        if (more2) {
          if (ks == 0) stage(S0{}, kt + 2, 2);
          if (ks == 1) stage(S1{}, kt + 2, 2);
          if (ks == 2) stage(S2{}, kt + 2, 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    G_BARRIER();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  for (;;) {
    {
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] = z;
        acc[i][1] = z;
      }
    }
    // K tile kt0 has landed (first tile: the 6 requests of kt0+1 may still fly; later tiles: the requests were issued in front of
    // the previous tile's epilogue stores -- vmcnt counts those too and only goes to 63, so everything is waited for: the stores
    // had the whole request latency to drain)
    if (first_tile && kt0 + 1 < nk) {
      if constexpr (W8) G_VMCNT(4); else G_VMCNT(6);
    } else {
      G_VMCNT(0);
    }
    first_tile = false;
    G_BARRIER();
    if (late) G_BARRIER();
    for (int kt = kt0; kt < nk; ++kt) {
      const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
      phase(P0{}, kt, more1, more2);
      phase(P1{}, kt, more1, more2);
    }
    if (!late) G_BARRIER();  // pairs with the last barrier of the late half: every fragment read of this tile has retired

    const int em0 = m0, en0 = n0;  // the finished tile's origin
    tpos += tstep;
    const bool more_tiles = PERSIST && tpos < run_len;
    if (more_tiles) {  // the next tile's first requests go out BEFORE this tile's stores
      tile_origin(tpos);
      set_sources();
      prologue_requests();
    }

    // ---- epilogue.  D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
#pragma clang loop unroll(full)
    for (int i = 0; i < 4; ++i)
#pragma clang loop unroll(full)
      for (int jn = 0; jn < 2; ++jn) {
        const f32x16 a = acc[i][jn];
        const int n = en0 + wc * 64 + jn * 32 + l31;
        const int mb = em0 + wr * 128 + i * 32 + 4 * hi;
        if (e.splits > 1) {
          float* slab = e.partial + (size_t)blockIdx.y * e.M * e.N;
#pragma clang loop unroll(full)
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < e.M && n < e.N) slab[(size_t)m * e.N + n] = a[r];
          }
        } else {
          epilogue_tile32<bf16_t>(e, mb, n, a);
        }
      }
    if (!more_tiles) break;
  }
}

}  // namespace

// e.splits / e.tiles_per_split / e.partial are set by the caller (srgpt_gemm) when it wants split-K; the deterministic
// slab reduction (splitk_reduce_kernel in gemm.hip) follows there.
int srgpt_gemm256_launch(const void* A, const void* W, int K, int lda, const Epilogue& e, hipStream_t s) {
  const int gx = cdiv(e.N, G_BN), gy = cdiv(e.M, G_BM);
  const bool w8 = e.wscale != nullptr;  // fp8 weight bytes + per-row scales
  // persistent form: more tiles than CUs and no split-K -- one block per CU (a multiple of 8: the XCD runs), each walks its tiles
  const int cus = srgpt_device_cus() & ~7;
  const bool persist = e.splits <= 1 && gx * gy > cus && cus >= 8 && SRGPT_KNOB("SRGPT_GEMM_PERSIST", 1) != 0;
#define G_LAUNCH(W8V, PV)                                                                                                 \
  do {                                                                                                                    \
    static std::atomic<uint64_t> attr_done{0};                                                                            \
    SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)gemm_bf16_256_kernel<W8V, PV>, G_LDS));                        \
    hipLaunchKernelGGL((gemm_bf16_256_kernel<W8V, PV>), dim3(PV ? cus : gx * gy, e.splits > 1 ? e.splits : 1), dim3(512), \
                       G_LDS, s, (const bf16_t*)A, W, K, lda, e, gx, gy);                                                 \
  } while (0)
  if (persist) {
    if (w8) G_LAUNCH(true, true); else G_LAUNCH(false, true);
  } else {
    if (w8) G_LAUNCH(true, false); else G_LAUNCH(false, false);
  }
#undef G_LAUNCH
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
