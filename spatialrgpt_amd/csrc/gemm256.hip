// bf16 MFMA GEMM, 256 x 256 x 64 block tile, 8 waves -- the kernel for shapes that fill the chip with 256^2 tiles
// (batched ViT / batched prefill / large square shapes).  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), both operands K-contiguous.
//
// Structure (MI355X: 1 block = 512 threads per CU, 2 waves per SIMD, 128 KiB of the 160 KiB LDS):
//   * waves 2 (M) x 4 (N); a wave owns 128 x 64 of C = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator VGPRs)
//   * LDS: two K-tile buffers x (A 256x64 + W 256x64) bf16, rows of 128 B in 16-byte slots, slot ^ ((row >> 1) & 7)
//     (the conflict-free ds_read_b128 fragment layout of gemm.hip), filled by LDS-DMA (global_load_lds_dwordx4: one
//     wave-instruction = 8 rows x 128 B = 1 KiB, full cache lines; the swizzle permutes which 16-byte chunk a lane fetches)
//   * a K tile is consumed in 4 PHASES, phase p = m-tile p of the wave (32 rows) x both n-tiles x K = 64 -> 8 MFMAs
//     (256 matrix-pipe cycles); the W fragments of the whole K tile are read in phase 0 and kept in 32 VGPRs, the A fragments
//     of m-tile p (16 VGPRs) in phase p.  So every LDS region has ONE reading phase: W after phase 0 and the A rows of m-tile
//     p after phase p are free, and the next-but-one K tile is DMA'd into the buffer that is still being multiplied:
//         phase 0: A rows of m-tiles 0,1 of tile t+1     phase 2: W rows   0..127 of tile t+2
//         phase 1: A rows of m-tiles 2,3 of tile t+1     phase 3: W rows 128..255 of tile t+2
//     (each "slot" = 16 KiB = 2 DMA instructions per thread, restaged >= 2 phases after its last ds_read)
//   * the two waves of a SIMD run ONE PHASE APART (waves 4-7 start one barrier late): while one wave issues its 8 MFMAs the
//     other reads fragments and issues DMA, so the matrix pipe always has a wave on it (s_setprio 1 around the MFMA cluster)
//   * DMA is never drained in the main loop: two counted waits per K tile -- vmcnt(6) in phase 0 (A rows of m-tiles 2,3 of
//     THIS tile have landed, 3 slots stay in flight) and vmcnt(4) in phase 2 (W + A rows 0,1 of the NEXT tile, 2 slots in
//     flight) -- each placed one full phase (a workgroup barrier both wave groups have passed) before the first ds_read of
//     that data; raw s_barrier only (a __syncthreads() would carry vmcnt(0)).
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

__global__ __launch_bounds__(512, 2) void gemm_bf16_256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                               int K, int lda, Epilogue e, int gx, int gy) {
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
  const int m0 = by * G_BM, n0 = bx * G_BN;

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
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gw_ = wave + 8 * i;                                  // W: 0..7, 8..15, 16..23, 24..31
    const int ga_ = (i & 1) * 16 + (i >> 1) * 8 + wave;            // A: w, 16+w, 8+w, 24+w
    pw[i] = W + (size_t)min(n0 + gw_ * 8 + lr, e.N - 1) * K + lc * 8;
    pa[i] = A + (size_t)min(m0 + ga_ * 8 + lr, e.M - 1) * lda + lc * 8;
  }
  auto dma = [&](const bf16_t* src, int lds_off) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + lds_off), 16, 0, 0);
  };
  // stage slot `sl` (compile-time) of K tile kt into buffer kt & 1
  auto stage = [&](auto sl_c, int kt) {
    constexpr int sl = decltype(sl_c)::value;
    const int k0 = kt * G_BK;
    const int boff = (kt & 1) * G_BUF;
    if constexpr (sl == 0) {
      dma(pw[0] + k0, boff + G_OPER + (wave + 0) * 1024);
      dma(pw[1] + k0, boff + G_OPER + (wave + 8) * 1024);
    } else if constexpr (sl == 1) {
      dma(pw[2] + k0, boff + G_OPER + (wave + 16) * 1024);
      dma(pw[3] + k0, boff + G_OPER + (wave + 24) * 1024);
    } else if constexpr (sl == 2) {
      dma(pa[0] + k0, boff + (wave + 0) * 1024);
      dma(pa[1] + k0, boff + (wave + 16) * 1024);
    } else {
      dma(pa[2] + k0, boff + (wave + 8) * 1024);
      dma(pa[3] + k0, boff + (wave + 24) * 1024);
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
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = frag0 + ((ks ^ (sw >> 1)) << 5);
  const int a_base = wr * 128 * 128;                       // + p * 32 * 128 per m-tile
  const int w_base = G_OPER + wc * 64 * 128;               // + jn * 32 * 128 per n-tile

  f32x16 acc[4][2];
  {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i][0] = z;
      acc[i][1] = z;
    }
  }
  bf16x8 fw[2][4], fa[4];

  // ---- prologue: tile kt0 complete, W of tile kt0+1 (its A rows are staged in phases 0 and 1 of tile kt0) ----
  stage(S0{}, kt0);
  stage(S1{}, kt0);
  stage(S2{}, kt0);
  stage(S3{}, kt0);
  if (kt0 + 1 < nk) {
    stage(S0{}, kt0 + 1);
    stage(S1{}, kt0 + 1);
    G_VMCNT(4);
  } else {
    G_VMCNT(0);
  }
  G_BARRIER();
  if (wr == 1) G_BARRIER();  // waves 4-7 run one phase behind waves 0-3

  auto phase = [&](auto p_c, int kt, bool more1, bool more2) {
    constexpr int p = decltype(p_c)::value;
    const char* buf = lds + (kt & 1) * G_BUF;
    // -------- load part: fragments of this phase, then the DMA slot of this phase --------
    if constexpr (p == 0) {
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          fw[jn][ks] = *reinterpret_cast<const bf16x8*>(buf + w_base + jn * 32 * 128 + koff[ks]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa[ks] = *reinterpret_cast<const bf16x8*>(buf + a_base + p * 32 * 128 + koff[ks]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (p == 0) {
      if (more1) {
        stage(S2{}, kt + 1);
        G_VMCNT(6);
      } else {
        G_VMCNT(0);
      }
    } else if constexpr (p == 1) {
      if (more1) stage(S3{}, kt + 1);
    } else if constexpr (p == 2) {
      if (more2) {
        stage(S0{}, kt + 2);
        G_VMCNT(4);
      } else {
        G_VMCNT(0);
      }
    } else {
      if (more2) stage(S1{}, kt + 2);
    }
    G_BARRIER();
    // -------- multiply part --------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
        acc[p][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fw[jn][ks], acc[p][jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    G_BARRIER();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;

  for (int kt = kt0; kt < nk; ++kt) {
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    phase(P0{}, kt, more1, more2);
    phase(P1{}, kt, more1, more2);
    phase(P2{}, kt, more1, more2);
    phase(P3{}, kt, more1, more2);
  }
  if (wr == 0) G_BARRIER();  // pairs with the last barrier of the late half

  // ---- epilogue.  D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
#pragma clang loop unroll(full)
  for (int i = 0; i < 4; ++i)
#pragma clang loop unroll(full)
    for (int jn = 0; jn < 2; ++jn) {
      const f32x16 a = acc[i][jn];
      const int n = n0 + wc * 64 + jn * 32 + l31;
      const int mb = m0 + wr * 128 + i * 32 + 4 * hi;
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
}

}  // namespace

// e.splits / e.tiles_per_split / e.partial are set by the caller (srgpt_gemm) when it wants split-K; the deterministic
// slab reduction (splitk_reduce_kernel in gemm.hip) follows there.
int srgpt_gemm256_launch(const void* A, const void* W, int K, int lda, const Epilogue& e, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)gemm_bf16_256_kernel, G_LDS));
  const int gx = cdiv(e.N, G_BN), gy = cdiv(e.M, G_BM);
  hipLaunchKernelGGL(gemm_bf16_256_kernel, dim3(gx * gy, e.splits > 1 ? e.splits : 1), dim3(512), G_LDS, s, (const bf16_t*)A,
                     (const bf16_t*)W, K, lda, e, gx, gy);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
