// GEMM family: C[M,N] = act(A[M,K] @ W[N,K]^T + bias) + residual      (see include/srgpt.h)
//
// bf16: LDS-tiled MFMA kernel (v_mfma_f32_32x32x16_bf16), 256 threads = 2x2 waves, BK = 64,
//       register-staged double buffering, XOR-swizzled 16-byte LDS slots: slot ^ ((row >> 1) & 7) -- with 128-byte rows the
//       16 lanes of a ds_read_b128 pass (16 consecutive rows, one slot) then cover all 64 banks exactly once (the row's
//       parity picks the 128-byte half, the swizzled slot the 16 bytes within it); slot ^ (row & 7) left rows r and r + 8
//       on the same banks: SQ_LDS_BANK_CONFLICT was 50 % of SQ_LDS_IDX_ACTIVE, fused epilogue (bias / activation / residual / deconv pixel-shuffle / fp32 out).
//       Both operands are K-contiguous (nn.Linear weight layout), so A and W tiles stage identically.
// fp32: plain LDS-tiled FMA kernel; exists for tight-tolerance parity runs of the same host path.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

int srgpt_gemm256_launch(const void* A, const void* W, int K, int lda, const Epilogue& e, hipStream_t s);  // gemm256.hip
int srgpt_gemm288_launch(const void* A, const void* W, int K, int lda, const Epilogue& e, hipStream_t s);  // gemm288.hip

namespace {

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
constexpr int BK = 64;  // bf16 elements per K tile = 8 slots of 16 B

#ifdef SRGPT_TUNING_KNOBS  // register-staged predecessor: only in the A/B build (make TUNING=1, SRGPT_GEMM_GLDS=0)
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bf16_mfma(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                      int K, int lda, Epilogue e) {
  constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 MFMA tiles per wave per dim
  constexpr int LA = BM / 32, LW = BN / 32;   // 16-byte slots each thread stages per tile
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (BM + BN) * BK];
  bf16_t* As = lds;
  bf16_t* Ws = lds + 2 * BM * BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int sc = tid & 7, sr = tid >> 3;  // staging slot column / row

  // two register sets: the global loads of K-tile t+2 are in flight while tile t is multiplied and tile t+1 sits
  // in the other set waiting for its LDS slot (prefetch distance = 2 tiles; one tile cannot cover HBM/L2 latency)
  u32x4 ra[2][LA], rw[2][LW];
  auto gload = [&](int kt, u32x4 (&da)[LA], u32x4 (&dw)[LW]) {
    const int k = kt * BK + sc * 8;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = m0 + sr + 32 * i;
      da[i] = (m < e.M && k < K) ? *reinterpret_cast<const u32x4*>(A + (size_t)m * lda + k) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int n = n0 + sr + 32 * i;
      dw[i] = (n < e.N && k < K) ? *reinterpret_cast<const u32x4*>(W + (size_t)n * K + k) : u32x4{0, 0, 0, 0};
    }
  };
  auto lstore = [&](int buf, const u32x4 (&da)[LA], const u32x4 (&dw)[LW]) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = sr + 32 * i;
      *reinterpret_cast<u32x4*>(As + (size_t)buf * BM * BK + r * BK + ((sc ^ ((r >> 1) & 7)) << 3)) = da[i];
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int r = sr + 32 * i;
      *reinterpret_cast<u32x4*>(Ws + (size_t)buf * BN * BK + r * BK + ((sc ^ ((r >> 1) & 7)) << 3)) = dw[i];
    }
  };

  f32x16 acc[TM][TN];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero16;

  const int nk_all = (K + BK - 1) / BK;
  const int kt0 = e.splits > 1 ? (int)blockIdx.z * e.tiles_per_split : 0;
  const int nk = e.splits > 1 ? min(nk_all, kt0 + e.tiles_per_split) : nk_all;

  auto compute = [&](int cur) {
    const bf16_t* as = As + (size_t)cur * BM * BK;
    const bf16_t* ws = Ws + (size_t)cur * BN * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fw[TN];
      const int slot = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / 2) + i * 32 + (lane & 31);
        fa[i] = *reinterpret_cast<const bf16x8*>(as + r * BK + ((slot ^ ((r >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / 2) + j * 32 + (lane & 31);
        fw[j] = *reinterpret_cast<const bf16x8*>(ws + r * BK + ((slot ^ ((r >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fw[j], acc[i][j], 0, 0, 0);
    }
  };

  // tile t lives in register set (t - kt0) & 1 until it is written to LDS buffer (t - kt0) & 1
  gload(kt0, ra[0], rw[0]);
  if (kt0 + 1 < nk) gload(kt0 + 1, ra[1], rw[1]);
  lstore(0, ra[0], rw[0]);
  __syncthreads();
  int kt = kt0;
  for (; kt + 1 < nk; kt += 2) {
    // even step: multiply LDS[0] (tile kt); set 0 is free -> fetch tile kt+2; stage tile kt+1 (set 1) into LDS[1]
    if (kt + 2 < nk) gload(kt + 2, ra[0], rw[0]);
    compute(0);
    lstore(1, ra[1], rw[1]);
    __syncthreads();
    // odd step: multiply LDS[1] (tile kt+1); set 1 is free -> fetch tile kt+3; stage tile kt+2 (set 0) into LDS[0]
    if (kt + 3 < nk) gload(kt + 3, ra[1], rw[1]);
    compute(1);
    if (kt + 2 < nk) lstore(0, ra[0], rw[0]);
    __syncthreads();
  }
  if (kt < nk) compute(0);  // odd number of tiles: the last one is already staged in LDS[0]

  // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma clang loop unroll(full)
  for (int i = 0; i < TM; ++i)
#pragma clang loop unroll(full)
    for (int j = 0; j < TN; ++j) {
      const f32x16 a = acc[i][j];
      const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
      if (e.splits > 1) {
        float* slab = e.partial + (size_t)blockIdx.z * e.M * e.N;
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

#endif  // SRGPT_TUNING_KNOBS

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel, direct-to-LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass, single LDS
// buffer (BM+BN) x 128 B -> 3 blocks per CU at 128x128, whose independent K loops overlap each other's barriers.
// A wave-instruction moves 8 rows x 128 B into 1 KiB of LDS at (wave-uniform base + lane * 16); the XOR slot swizzle of
// the register-staged kernel is kept by permuting WHICH 16-byte chunk of its row a lane fetches (chunk = slot ^ ((row >> 1) & 7)),
// so the fragment reads below are the same conflict-free ds_read_b128.  Rows past M / N re-read the last valid row
// (their outputs are never stored); a ragged last K tile (K % 64 != 0) is staged through registers with zero fill.
// ------------------------------------------------------------------------------------------------
// NBUF = 1: single buffer, two barriers per K tile, overlap comes from the other blocks of the CU (large grids).
// NBUF = 2: the next tile's LDS-DMA is issued before the current tile is multiplied, one barrier per K tile -- for grids that
//           leave a CU with only 1-2 blocks (the M = 259 prefill and ViT shapes), where each K step would otherwise pay a
//           full memory latency.
template <int BM, int BN, int NBUF>
__global__ __launch_bounds__(256, NBUF == 1 ? 3 : 2) void gemm_bf16_glds(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                         int K, int lda, Epilogue e) {
  // wave layout: 2 x 2 waves of (BM/2) x (BN/2) when BM is a multiple of 64; BM = 96 (three 32-row MFMA tiles -- 259 rows pad to
  // 288 instead of 320 and a tile step moves 17.8 instead of 23.4 bytes per kFLOP through the fill path that bounds this kernel):
  // 1 x 4 waves of 96 x (BN/4)
  constexpr int WAVES_M = BM % 64 == 0 ? 2 : 1, WAVES_N = 4 / WAVES_M;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // rows / columns of a wave's sub-tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM * 32 * WAVES_M == BM && TN * 32 * WAVES_N == BN && BM % 32 == 0 && BN % 32 == 0, "tile / wave layout");
  constexpr int TILE = (BM + BN) * BK;  // elements per stage buffer
  __shared__ __attribute__((aligned(1024))) bf16_t lds[NBUF * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = WAVES_M == 2 ? wave >> 1 : 0, wn = WAVES_M == 2 ? wave & 1 : wave;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // row within the 8-row group; logical chunk this lane fetches = physical slot ^ swz(row), swz(row) = (row >> 1) & 7
  // (group base rows are multiples of 8: (row >> 1) & 7 = ((wave & 1) << 2) | (lr >> 1) for every group of this wave)
  const int lr = lane >> 3, lc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));

  const int nk_all = (K + BK - 1) / BK;
  const int kt0 = e.splits > 1 ? (int)blockIdx.z * e.tiles_per_split : 0;
  const int nk = e.splits > 1 ? min(nk_all, kt0 + e.tiles_per_split) : nk_all;

  // per-lane row base pointers (clamped rows), one per 8-row group this wave stages
  constexpr int GA = BM / 32, GW = BN / 32;  // groups per wave (4 waves x 8 rows)
  const bf16_t* pa[GA];
  const bf16_t* pw[GW];
#pragma unroll
  for (int i = 0; i < GA; ++i) pa[i] = A + (size_t)min(m0 + (wave + 4 * i) * 8 + lr, e.M - 1) * lda + lc * 8;
#pragma unroll
  for (int i = 0; i < GW; ++i) pw[i] = W + (size_t)min(n0 + (wave + 4 * i) * 8 + lr, e.N - 1) * K + lc * 8;

  auto stage = [&](int kt, int buf) {
    bf16_t* As = lds + buf * TILE;
    bf16_t* Ws = As + BM * BK;
    const int k0 = kt * BK;
    if (k0 + BK <= K) {
#pragma unroll
      for (int i = 0; i < GA; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[i] + k0),
                                         (__attribute__((address_space(3))) void*)(As + (wave + 4 * i) * 8 * BK), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < GW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pw[i] + k0),
                                         (__attribute__((address_space(3))) void*)(Ws + (wave + 4 * i) * 8 * BK), 16, 0, 0);
    } else {  // ragged last tile: zero-filled through registers, same swizzled slots
      const bool ok = k0 + lc * 8 < K;
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        const u32x4 v = ok ? *reinterpret_cast<const u32x4*>(pa[i] + k0) : u32x4{0, 0, 0, 0};
        *reinterpret_cast<u32x4*>(As + ((wave + 4 * i) * 8 + lr) * BK + ((lane & 7) << 3)) = v;
      }
#pragma unroll
      for (int i = 0; i < GW; ++i) {
        const u32x4 v = ok ? *reinterpret_cast<const u32x4*>(pw[i] + k0) : u32x4{0, 0, 0, 0};
        *reinterpret_cast<u32x4*>(Ws + ((wave + 4 * i) * 8 + lr) * BK + ((lane & 7) << 3)) = v;
      }
    }
  };

  f32x16 acc[TM][TN];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero16;

  auto compute = [&](int buf) {
    const bf16_t* As = lds + buf * TILE;
    const bf16_t* Ws = As + BM * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fw[TN];
      const int slot = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wm * WTM + i * 32 + (lane & 31);
        fa[i] = *reinterpret_cast<const bf16x8*>(As + r * BK + ((slot ^ ((r >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = wn * WTN + j * 32 + (lane & 31);
        fw[j] = *reinterpret_cast<const bf16x8*>(Ws + r * BK + ((slot ^ ((r >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fw[j], acc[i][j], 0, 0, 0);
    }
  };
  if (NBUF == 1) {
    for (int kt = kt0; kt < nk; ++kt) {
      stage(kt, 0);
      __syncthreads();  // carries the vmcnt(0) that lands the LDS-DMA
      compute(0);
      __syncthreads();  // every wave is done reading before the next tile overwrites the buffer
    }
  } else {
    stage(kt0, 0);
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = (kt - kt0) & 1;
      __syncthreads();  // tile kt has landed (vmcnt(0)); every wave has finished tile kt-1, so the other buffer is free
      if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
      compute(cur);
    }
  }

  // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma clang loop unroll(full)
  for (int i = 0; i < TM; ++i)
#pragma clang loop unroll(full)
    for (int j = 0; j < TN; ++j) {
      const f32x16 a = acc[i][j];
      const int n = n0 + wn * WTN + j * 32 + (lane & 31);
      const int mb = m0 + wm * WTM + i * 32 + 4 * (lane >> 5);
      if (e.splits > 1) {
        float* slab = e.partial + (size_t)blockIdx.z * e.M * e.N;
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

// split-K second pass: fixed-order sum of the fp32 slabs, then the fused epilogue
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(Epilogue e) {
  const size_t total = (size_t)e.M * e.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float acc = 0.f;
    for (int z = 0; z < e.splits; ++z) acc += e.partial[(size_t)z * total + i];
    const int m = (int)(i / e.N), n = (int)(i - (size_t)m * e.N);
    epilogue_store<T>(e, m, n, acc);
  }
}

// The same pass, four columns per thread (N % 4 == 0, plain row-major output, bias / residual not periodic): the slabs are read
// with 16-byte loads, all splits of a thread in flight together (the scalar form above paid a 4-byte load and a 2-byte store per
// element: 11 us per reduction at M = 259, N = 4096), bias and residual as 8-byte loads, one 8-byte (bf16) / 16-byte (fp32) store.
// Same summation order, same rounding points.
template <typename T, int MAXS>
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(Epilogue e) {
  const size_t total4 = (size_t)e.M * e.N / 4;
  const int n4 = e.N / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    f32x4 p[MAXS];
#pragma unroll
    for (int z = 0; z < MAXS; ++z)
      p[z] = *reinterpret_cast<const f32x4*>(e.partial + (size_t)min(z, e.splits - 1) * (total4 * 4) + i * 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int z = 0; z < MAXS; ++z)
      if (z < e.splits) acc += p[z];
    const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = e.wscale ? acc[q] * e.wscale[n + q] : acc[q];
    if (e.bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += to_f(reinterpret_cast<const T*>(e.bias)[n + q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = rnd<T>(v[q]);  // nn.Linear / conv output is materialised in T
    if (e.act != SRGPT_ACT_NONE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = rnd<T>(apply_act<T>(v[q], e.act));
    }
    if (e.residual) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = rnd<T>(v[q] + to_f(reinterpret_cast<const T*>(e.residual)[(size_t)m * e.N + n + q]));
    }
    const size_t off = (size_t)m * e.ldc + n;
    if (e.out_f32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(e.C) + off) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (bf16_t)v[q];
      static_assert(sizeof(T) == 2, "splitk_reduce4_kernel stores bf16 or fp32");
      *reinterpret_cast<bf16x4*>(reinterpret_cast<T*>(e.C) + off) = o;
    }
  }
}

// The same pass with the norm of the finished rows behind it (LlamaDecoderLayer: o_proj / down_proj + residual, then the next
// RMSNorm -- modeling_llama.py:611-684; the ViT encoder layer: out_proj / fc2 + residual, then the next LayerNorm): one block per
// row, thread t owns the 8-element chunks t, t + 256, ... exactly as rmsnorm_kernel / layernorm_kernel (norm.hip) do, sums the slabs
// in slab order, applies the epilogue with reduce4's roundings, stores the row of C, and runs the norm's statistics over the
// ROUNDED values in the norm kernels' element order through the same block_sum -- so C and the normalised row are bit-identical to
// [splitk_reduce4_kernel -> rmsnorm_kernel / layernorm_kernel], one launch and one read of the row less.
template <int MAXS, int CH>  // CH chunks of 8 per thread: N <= 2048 * CH
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(Epilogue e) {
  __shared__ float red[16];
  const int m = blockIdx.x, nch = e.N / 8;
  const size_t total = (size_t)e.M * e.N;
  float x[CH][8];
  float ssq = 0.f, sum = 0.f;  // rmsnorm_kernel's / layernorm_kernel's first accumulation, in their element order
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < nch) {
      const int n = c * 8;
      f32x4 p[MAXS][2];
#pragma unroll
      for (int z = 0; z < MAXS; ++z) {
        const float* src = e.partial + (size_t)min(z, e.splits - 1) * total + (size_t)m * e.N + n;
        p[z][0] = *reinterpret_cast<const f32x4*>(src);
        p[z][1] = *reinterpret_cast<const f32x4*>(src + 4);
      }
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int z = 0; z < MAXS; ++z)
        if (z < e.splits) {
          a0 += p[z][0];
          a1 += p[z][1];
        }
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = a0[q];
        v[4 + q] = a1[q];
      }
      if (e.wscale) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= e.wscale[n + q];
      }
      if (e.bias) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += to_f(reinterpret_cast<const bf16_t*>(e.bias)[n + q]);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = rnd<bf16_t>(v[q]);
      if (e.act != SRGPT_ACT_NONE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = rnd<bf16_t>(apply_act<bf16_t>(v[q], e.act));
      }
      if (e.residual) {
        const Vec16<bf16_t> r = *reinterpret_cast<const Vec16<bf16_t>*>(reinterpret_cast<const bf16_t*>(e.residual) + (size_t)m * e.N + n);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = rnd<bf16_t>(v[q] + r.get(q));
      }
      Vec16<bf16_t> o;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        o.set(q, v[q]);
        x[k][q] = v[q];
        ssq += v[q] * v[q];
        sum += v[q];
      }
      *reinterpret_cast<Vec16<bf16_t>*>(reinterpret_cast<bf16_t*>(e.C) + (size_t)m * e.ldc + n) = o;
    }
  }
  if (e.norm_kind == SRGPT_NORM_RMS) {
    const float r = rsqrtf(block_sum(ssq, red) / (float)e.N + e.norm_eps);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = threadIdx.x + 256 * k;
      if (c < nch) {
        const Vec16<bf16_t> g = *reinterpret_cast<const Vec16<bf16_t>*>(reinterpret_cast<const bf16_t*>(e.norm_w) + c * 8);
        Vec16<bf16_t> o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.set(q, g.get(q) * rnd<bf16_t>(x[k][q] * r));  // weight * h.to(input_dtype)
        *reinterpret_cast<Vec16<bf16_t>*>(reinterpret_cast<bf16_t*>(e.norm_y) + (size_t)m * e.N + c * 8) = o;
      }
    }
  } else {  // LayerNorm: layernorm_kernel's three passes over the row, from registers
    const float mean = block_sum(sum, red) / (float)e.N;
    float qv = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = threadIdx.x + 256 * k;
      if (c < nch) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float d = x[k][q] - mean;
          qv += d * d;
        }
      }
    }
    const float rstd = rsqrtf(block_sum(qv, red) / (float)e.N + e.norm_eps);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = threadIdx.x + 256 * k;
      if (c < nch) {
        const Vec16<bf16_t> g = *reinterpret_cast<const Vec16<bf16_t>*>(reinterpret_cast<const bf16_t*>(e.norm_w) + c * 8);
        const Vec16<bf16_t> be = *reinterpret_cast<const Vec16<bf16_t>*>(reinterpret_cast<const bf16_t*>(e.norm_b) + c * 8);
        Vec16<bf16_t> o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.set(q, (x[k][q] - mean) * rstd * g.get(q) + be.get(q));
        *reinterpret_cast<Vec16<bf16_t>*>(reinterpret_cast<bf16_t*>(e.norm_y) + (size_t)m * e.N + c * 8) = o;
      }
    }
  }
}

// The slab reduction of the prefill's q/k/v projection with rope_kv_append_kernel (misc.hip) behind it: one block per token row, the
// reduced row (rounded to bf16 as the plain reduction stores it) goes to LDS, then every (head, pair) is rotated with
// rope_kv_append_kernel's arithmetic -- q back into the q/k/v buffer, k into the cache at the token's position -- and v is copied
// into the cache.  The un-rotated k and the v columns are stored to the q/k/v buffer as well, so every byte the two launches write
// is the same (tests compare all three buffers).
template <int MAXS>
__global__ __launch_bounds__(256) void splitk_reduce_rope_kernel(Epilogue e) {
  extern __shared__ __attribute__((aligned(16))) bf16_t rrow[];
  const int m = blockIdx.x, nch = e.N / 8;
  const size_t total = (size_t)e.M * e.N;
  const int D = e.rope_D, half = D >> 1, Hq = e.rope_Hq, Hkv = e.rope_Hkv;
  for (int c = threadIdx.x; c < nch; c += 256) {
    const int n = c * 8;
    f32x4 p[MAXS][2];
#pragma unroll
    for (int z = 0; z < MAXS; ++z) {
      const float* src = e.partial + (size_t)min(z, e.splits - 1) * total + (size_t)m * e.N + n;
      p[z][0] = *reinterpret_cast<const f32x4*>(src);
      p[z][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int z = 0; z < MAXS; ++z)
      if (z < e.splits) {
        a0 += p[z][0];
        a1 += p[z][1];
      }
    Vec16<bf16_t> o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o.set(q, a0[q]);
      o.set(4 + q, a1[q]);
    }
    *reinterpret_cast<Vec16<bf16_t>*>(rrow + n) = o;
    if (n >= Hq * D) *reinterpret_cast<Vec16<bf16_t>*>(reinterpret_cast<bf16_t*>(e.C) + (size_t)m * e.ldc + n) = o;  // k (un-rotated), v
  }
  __syncthreads();
  const int b = m / e.rope_T, t = m - b * e.rope_T;
  const int pos = (e.rope_pos0 ? e.rope_pos0[b] : 0) + t;
  const bf16_t* cos_tab = reinterpret_cast<const bf16_t*>(e.rope_cos);
  const bf16_t* sin_tab = reinterpret_cast<const bf16_t*>(e.rope_sin);
  bf16_t* qrow = reinterpret_cast<bf16_t*>(e.C) + (size_t)m * e.ldc;
  const int nrot = (Hq + Hkv) * half;
  for (int w = threadIdx.x; w < nrot; w += 256) {
    const int h = w / half, i = w - h * half;
    const float c = to_f(cos_tab[(size_t)pos * half + i]), s = to_f(sin_tab[(size_t)pos * half + i]);
    const float x1 = to_f(rrow[h * D + i]), x2 = to_f(rrow[h * D + i + half]);
    // q*cos + rotate_half(q)*sin with every intermediate materialised in bf16
    const float o1 = rnd<bf16_t>(rnd<bf16_t>(x1 * c) + rnd<bf16_t>(-x2 * s));
    const float o2 = rnd<bf16_t>(rnd<bf16_t>(x2 * c) + rnd<bf16_t>(x1 * s));
    bf16_t* dst = h < Hq ? qrow + (size_t)h * D
                         : reinterpret_cast<bf16_t*>(e.rope_k) + (((size_t)b * Hkv + (h - Hq)) * e.rope_max_pos + pos) * D;
    dst[i] = from_f<bf16_t>(o1);
    dst[i + half] = from_f<bf16_t>(o2);
  }
  const bf16_t* vsrc = rrow + (size_t)(Hq + Hkv) * D;
  bf16_t* vc = reinterpret_cast<bf16_t*>(e.rope_v);
  for (int w = threadIdx.x; w < Hkv * D; w += 256) {
    const int h = w / D, i = w - h * D;
    vc[(((size_t)b * Hkv + h) * e.rope_max_pos + pos) * D + i] = vsrc[w];
  }
}

// launches the reduction that fits the epilogue (returns through SRGPT_LAUNCH_CHECK at the call site); true when the RMSNorm the
// epilogue asks for went into it
template <typename T>
static inline bool launch_splitk_reduce(const Epilogue& e, hipStream_t s) {
  const size_t total = (size_t)e.M * e.N;
  const bool vec4 = e.N % 4 == 0 && e.ldc % 4 == 0 && e.out_mode != SRGPT_OUT_DECONV2X && e.bias_mod <= 0 && e.res_mod <= 0 &&
                    e.splits <= 8 && ((uintptr_t)e.C % 16 == 0);
  if (e.rope_k && std::is_same<T, bf16_t>::value && vec4 && !e.out_f32 && !e.bias && !e.residual && !e.wscale &&
      e.act == SRGPT_ACT_NONE && e.N % 8 == 0 && e.ldc % 8 == 0 && e.N <= 24576) {
    const size_t lds = (size_t)e.N * sizeof(bf16_t);  // <= 48 KB
    if (e.splits <= 4) hipLaunchKernelGGL((splitk_reduce_rope_kernel<4>), dim3(e.M), dim3(256), lds, s, e);
    else hipLaunchKernelGGL((splitk_reduce_rope_kernel<8>), dim3(e.M), dim3(256), lds, s, e);
    return true;
  }
  if (e.norm_y && std::is_same<T, bf16_t>::value && vec4 && !e.out_f32 && e.N % 8 == 0 && e.ldc % 8 == 0 && e.N <= 8192 &&
      ((uintptr_t)e.norm_y % 16 == 0) && ((uintptr_t)e.norm_w % 16 == 0) && (!e.norm_b || (uintptr_t)e.norm_b % 16 == 0) &&
      (!e.residual || (uintptr_t)e.residual % 16 == 0)) {
    const bool wide = e.N > 4096;
    if (e.splits <= 4) {
      if (wide) hipLaunchKernelGGL((splitk_reduce_norm_kernel<4, 4>), dim3(e.M), dim3(256), 0, s, e);
      else hipLaunchKernelGGL((splitk_reduce_norm_kernel<4, 2>), dim3(e.M), dim3(256), 0, s, e);
    } else {
      if (wide) hipLaunchKernelGGL((splitk_reduce_norm_kernel<8, 4>), dim3(e.M), dim3(256), 0, s, e);
      else hipLaunchKernelGGL((splitk_reduce_norm_kernel<8, 2>), dim3(e.M), dim3(256), 0, s, e);
    }
    return true;
  }
  if (vec4) {
    int rgrid = (int)((total / 4 + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    if (e.splits <= 4) hipLaunchKernelGGL((splitk_reduce4_kernel<T, 4>), dim3(rgrid), dim3(256), 0, s, e);
    else hipLaunchKernelGGL((splitk_reduce4_kernel<T, 8>), dim3(rgrid), dim3(256), 0, s, e);
    return false;
  }
  int rgrid = (int)((total + 255) / 256);
  if (rgrid > 2048) rgrid = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(rgrid), dim3(256), 0, s, e);
  return false;
}

// ------------------------------------------------------------------------------------------------
// fp32 FMA kernel (parity path): 64x64 tile, BK 16, each thread 4x4 outputs
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_simple(const float* __restrict__ A, const float* __restrict__ W,
                                                       int K, int lda, Epilogue e) {
  constexpr int BM = 64, BN = 64, BKF = 16;
  __shared__ float As[BKF][BM + 4];
  __shared__ float Ws[BKF][BN + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  float acc[4][4] = {};
  const int lr = tid >> 2, lc = (tid & 3) * 4;  // 64 rows x 4 float4 per tile
  for (int k0 = 0; k0 < K; k0 += BKF) {
    {
      const int m = m0 + lr, k = k0 + lc;
      f32x4 v = {0, 0, 0, 0};
      if (m < e.M && k < K) v = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) As[lc + i][lr] = v[i];
      const int n = n0 + lr;
      f32x4 u = {0, 0, 0, 0};
      if (n < e.N && k < K) u = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) Ws[lc + i][lr] = u[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BKF; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Ws[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) epilogue_store<float>(e, m0 + ty * 4 + i, n0 + tx * 4 + j, acc[i][j]);
}

// ------------------------------------------------------------------------------------------------
// fp8-weight fallback for K that is not a multiple of 64 (tiny test geometries): 64x64 tile, BK 16, scalar loads, fp32 FMA
// on the exactly widened operands -- correctness first, any K
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_w8_simple(const bf16_t* __restrict__ A, const unsigned char* __restrict__ W8,
                                                      int K, int lda, Epilogue e) {
  constexpr int BM = 64, BN = 64, BKF = 16;
  __shared__ float As[BKF][BM + 4];
  __shared__ float Ws[BKF][BN + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  float acc[4][4] = {};
  const int lr = tid >> 2, lc = (tid & 3) * 4;  // 64 rows x 4 groups of 4 k per tile
  for (int k0 = 0; k0 < K; k0 += BKF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + lc + i;
      const int m = m0 + lr, n = n0 + lr;
      As[lc + i][lr] = (m < e.M && k < K) ? to_f(A[(size_t)m * lda + k]) : 0.f;
      float wv = 0.f;
      if (n < e.N && k < K) {
        const int code = W8[(size_t)n * K + k];
        wv = __builtin_amdgcn_cvt_pk_f32_fp8(code, false)[0];
      }
      Ws[lc + i][lr] = wv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BKF; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Ws[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) epilogue_store<bf16_t>(e, m0 + ty * 4 + i, n0 + tx * 4 + j, acc[i][j]);
}

}  // namespace

// the split-K slab reduction + epilogue for kernels in other files (gemm_f8.hip)
int srgpt_splitk_reduce_bf16(const Epilogue& e, hipStream_t s) {
  launch_splitk_reduce<bf16_t>(e, s);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int64_t srgpt_gemm_ws_bytes(int M, int N) { return (int64_t)8 * M * N * 4; }

// norm_w / norm_y / norm_eps: the RMSNorm of the output rows (srgpt_gemm_rmsnorm below); *norm_done = it went into the reduction
static int gemm_impl(const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K, int lda,
                     int ldc, int act, int bias_mod, int res_mod, int out_f32, int out_mode, int gw, void* ws, int64_t ws_bytes,
                     int dtype, srgpt_stream_t stream, int norm_kind, const void* norm_w, const void* norm_b, void* norm_y, float norm_eps,
                     bool* norm_done, const Epilogue* rope = nullptr) {
  SRGPT_CHECK(A && W && C, SRGPT_ERR_ARG, "srgpt_gemm: null pointer");
  SRGPT_CHECK(M > 0 && N > 0 && K > 0, SRGPT_ERR_ARG, "srgpt_gemm: bad shape M=%d N=%d K=%d", M, N, K);
  SRGPT_CHECK(dtype == SRGPT_F32 || dtype == SRGPT_BF16, SRGPT_ERR_ARG, "srgpt_gemm: bad dtype %d", dtype);
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(K % vec == 0 && lda % vec == 0, SRGPT_ERR_ARG,
              "srgpt_gemm: K=%d and lda=%d must be multiples of %d (16-byte rows)", K, lda, vec);
  SRGPT_CHECK(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), SRGPT_ERR_ARG, "srgpt_gemm: A/W must be 16-byte aligned");
  if (out_mode == SRGPT_OUT_DECONV2X) {
    SRGPT_CHECK(N % 4 == 0 && gw > 0 && M % (gw * gw) == 0, SRGPT_ERR_ARG, "srgpt_gemm: bad deconv geometry");
  } else {
    SRGPT_CHECK(out_mode == SRGPT_OUT_PLAIN, SRGPT_ERR_ARG, "srgpt_gemm: unknown out_mode %d", out_mode);
    SRGPT_CHECK(ldc >= N, SRGPT_ERR_ARG, "srgpt_gemm: ldc < N");
  }
  Epilogue e{bias, residual, C, M, N, ldc, act, bias_mod, res_mod, out_f32, out_mode, gw, nullptr, 1, 0, nullptr, norm_w, norm_b, norm_y, norm_eps, norm_kind};
  if (rope) {
    e.rope_k = rope->rope_k, e.rope_v = rope->rope_v, e.rope_pos0 = rope->rope_pos0, e.rope_cos = rope->rope_cos;
    e.rope_sin = rope->rope_sin, e.rope_T = rope->rope_T, e.rope_Hq = rope->rope_Hq, e.rope_Hkv = rope->rope_Hkv;
    e.rope_D = rope->rope_D, e.rope_max_pos = rope->rope_max_pos;
  }
  hipStream_t s = as_stream(stream);
  if (dtype == SRGPT_F32) {
    dim3 grid(cdiv(N, 64), cdiv(M, 64));
    hipLaunchKernelGGL(gemm_f32_simple, grid, dim3(256), 0, s, (const float*)A, (const float*)W, K, lda, e);
    SRGPT_LAUNCH_CHECK();
    return SRGPT_OK;
  }
  // ---- 288 x 128 whole-M kernel (gemm288.hip) for the bs = 1 prefill products (224 < M <= 272): W crosses the global -> LDS path
  //      once, requests three K tiles deep.  Column tiles x K splits should come to about one block per CU.
  {
    const int cus = srgpt_device_cus();
    const int nk = K / 64;
    const int gx = cdiv(N, 128);
    const int mode = SRGPT_KNOB("SRGPT_GEMM_288", 1);  // tuning build: 0 = never, 2 = for any M <= 272, > 2 = forced split count
    // measured at M = 259 (profiles/r04_gemm288.txt): gate/up 106.5 -> 84.2 us, down 64.8 -> 50.6, q/k/v 33.8 -> 32.0; o (27.5 vs
    // 27.9: 8 K tiles per block once K is split for 256 CUs, three of them pipeline fill) stays on the small tiles
    const int anym = SRGPT_KNOB("SRGPT_GEMM_288_ANYM", 0);  // tuning build (round 5 probe): several 272-row tiles (grid.z) for M > 272
    bool use288 = mode != 0 && K % 64 == 0 && nk >= 4 &&
                  ((M > 224 && M <= 272 && (int64_t)N * K >= (int64_t)24 << 20) || (mode >= 2 && (M <= 272 || anym)));
    int sp = 1;
    if (use288) {
      const int gz = cdiv(M, 272);
      if (gx * gz < cus * 3 / 4 && ws) {
        sp = (cus + gx * gz / 2) / (gx * gz);
        if (sp > nk / 8) sp = nk / 8;  // keep >= 8 K tiles per split: three of them are pipeline fill
        if (sp > 8) sp = 8;
        while (sp > 1 && (int64_t)sp * M * N * 4 > ws_bytes) --sp;
        if (sp < 1) sp = 1;
      }
      if (mode > 2 && ws) {
        sp = mode - 2;
        if (sp > nk) sp = nk;
        while (sp > 1 && (int64_t)sp * M * N * 4 > ws_bytes) --sp;
      }
      if ((long)gx * gz * sp < cus / 2) use288 = mode >= 2;  // too few blocks to fill the chip: the small tiles overlap better
    }
    if (use288) {
      if (sp > 1) {
        e.partial = reinterpret_cast<float*>(ws);
        e.tiles_per_split = cdiv(nk, sp);
        e.splits = cdiv(nk, e.tiles_per_split);
      }
      SRGPT_TRY(srgpt_gemm288_launch(A, W, K, lda, e, s));
      if (e.splits > 1) {
        {
          const bool fused = launch_splitk_reduce<bf16_t>(e, s);
          if (norm_done) *norm_done = fused;
        }
        SRGPT_LAUNCH_CHECK();
      }
      return SRGPT_OK;
    }
  }
  // ---- 256 x 256 eight-wave kernel (gemm256.hip); rule calibrated on MI355X measurements (profiles/r02_gemm256_*.txt,
  //      profiles/r02_gemm_final.txt: one block per CU, ~15 us of launch + prologue + epilogue per round of tiles) ----
  //   K >= 2048: it wins or ties on every shape with M >= 384 (prefill b8 qkv 111 vs 172 us, down 334 vs 455, b4 down 140 vs 212);
  //              an under-filled grid splits K (deterministic slabs) up to ~1.1 rounds of blocks
  //   K <  2048: the fixed cost per round is a quarter of the tile time, so only when the last round is nearly full
  //              (>= 88 %: ViT out-proj 50 vs 62 us; ViT qkv / fc1 at 84 / 76 % stay on the small-tile kernel: 131 vs 143 us)
  {
    const int cus = srgpt_device_cus();
    const int nk256 = K / 64;
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    // M >= 384 and at most 25 % of padded rows (M = 518 would fill 3 row tiles to 67 %)
    bool use256 = K % 64 == 0 && K >= 256 && M >= 384 && (long)M * 4 >= (long)cdiv(M, 256) * 256 * 3;
    int sp = 1;
    if (use256) {
      if (K >= 2048) {
        if (t256 < cus && ws) {
          sp = (int)((cus * 11 / 10 + t256 / 2) / t256);
          if (sp < 1) sp = 1;
          if (sp > 4) sp = 4;
          while (sp > 1 && (nk256 / sp < 8 || (int64_t)sp * M * N * 4 > ws_bytes)) --sp;
        }
      } else {
        const long rounds = (t256 + cus - 1) / cus;
        use256 = t256 * 100 >= rounds * cus * 88;
      }
    }
    const int f256 = SRGPT_KNOB("SRGPT_GEMM_FORCE_256", 0);  // tuning build: 1 = whenever legal, -1 = never
    if (f256 > 0) use256 = K % 64 == 0 && K >= 128, sp = 1;
    if (f256 < 0) use256 = false;
    if (use256) {
      if (sp > 1) {
        e.partial = reinterpret_cast<float*>(ws);
        e.tiles_per_split = cdiv(nk256, sp);
        e.splits = cdiv(nk256, e.tiles_per_split);
      }
      SRGPT_TRY(srgpt_gemm256_launch(A, W, K, lda, e, s));
      if (e.splits > 1) {
        {
          const bool fused = launch_splitk_reduce<bf16_t>(e, s);
          if (norm_done) *norm_done = fused;
        }
        SRGPT_LAUNCH_CHECK();
      }
      return SRGPT_OK;
    }
  }
  // ---- tile / split-K selection: fill the 256 CUs with >= ~2 blocks each ----
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
  const long t64x128 = (long)cdiv(M, 64) * cdiv(N, 128);
  const int nk = cdiv(K, BK);
  int bm = 128, splits = 1;
  long tiles = t128;
  const bool pad_waste = (long)cdiv(M, 128) * 128 * 10 > (long)cdiv(M, 64) * 64 * 11;  // > 10 % fewer padded rows with BM = 64
  // 128x128 only when it alone fills the chip at 4 blocks per CU; below that 64x128 has twice the blocks to overlap
  // (measured with the direct-to-LDS kernel: M=1458 N=4304 K=1152: 32.5 us vs 46.1 us; equal at 4096^3)
  if (t128 < 1024 || pad_waste) {
    bm = 64;
    tiles = t64x128;
    // 96-row tiles (three 32-row MFMA tiles per wave, 1 x 4 waves) where they do not pad M by more than 8 % over 64-row tiles:
    // this kernel is bound by the global -> LDS fill rate, and a 96x128 step moves 17.8 B/kFLOP against 23.4 (M = 259 pads to
    // 288 instead of 320: gate/up 156 -> 106 us, ViT fc1 38 -> 25 us, profiles/r02_gemm_bm96.txt).  The split count keeps
    // following the 64-row tile count (the measured configuration).
    if (M > 64 && (long)cdiv(M, 96) * 96 * 100 <= (long)cdiv(M, 64) * 64 * 108) bm = 96;
    if (tiles < 384 && ws) {
      splits = (int)((512 + tiles - 1) / tiles);
      if (splits > nk / 8) splits = nk / 8;  // keep >= 8 K-tiles (512 columns of K) per split
      if (splits > 8) splits = 8;
      while (splits > 1 && (int64_t)splits * M * N * 4 > ws_bytes) --splits;
      if (splits < 1) splits = 1;
    }
  }
  {  // tuning knobs (scripts/experiments/ubench_gemm.py sweeps them); unset in production
    const int f_bm = SRGPT_KNOB("SRGPT_GEMM_FORCE_BM", 0);
    const int f_sp = SRGPT_KNOB("SRGPT_GEMM_FORCE_SPLITS", 0);
    if (f_bm == 64 || f_bm == 128 || f_bm == 96) bm = f_bm;
    if (f_sp > 0 && ws) {
      splits = f_sp;
      if (splits > nk) splits = nk;
      while (splits > 1 && (int64_t)splits * M * N * 4 > ws_bytes) --splits;
    }
    if (bm == 128) splits = 1;
  }
  if (splits > 1) {
    e.partial = reinterpret_cast<float*>(ws);
    e.tiles_per_split = cdiv(nk, splits);
    splits = cdiv(nk, e.tiles_per_split);  // no empty split
    e.splits = splits;
  }
  const int use_glds = SRGPT_KNOB("SRGPT_GEMM_GLDS", 1);  // A/B knob (tuning build only)
  (void)use_glds;
  if (bm == 128) {
    dim3 grid(cdiv(N, 128), cdiv(M, 128), 1);
#ifdef SRGPT_TUNING_KNOBS
    if (!use_glds)
      hipLaunchKernelGGL((gemm_bf16_mfma<128, 128>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
    else
#endif
      hipLaunchKernelGGL((gemm_bf16_glds<128, 128, 1>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
  } else if (bm == 96) {
    dim3 grid(cdiv(N, 128), cdiv(M, 96), e.splits);  // (96x256 tiles measured: slower on every shape)
    const int f_nbuf = SRGPT_KNOB("SRGPT_GEMM_FORCE_NBUF", 0);
    const long blocks = (long)grid.x * grid.y * e.splits;
    // single buffer (5 blocks per CU overlap each other's K steps) only for un-split grids of >= 2 blocks per CU (gate/up 672,
    // ViT fc1 544 blocks: 106 vs 134 us, 25 vs 33 us); split-K and smaller grids double-buffer (q/k/v 34 vs 38 us)
    const bool dbuf = f_nbuf ? f_nbuf == 2 : !(e.splits <= 1 && blocks >= 2L * srgpt_device_cus());
    if (dbuf)
      hipLaunchKernelGGL((gemm_bf16_glds<96, 128, 2>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
    else
      hipLaunchKernelGGL((gemm_bf16_glds<96, 128, 1>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
  } else {
    dim3 grid(cdiv(N, 128), cdiv(M, 64), e.splits);
    const int f_nbuf = SRGPT_KNOB("SRGPT_GEMM_FORCE_NBUF", 0);
    const long blocks = tiles * e.splits;
    const bool dbuf = f_nbuf ? f_nbuf == 2 : blocks < 3L * srgpt_device_cus();
#ifdef SRGPT_TUNING_KNOBS
    if (!use_glds)
      hipLaunchKernelGGL((gemm_bf16_mfma<64, 128>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
    else
#endif
    if (dbuf)
      hipLaunchKernelGGL((gemm_bf16_glds<64, 128, 2>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
    else
      hipLaunchKernelGGL((gemm_bf16_glds<64, 128, 1>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
  }
  SRGPT_LAUNCH_CHECK();
  if (e.splits > 1) {
    {
          const bool fused = launch_splitk_reduce<bf16_t>(e, s);
          if (norm_done) *norm_done = fused;
        }
    SRGPT_LAUNCH_CHECK();
  }
  return SRGPT_OK;
}

// C = epilogue((A @ fp8(W8)^T) * wscale): bf16 activations, OCP e4m3fn weight bytes, one fp32 scale per weight row -- the
// prefill-side companion of srgpt_gemv_w8 (BASELINE config 5).  Always the 256 x 256 kernel (W tile staged as bytes, widened
// to bf16 between LDS and the MFMA operands); K splits (deterministic slabs) when the tiles do not fill the chip.
extern "C" int srgpt_gemm(const void* A, const void* W, const void* bias, const void* residual, void* C, int M,
                          int N, int K, int lda, int ldc, int act, int bias_mod, int res_mod, int out_f32,
                          int out_mode, int gw, void* ws, int64_t ws_bytes, int dtype, srgpt_stream_t stream) {
  return gemm_impl(A, W, bias, residual, C, M, N, K, lda, ldc, act, bias_mod, res_mod, out_f32, out_mode, gw, ws, ws_bytes, dtype,
                   stream, 0, nullptr, nullptr, nullptr, 0.f, nullptr);
}

// C = A W^T + bias + residual (dense rows) and Y = norm(C) in one call (include/srgpt.h; the layer loops of model.hip).  The norm
// rides in the split-K reduction when the product is split; otherwise it is the ordinary srgpt_rmsnorm / srgpt_layernorm launch --
// either way the two outputs are bit-identical to srgpt_gemm followed by the norm.
extern "C" int srgpt_gemm_norm(const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K,
                               void* ws, int64_t ws_bytes, int norm_kind, const void* norm_w, const void* norm_b, void* Y,
                               float norm_eps, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(norm_kind == SRGPT_NORM_RMS || norm_kind == SRGPT_NORM_LAYER, SRGPT_ERR_ARG, "srgpt_gemm_norm: unknown norm kind %d",
              norm_kind);
  SRGPT_CHECK(norm_w && Y && (norm_kind == SRGPT_NORM_RMS || norm_b), SRGPT_ERR_ARG, "srgpt_gemm_norm: null pointer");
  SRGPT_CHECK(Y != C, SRGPT_ERR_ARG, "srgpt_gemm_norm: Y must not alias C");  // Y may be A: the norm runs after the product
  bool fused = false;
  SRGPT_TRY(gemm_impl(A, W, bias, residual, C, M, N, K, K, N, SRGPT_ACT_NONE, 0, 0, 0, SRGPT_OUT_PLAIN, 0, ws, ws_bytes, dtype, stream,
                      norm_kind, norm_w, norm_b, Y, norm_eps, &fused));
  if (!fused) {
    if (norm_kind == SRGPT_NORM_RMS) SRGPT_TRY(srgpt_rmsnorm(C, norm_w, Y, M, N, norm_eps, dtype, stream));
    else SRGPT_TRY(srgpt_layernorm(C, norm_w, norm_b, Y, M, N, norm_eps, SRGPT_ACT_NONE, dtype, stream));
  }
  return SRGPT_OK;
}

// qkv = A W^T for the B * T token rows of a prefill, then RoPE on q (in place) and k, k and v appended to the caches -- the
// projection and srgpt_rope_kv_append in one call (include/srgpt.h).  The rotation rides in the split-K reduction when the product is
// split; otherwise it is the ordinary srgpt_rope_kv_append launch.  Every byte written is the same either way.
extern "C" int srgpt_gemm_rope_kv_append(const void* A, const void* W, void* qkv, int K, void* ws, int64_t ws_bytes, void* kcache,
                                         void* vcache, const int* pos0, const void* cos_tab, const void* sin_tab, int B, int T_,
                                         int Hq, int Hkv, int D, int max_pos, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(qkv && kcache && vcache && cos_tab && sin_tab, SRGPT_ERR_ARG, "srgpt_gemm_rope_kv_append: null pointer");
  SRGPT_CHECK(B > 0 && T_ > 0 && Hq > 0 && Hkv > 0 && D > 0 && (D & 1) == 0 && T_ <= max_pos, SRGPT_ERR_ARG,
              "srgpt_gemm_rope_kv_append: bad shape");
  const int M = B * T_, N = (Hq + 2 * Hkv) * D;
  Epilogue r{};
  r.rope_k = kcache, r.rope_v = vcache, r.rope_pos0 = pos0, r.rope_cos = cos_tab, r.rope_sin = sin_tab;
  r.rope_T = T_, r.rope_Hq = Hq, r.rope_Hkv = Hkv, r.rope_D = D, r.rope_max_pos = max_pos;
  bool fused = false;
  SRGPT_TRY(gemm_impl(A, W, nullptr, nullptr, qkv, M, N, K, K, N, SRGPT_ACT_NONE, 0, 0, 0, SRGPT_OUT_PLAIN, 0, ws, ws_bytes, dtype,
                      stream, 0, nullptr, nullptr, nullptr, 0.f, &fused, &r));
  if (!fused) SRGPT_TRY(srgpt_rope_kv_append(qkv, kcache, vcache, pos0, cos_tab, sin_tab, B, T_, Hq, Hkv, D, max_pos, dtype, stream));
  return SRGPT_OK;
}

// out[M, I] = rnd(rnd(silu(A Wg^T)) * (A Wu^T)) for the stacked weight Wgu = [Wg; Wu] ([2 I, K]) -- LlamaMLP's gate / up products and
// srgpt_silu_mul in one call (include/srgpt.h).  On the whole-M kernel (225 .. 272 rows: the bs = 1 prefill) a block multiplies 64
// gate columns and the 64 matching up columns and the activation is its epilogue: the [M, 2 I] intermediate is neither written nor
// read.  Every other shape runs srgpt_gemm into `gu_scratch` and srgpt_silu_mul -- the result is the same to the last bit.
extern "C" int srgpt_gemm_swiglu(const void* A, const void* Wgu, void* out, int M, int I, int K, void* gu_scratch, void* ws,
                                 int64_t ws_bytes, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(A && Wgu && out, SRGPT_ERR_ARG, "srgpt_gemm_swiglu: null pointer");
  SRGPT_CHECK(M > 0 && I > 0 && K > 0, SRGPT_ERR_ARG, "srgpt_gemm_swiglu: bad shape M=%d I=%d K=%d", M, I, K);
  const int mode = SRGPT_KNOB("SRGPT_GEMM_SWIGLU", 1);  // tuning build: 0 = always the two launches
  const bool fused = mode != 0 && dtype == SRGPT_BF16 && M > 224 && M <= 272 && K % 64 == 0 && K / 64 >= 4 && I % 64 == 0 &&
                     I / 64 >= srgpt_device_cus() / 2 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)Wgu % 16 == 0);
  if (fused) {
    Epilogue e{};
    e.C = out, e.M = M, e.N = I, e.ldc = I, e.act = SRGPT_ACT_NONE, e.splits = 1, e.swiglu_inter = I;
    SRGPT_TRY(srgpt_gemm288_launch(A, Wgu, K, K, e, as_stream(stream)));
    return SRGPT_OK;
  }
  SRGPT_CHECK(gu_scratch, SRGPT_ERR_ARG, "srgpt_gemm_swiglu: this shape needs the [M, 2 I] scratch buffer");
  SRGPT_TRY(srgpt_gemm(A, Wgu, nullptr, nullptr, gu_scratch, M, 2 * I, K, K, 2 * I, SRGPT_ACT_NONE, 0, 0, 0, SRGPT_OUT_PLAIN, 0, ws,
                       ws_bytes, dtype, stream));
  return srgpt_silu_mul(gu_scratch, out, M, I, dtype, stream);
}

extern "C" int srgpt_gemm_w8(const void* A, const void* W8, const float* wscale, const void* bias, const void* residual,
                             void* C, int M, int N, int K, int lda, int ldc, int act, int out_f32, void* ws,
                             int64_t ws_bytes, srgpt_stream_t stream) {
  SRGPT_CHECK(A && W8 && wscale && C, SRGPT_ERR_ARG, "srgpt_gemm_w8: null pointer");
  SRGPT_CHECK(M > 0 && N > 0 && K > 0, SRGPT_ERR_ARG, "srgpt_gemm_w8: bad shape M=%d N=%d K=%d", M, N, K);
  SRGPT_CHECK(lda >= K, SRGPT_ERR_ARG, "srgpt_gemm_w8: lda < K");
  SRGPT_CHECK(ldc >= N, SRGPT_ERR_ARG, "srgpt_gemm_w8: ldc < N");
  Epilogue e{bias, residual, C, M, N, ldc, act, 0, 0, out_f32, SRGPT_OUT_PLAIN, 0, nullptr, 1, 0, wscale};
  hipStream_t s = as_stream(stream);
  if (K % 64 != 0 || lda % 8 != 0 || ((uintptr_t)A % 16) || ((uintptr_t)W8 % 16)) {  // odd geometries: the scalar kernel
    hipLaunchKernelGGL(gemm_w8_simple, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(256), 0, s, (const bf16_t*)A,
                       (const unsigned char*)W8, K, lda, e);
    SRGPT_LAUNCH_CHECK();
    return SRGPT_OK;
  }
  const int cus = srgpt_device_cus();
  const int nk = K / 64;
  const long tiles = (long)cdiv(M, 256) * cdiv(N, 256);
  int sp = 1;
  if (tiles < cus && ws) {
    sp = (int)(cus / tiles);
    if (sp > 4) sp = 4;
    while (sp > 1 && (nk / sp < 8 || (int64_t)sp * M * N * 4 > ws_bytes)) --sp;
  }
  if (sp > 1) {
    e.partial = reinterpret_cast<float*>(ws);
    e.tiles_per_split = cdiv(nk, sp);
    e.splits = cdiv(nk, e.tiles_per_split);
  }
  SRGPT_TRY(srgpt_gemm256_launch(A, W8, K, lda, e, s));
  if (e.splits > 1) {
    launch_splitk_reduce<bf16_t>(e, s);
    SRGPT_LAUNCH_CHECK();
  }
  return SRGPT_OK;
}
