// GEMM family: C[M,N] = act(A[M,K] @ W[N,K]^T + bias) + residual      (see include/srgpt.h)
//
// bf16: LDS-tiled MFMA kernel (v_mfma_f32_32x32x16_bf16), 256 threads = 2x2 waves, BK = 64,
//       register-staged double buffering, XOR-swizzled 16-byte LDS slots (conflict <= 2-way on
//       ds_read_b128), fused epilogue (bias / activation / residual / deconv pixel-shuffle / fp32 out).
//       Both operands are K-contiguous (nn.Linear weight layout), so A and W tiles stage identically.
// fp32: plain LDS-tiled FMA kernel; exists for tight-tolerance parity runs of the same host path.
#include "common.h"

namespace {

struct Epilogue {
  const void* bias;
  const void* residual;
  void* C;
  int M, N, ldc, act, bias_mod, res_mod, out_f32, out_mode, gw;
};

template <typename T>
__device__ __forceinline__ void epilogue_store(const Epilogue& e, int m, int n, float acc) {
  if (m >= e.M || n >= e.N) return;
  float v = acc;
  if (e.bias) {
    const int bi = e.bias_mod > 0 ? n % e.bias_mod : n;
    v += to_f(reinterpret_cast<const T*>(e.bias)[bi]);
  }
  v = rnd<T>(v);  // nn.Linear / conv output is materialised in T
  if (e.act != SRGPT_ACT_NONE) v = rnd<T>(apply_act<T>(v, e.act));
  if (e.residual) {
    const int rm = e.res_mod > 0 ? m % e.res_mod : m;
    v = rnd<T>(v + to_f(reinterpret_cast<const T*>(e.residual)[(size_t)rm * e.N + n]));
  }
  size_t off;
  if (e.out_mode == SRGPT_OUT_DECONV2X) {
    const int cout = e.N >> 2, gg = e.gw * e.gw;
    const int img = m / gg, rem = m - img * gg, i = rem / e.gw, j = rem - i * e.gw;
    const int tap = n / cout, co = n - tap * cout, a = tap >> 1, b = tap & 1;
    const int ow = 2 * e.gw;
    off = ((size_t)img * ow * ow + (size_t)(2 * i + a) * ow + (2 * j + b)) * cout + co;
  } else {
    off = (size_t)m * e.ldc + n;
  }
  if (e.out_f32)
    reinterpret_cast<float*>(e.C)[off] = v;
  else
    reinterpret_cast<T*>(e.C)[off] = from_f<T>(v);
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
constexpr int BK = 64;  // bf16 elements per K tile = 8 slots of 16 B

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

  u32x4 ra[LA], rw[LW];
  auto gload = [&](int kt) {
    const int k = kt * BK + sc * 8;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = m0 + sr + 32 * i;
      ra[i] = (m < e.M && k < K) ? *reinterpret_cast<const u32x4*>(A + (size_t)m * lda + k) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int n = n0 + sr + 32 * i;
      rw[i] = (n < e.N && k < K) ? *reinterpret_cast<const u32x4*>(W + (size_t)n * K + k) : u32x4{0, 0, 0, 0};
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = sr + 32 * i;
      *reinterpret_cast<u32x4*>(As + (size_t)buf * BM * BK + r * BK + ((sc ^ (r & 7)) << 3)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int r = sr + 32 * i;
      *reinterpret_cast<u32x4*>(Ws + (size_t)buf * BN * BK + r * BK + ((sc ^ (r & 7)) << 3)) = rw[i];
    }
  };

  f32x16 acc[TM][TN];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero16;

  const int nk = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    const bf16_t* as = As + (size_t)cur * BM * BK;
    const bf16_t* ws = Ws + (size_t)cur * BN * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fw[TN];
      const int slot = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / 2) + i * 32 + (lane & 31);
        fa[i] = *reinterpret_cast<const bf16x8*>(as + r * BK + ((slot ^ (r & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / 2) + j * 32 + (lane & 31);
        fw[j] = *reinterpret_cast<const bf16x8*>(ws + r * BK + ((slot ^ (r & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fw[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma clang loop unroll(full)
  for (int i = 0; i < TM; ++i)
#pragma clang loop unroll(full)
    for (int j = 0; j < TN; ++j) {
      const f32x16 a = acc[i][j];
      const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
#pragma clang loop unroll(full)
      for (int r = 0; r < 16; ++r) epilogue_store<bf16_t>(e, mb + (r & 3) + 8 * (r >> 2), n, a[r]);
    }
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

}  // namespace

extern "C" int srgpt_gemm(const void* A, const void* W, const void* bias, const void* residual, void* C, int M,
                          int N, int K, int lda, int ldc, int act, int bias_mod, int res_mod, int out_f32,
                          int out_mode, int gw, int dtype, srgpt_stream_t stream) {
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
  Epilogue e{bias, residual, C, M, N, ldc, act, bias_mod, res_mod, out_f32, out_mode, gw};
  hipStream_t s = as_stream(stream);
  if (dtype == SRGPT_F32) {
    dim3 grid(cdiv(N, 64), cdiv(M, 64));
    hipLaunchKernelGGL(gemm_f32_simple, grid, dim3(256), 0, s, (const float*)A, (const float*)W, K, lda, e);
  } else {
    // pick the tile so that the grid covers the 256 CUs when the problem allows it
    const long b128 = (long)cdiv(M, 128) * cdiv(N, 128);
    if (b128 >= 256 || (M > 64 && (long)cdiv(M, 64) * cdiv(N, 128) < 64)) {
      dim3 grid(cdiv(N, 128), cdiv(M, 128));
      hipLaunchKernelGGL((gemm_bf16_mfma<128, 128>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
    } else if ((long)cdiv(M, 64) * cdiv(N, 128) >= 192 || N >= 4 * M) {
      dim3 grid(cdiv(N, 128), cdiv(M, 64));
      hipLaunchKernelGGL((gemm_bf16_mfma<64, 128>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
    } else {
      dim3 grid(cdiv(N, 64), cdiv(M, 64));
      hipLaunchKernelGGL((gemm_bf16_mfma<64, 64>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)W, K, lda, e);
    }
  }
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
