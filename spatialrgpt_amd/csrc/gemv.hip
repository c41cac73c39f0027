// Decode-path weight-streaming GEMV (batch <= 4 rows): out[b, n] = W[n, :] . x[b, :]   (HBM-bound)
//
// Roofline: every weight byte is read exactly once per token (non-temporal 16-byte loads, one
// 1 KiB wave-instruction per row chunk); x lives in LDS; fp32 accumulate on the VALU (24 lane-ops
// per 16 B -- 12 % of VALU issue at HBM rate, so no MFMA reshaping).  Each wave owns pairs of weight
// rows and keeps 8 (in flight) + 8 (being consumed) row chunks in registers, flattened across row
// pairs so the load stream never drains; the first batch is issued BEFORE the x / RMSNorm prologue so
// HBM is busy while every block stages x.  Grid = 2 blocks per CU, grid-strided over row pairs.
//
// Fusions (see include/srgpt.h): RMSNorm prologue (LlamaRMSNorm), SwiGLU epilogue, residual add,
// fp32 logits.  Rounding points mirror PyTorch's bf16 materialisation of each intermediate.
#include "common.h"

namespace {

constexpr int U = 4;  // K-chunks per row per batch (loads in flight = 2 rows * U)

template <typename T>
struct WChunk;  // 16 bytes of weights -> VEC floats
template <>
struct WChunk<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void cvt(const u32x4& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = bf16lo(r[i]);
      f[2 * i + 1] = bf16hi(r[i]);
    }
  }
};
template <>
struct WChunk<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void cvt(const u32x4& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(r[i]);
  }
};

template <typename T, int B, bool SWIGLU>
__global__ __launch_bounds__(256, 2) void gemv_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                      const T* __restrict__ norm_w, float norm_eps,
                                                      const T* __restrict__ residual, void* __restrict__ out, int N,
                                                      int K, int out_f32) {
  constexpr int VEC = WChunk<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);  // [B][K]
  __shared__ float red[16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = K / VEC;                      // 16-byte chunks per row
  const int nit = (nchunks + 63) >> 6;              // chunk iterations per row (64 lanes each)
  const int NB = (nit + U - 1) / U;                 // load batches per row pair
  const int units = SWIGLU ? N : (N + 1) >> 1;      // work units: (gate_n, up_n) or (row 2u, row 2u+1)
  const int wstride = gridDim.x * 4;
  int unit = blockIdx.x * 4 + wave;

  auto rowA = [&](int u) { return SWIGLU ? u : 2 * u; };
  auto rowB = [&](int u) { return SWIGLU ? N + u : min(2 * u + 1, N - 1); };

  u32x4 nxt[2][U], cur[2][U];
  auto load = [&](int u, int b) {
    const u32x4* pa = reinterpret_cast<const u32x4*>(W + (size_t)rowA(u) * K);
    const u32x4* pb = reinterpret_cast<const u32x4*>(W + (size_t)rowB(u) * K);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int c = (b * U + j) * 64 + lane;
      if (c < nchunks) {
        nxt[0][j] = __builtin_nontemporal_load(pa + c);
        nxt[1][j] = __builtin_nontemporal_load(pb + c);
      } else {
        nxt[0][j] = u32x4{0, 0, 0, 0};
        nxt[1][j] = u32x4{0, 0, 0, 0};
      }
    }
  };

  // first weight batch goes out before the activation prologue
  if (unit < units) load(unit, 0);

  // ---- prologue: stage x (and RMSNorm it) into LDS
  {
    float ss[B];
#pragma unroll
    for (int b = 0; b < B; ++b) ss[b] = 0.f;
    for (int c = tid; c < B * nchunks; c += 256) {
      Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + (size_t)c * VEC);
      *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
      if (norm_w) {
        const int b = c / nchunks;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += v.get(i) * v.get(i);
#pragma unroll
        for (int bb = 0; bb < B; ++bb)
          if (bb == b) ss[bb] += s;
      }
    }
    if (norm_w) {
      float rs[B];
#pragma unroll
      for (int b = 0; b < B; ++b) rs[b] = rsqrtf(block_sum(ss[b], red) / (float)K + norm_eps);
      __syncthreads();
      for (int c = tid; c < B * nchunks; c += 256) {
        const int b = c / nchunks, kc = c - b * nchunks;
        Vec16<T> v = *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC);
        Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(norm_w + (size_t)kc * VEC);
        float r = rs[0];
#pragma unroll
        for (int bb = 1; bb < B; ++bb)
          if (bb == b) r = rs[bb];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v.set(i, g.get(i) * rnd<T>(v.get(i) * r));  // weight * h.to(dtype)
        *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
      }
    }
    __syncthreads();
  }

  float acc[2][B];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int b = 0; b < B; ++b) acc[r][b] = 0.f;

  int bidx = 0;
  while (unit < units) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      cur[0][j] = nxt[0][j];
      cur[1][j] = nxt[1][j];
    }
    int nunit = unit, nb = bidx + 1;
    if (nb == NB) {
      nb = 0;
      nunit += wstride;
    }
    if (nunit < units) load(nunit, nb);

#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int c = (bidx * U + j) * 64 + lane;
      if (c < nchunks) {
        float wa[VEC], wb[VEC];
        WChunk<T>::cvt(cur[0][j], wa);
        WChunk<T>::cvt(cur[1][j], wb);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(xs + (size_t)b * K + (size_t)c * VEC);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float xf = xv.get(i);
            acc[0][b] = fmaf(wa[i], xf, acc[0][b]);
            acc[1][b] = fmaf(wb[i], xf, acc[1][b]);
          }
        }
      }
    }

    if (bidx == NB - 1) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float a0 = wave_sum(acc[0][b]);
        const float a1 = wave_sum(acc[1][b]);
        acc[0][b] = 0.f;
        acc[1][b] = 0.f;
        if (lane == 0) {
          if (SWIGLU) {
            const float g = rnd<T>(a0), u = rnd<T>(a1);
            const float v = rnd<T>(rnd<T>(silu(g)) * u);
            reinterpret_cast<T*>(out)[(size_t)b * N + unit] = from_f<T>(v);
          } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const int n = 2 * unit + r;
              if (n < N) {
                float v = rnd<T>(r == 0 ? a0 : a1);
                if (residual) v = rnd<T>(to_f(residual[(size_t)b * N + n]) + v);
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
    unit = nunit;
    bidx = nb;
  }
}

template <typename T, int B>
int launch_gemv(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out, int N,
                int K, int swiglu, int out_f32, hipStream_t s) {
  const size_t lds = (size_t)B * K * sizeof(T);
  SRGPT_CHECK(lds <= 150 * 1024, SRGPT_ERR_UNSUPPORTED, "srgpt_gemv: batch*K too large for LDS (%zu bytes)", lds);
  const int units = swiglu ? N : (N + 1) / 2;
  const int cus = srgpt_device_cus();
  const int per_cu = lds > 70 * 1024 ? 1 : 2;
  int grid = (units + 3) / 4;
  if (grid > cus * per_cu) grid = cus * per_cu;
  if (grid < 1) grid = 1;
  if (swiglu) {
    auto kfn = gemv_kernel<T, B, true>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, (const T*)x, (const T*)W, (const T*)norm_w, eps,
                       (const T*)residual, out, N, K, out_f32);
  } else {
    auto kfn = gemv_kernel<T, B, false>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, (const T*)x, (const T*)W, (const T*)norm_w, eps,
                       (const T*)residual, out, N, K, out_f32);
  }
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

template <typename T>
int dispatch_b(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out,
               int batch, int N, int K, int swiglu, int out_f32, hipStream_t s) {
  switch (batch) {
    case 1: return launch_gemv<T, 1>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    case 2: return launch_gemv<T, 2>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    case 3: return launch_gemv<T, 3>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    case 4: return launch_gemv<T, 4>(x, W, norm_w, eps, residual, out, N, K, swiglu, out_f32, s);
    default:
      srgpt_set_error("srgpt_gemv: batch %d not supported (1..4)", batch);
      return SRGPT_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int srgpt_gemv(const void* x, const void* W, const void* norm_w, float norm_eps, const void* residual,
                          void* out, int batch, int N, int K, int swiglu, int out_f32, int dtype,
                          srgpt_stream_t stream) {
  SRGPT_CHECK(x && W && out, SRGPT_ERR_ARG, "srgpt_gemv: null pointer");
  SRGPT_CHECK(N > 0 && K > 0 && batch > 0, SRGPT_ERR_ARG, "srgpt_gemv: bad shape");
  SRGPT_CHECK(dtype == SRGPT_F32 || dtype == SRGPT_BF16, SRGPT_ERR_ARG, "srgpt_gemv: bad dtype %d", dtype);
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(K % vec == 0, SRGPT_ERR_ARG, "srgpt_gemv: K=%d must be a multiple of %d", K, vec);
  SRGPT_CHECK(!(swiglu && (residual || out_f32)), SRGPT_ERR_ARG, "srgpt_gemv: swiglu excludes residual/out_f32");
  hipStream_t s = as_stream(stream);
  if (dtype == SRGPT_BF16)
    return dispatch_b<bf16_t>(x, W, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, s);
  return dispatch_b<float>(x, W, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, s);
}
