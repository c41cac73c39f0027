// Decode-path weight-streaming GEMV (batch <= 4 rows): out[b, n] = W[n, :] . x[b, :]   (HBM-bound)
//
// Roofline: every weight byte is read exactly once per token (non-temporal 16-byte loads, one 1 KiB
// wave-instruction per row chunk); x lives in LDS; fp32 accumulate on the VALU (24 lane-ops per 16 B --
// ~12 % of VALU issue at HBM rate, so no MFMA reshaping).
//
// Work decomposition: a unit = one output row (plain) or one (gate row, up row) pair (SwiGLU); each wave
// owns units grid-strided, each unit's K range is cut into batches of 8 row chunks (8 KiB per wave).
// The (unit, batch) sequence of a wave is flattened and software-pipelined THREE batches deep
// (cur / n1 / n2 registers): 16 chunk loads stay in flight per wave while 8 are consumed, the stream
// never drains at row boundaries, and the first two batches are issued BEFORE the activation / RMSNorm
// prologue, so short matrices (o_proj: 2 batches per wave) pay a single memory latency.
// Grid = 2 blocks per CU (8 waves/CU, ~128 KiB in flight per CU).
//
// Fusions (include/srgpt.h): RMSNorm prologue (LlamaRMSNorm), SwiGLU epilogue, residual add, fp32 logits.
// Rounding points mirror PyTorch's bf16 materialisation of each intermediate.
#include <stdlib.h>

#include "common.h"

// Optional per-wave timestamping (scripts/ubench_gemv_ts.hip builds this file with -DSRGPT_GEMV_TS)
#ifdef SRGPT_GEMV_TS
__device__ long long srgpt_gemv_ts[8 * 8192];
extern "C" void* srgpt_gemv_ts_ptr() {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(srgpt_gemv_ts));
  return p;
}
#define SRGPT_TS(slot)                                                                             \
  do {                                                                                             \
    if (lane == 0) srgpt_gemv_ts[((int)blockIdx.x * 4 + wave) * 8 + (slot)] = wall_clock64();     \
  } while (0)
#else
#define SRGPT_TS(slot)
#endif

namespace {

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

template <typename T, int B, bool SWIGLU, int NX>
__global__ __launch_bounds__(256, 2) void gemv_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                      const T* __restrict__ norm_w, float norm_eps,
                                                      const T* __restrict__ residual, void* __restrict__ out, int N,
                                                      int K, int out_f32) {
  constexpr int VEC = WChunk<T>::VEC;
  constexpr int R = SWIGLU ? 2 : 1;  // weight rows per unit
  constexpr int U = 8 / R;           // K-chunks per row per batch (8 loads per batch)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);  // [B][K]
  __shared__ float red[16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = K / VEC;          // 16-byte chunks per row
  const int nit = (nchunks + 63) >> 6;  // chunk iterations per row (64 lanes each)
  const int NB = (nit + U - 1) / U;     // batches per unit
  const int units = N;
  const int wstride = gridDim.x * 4;

  struct Cursor {
    int unit, b;
  };
  auto advance = [&](Cursor c) {
    Cursor n{c.unit, c.b + 1};
    if (n.b == NB) {
      n.b = 0;
      n.unit += wstride;
    }
    return n;
  };
  // loads are UNCONDITIONAL (indices clamped, surplus data discarded by the consumer): straight-line code lets
  // the compiler emit counted s_waitcnt vmcnt(N) instead of draining the queue at every use
  auto load = [&](Cursor c, u32x4 (&dst)[R][U]) {
    // dead cursors (past the wave's last unit) must issue NOTHING: redundant loads to a clamped address hot-spot
    // one L2 channel and were measured to add microseconds to every launch.  The branch is scalar (readfirstlane).
    const int u = __builtin_amdgcn_readfirstlane(c.unit);
    if (u < units) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)(u + r * N) * K);
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int ch = min((c.b * U + j) * 64 + lane, nchunks - 1);
          dst[r][j] = __builtin_nontemporal_load(p + ch);
        }
      }
    }
  };

  u32x4 cur[R][U], n1[R][U], n2[R][U];
  SRGPT_TS(0);
  Cursor c0{(int)blockIdx.x * 4 + wave, 0};
  Cursor c1 = advance(c0), c2 = advance(c1);
  // ---- prologue: stage x (and RMSNorm it) into LDS.
  // Program order matters: vmcnt retires in order, so the (L2-resident) activation and gain loads are issued
  // FIRST and the two weight batches right after them -- the prologue then waits only for its own small loads
  // while 16 KiB of weights per wave are already in flight.
  // NX = activation chunks per thread held in registers (host picks 2 or 8; covers B*K <= NX*2048 elements)
  const int total = B * nchunks;
  Vec16<T> xr[NX], gr[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int c = min(tid + 256 * j, total - 1);
    xr[j] = *reinterpret_cast<const Vec16<T>*>(x + (size_t)c * VEC);
  }
  if (norm_w) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int c = min(tid + 256 * j, total - 1);
      gr[j] = *reinterpret_cast<const Vec16<T>*>(norm_w + (size_t)(c % nchunks) * VEC);
    }
  }
  load(c0, n1);
  load(c1, n2);
  SRGPT_TS(1);
  {
    float ss[B];
#pragma unroll
    for (int b = 0; b < B; ++b) ss[b] = 0.f;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int c = tid + 256 * j;
      if (c < total) {
        if (norm_w) {
          const int b = c / nchunks;
          float sq = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) sq += xr[j].get(i) * xr[j].get(i);
#pragma unroll
          for (int bb = 0; bb < B; ++bb)
            if (bb == b) ss[bb] += sq;
        } else {
          *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = xr[j];
        }
      }
    }
    for (int c = tid + 256 * NX; c < total; c += 256) {  // rare: B*K > 16384
      Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + (size_t)c * VEC);
      *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
      if (norm_w) {
        const int b = c / nchunks;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) sq += v.get(i) * v.get(i);
#pragma unroll
        for (int bb = 0; bb < B; ++bb)
          if (bb == b) ss[bb] += sq;
      }
    }
    if (norm_w) {
      float rs[B];
#pragma unroll
      for (int b = 0; b < B; ++b) rs[b] = rsqrtf(block_sum(ss[b], red) / (float)K + norm_eps);
      auto pick = [&](int b) {
        float r = rs[0];
#pragma unroll
        for (int bb = 1; bb < B; ++bb)
          if (bb == b) r = rs[bb];
        return r;
      };
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int c = tid + 256 * j;
        if (c < total) {
          const float r = pick(c / nchunks);
          Vec16<T> v;
#pragma unroll
          for (int i = 0; i < VEC; ++i) v.set(i, gr[j].get(i) * rnd<T>(xr[j].get(i) * r));  // weight * h.to(dtype)
          *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
        }
      }
      for (int c = tid + 256 * NX; c < total; c += 256) {
        const int b = c / nchunks, kc = c - b * nchunks;
        Vec16<T> v = *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC);  // written by this same thread above
        const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(norm_w + (size_t)kc * VEC);
        const float r = pick(b);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v.set(i, g.get(i) * rnd<T>(v.get(i) * r));
        *reinterpret_cast<Vec16<T>*>(xs + (size_t)c * VEC) = v;
      }
    }
    __syncthreads();
  }

  float acc[R][B];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
  SRGPT_TS(2);
#ifdef SRGPT_GEMV_TS
  int ts_it = 0;
#endif

  while (c0.unit < units) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < U; ++j) {
        cur[r][j] = n1[r][j];
        n1[r][j] = n2[r][j];
      }
    const Cursor c3 = advance(c2);
    load(c2, n2);  // third batch ahead of the one being consumed

#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int ch = (c0.b * U + j) * 64 + lane;
      if (ch < nchunks) {
        float wf[R][VEC];
#pragma unroll
        for (int r = 0; r < R; ++r) WChunk<T>::cvt(cur[r][j], wf[r]);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(xs + (size_t)b * K + (size_t)ch * VEC);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float xf = xv.get(i);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][b] = fmaf(wf[r][i], xf, acc[r][b]);
          }
        }
      }
    }

    if (c0.b == NB - 1) {
      const int n = c0.unit;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        float a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          a[r] = wave_sum(acc[r][b]);
          acc[r][b] = 0.f;
        }
        if (lane == 0) {
          if (SWIGLU) {
            const float g = rnd<T>(a[0]), u = rnd<T>(a[R - 1]);
            reinterpret_cast<T*>(out)[(size_t)b * N + n] = from_f<T>(rnd<T>(silu(g)) * u);
          } else {
            float v = rnd<T>(a[0]);
            if (residual) v = rnd<T>(to_f(residual[(size_t)b * N + n]) + v);
            if (out_f32)
              reinterpret_cast<float*>(out)[(size_t)b * N + n] = v;
            else
              reinterpret_cast<T*>(out)[(size_t)b * N + n] = from_f<T>(v);
          }
        }
      }
    }
#ifdef SRGPT_GEMV_TS
    if (ts_it == 0) SRGPT_TS(3);
    ++ts_it;
#endif
    c0 = c1;
    c1 = c2;
    c2 = c3;
  }
  SRGPT_TS(4);
}

template <typename T, int B>
int launch_gemv(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out, int N,
                int K, int swiglu, int out_f32, hipStream_t s) {
  const size_t lds = (size_t)B * K * sizeof(T);
  SRGPT_CHECK(lds <= 150 * 1024, SRGPT_ERR_UNSUPPORTED, "srgpt_gemv: batch*K too large for LDS (%zu bytes)", lds);
  const int cus = srgpt_device_cus();
  static const int env_per_cu = getenv("SRGPT_GEMV_BLOCKS_PER_CU") ? atoi(getenv("SRGPT_GEMV_BLOCKS_PER_CU")) : 0;  // tuning knob
  const int per_cu = lds > 70 * 1024 ? 1 : (env_per_cu > 0 ? env_per_cu : 2);
  int grid = (N + 3) / 4;
  if (grid > cus * per_cu) grid = cus * per_cu;
  if (grid < 1) grid = 1;
  const int chunks = B * (K / WChunk<T>::VEC);
#define SRGPT_GEMV_LAUNCH(SW, NXV)                                                                              \
  do {                                                                                                          \
    auto kfn = gemv_kernel<T, B, SW, NXV>;                                                                      \
    static bool attr_set = false;                                                                               \
    if (lds > 48 * 1024 && !attr_set) {                                                                         \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);      \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, (const T*)x, (const T*)W, (const T*)norm_w, eps,     \
                       (const T*)residual, out, N, K, out_f32);                                                 \
  } while (0)
  if (swiglu) {
    if (chunks <= 512) SRGPT_GEMV_LAUNCH(true, 2); else SRGPT_GEMV_LAUNCH(true, 8);
  } else {
    if (chunks <= 512) SRGPT_GEMV_LAUNCH(false, 2); else SRGPT_GEMV_LAUNCH(false, 8);
  }
#undef SRGPT_GEMV_LAUNCH
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
