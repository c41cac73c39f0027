// Token-stream and elementwise kernels: embedding gather, row scatter (splice), RoPE + KV append
// (prefill), SiLU*up, argmax, decode bookkeeping.  All HBM/latency-bound; 16-byte vector access.
#include <stdarg.h>

#include "common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void srgpt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* srgpt_last_error(void) { return g_err; }
extern "C" int srgpt_abi_version(void) { return 9; }
extern "C" int srgpt_device_cus(void) {
  // per device ordinal, filled once each (benign race: every writer stores the same value)
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& c = cus[dev & 63];
  int v = c.load(std::memory_order_relaxed);
  if (v == 0) {
    int n = 0;
    v = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;  // MI355X
    c.store(v, std::memory_order_relaxed);
  }
  return v;
}

namespace {

template <typename T>
__global__ void embed_rows_kernel(const T* __restrict__ table, const int64_t* __restrict__ ids, T* __restrict__ out,
                                  int cols) {
  constexpr int VEC = Vec16<T>::N;
  const int64_t id = ids[blockIdx.x];
  const T* src = table + (size_t)id * cols;
  T* dst = out + (size_t)blockIdx.x * cols;
  for (int c = threadIdx.x; c < cols / VEC; c += blockDim.x)
    *reinterpret_cast<u32x4*>(dst + c * VEC) = *reinterpret_cast<const u32x4*>(src + c * VEC);
}

template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, const int* __restrict__ src_idx,
                                    const int* __restrict__ idx, T* __restrict__ dst, int cols) {
  constexpr int VEC = Vec16<T>::N;
  const int r = idx[blockIdx.x];
  if (r < 0) return;
  const T* s = src + (size_t)(src_idx ? src_idx[blockIdx.x] : blockIdx.x) * cols;
  T* d = dst + (size_t)r * cols;
  for (int c = threadIdx.x; c < cols / VEC; c += blockDim.x)
    *reinterpret_cast<u32x4*>(d + c * VEC) = *reinterpret_cast<const u32x4*>(s + c * VEC);
}

template <typename T>
__global__ void silu_mul_kernel(const T* __restrict__ gu, T* __restrict__ out, int inter, size_t total_chunks) {
  constexpr int VEC = Vec16<T>::N;
  const int cpr = inter / VEC;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < total_chunks; c += (size_t)gridDim.x * blockDim.x) {
    const size_t row = c / cpr;
    const int cc = (int)(c - row * cpr);
    const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(gu + row * 2 * inter + (size_t)cc * VEC);
    const Vec16<T> u = *reinterpret_cast<const Vec16<T>*>(gu + row * 2 * inter + inter + (size_t)cc * VEC);
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, rnd<T>(silu(g.get(i))) * u.get(i));
    *reinterpret_cast<Vec16<T>*>(out + row * inter + (size_t)cc * VEC) = o;
  }
}

// first-max-wins argmax over fp32 logits, one block per batch row
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int64_t* __restrict__ ids_out,
                                                      int V) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const float* row = logits + (size_t)blockIdx.x * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const float v = row[i];
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sv[w] = best;
    si[w] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i)
      if (sv[i] > best || (sv[i] == best && si[i] < bi)) {
        best = sv[i];
        bi = si[i];
      }
    ids_out[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
  }
}

// RoPE (rotate_half form, modeling_llama.py:160-191) + cache append for a block of T tokens.
// one block per (token, batch); thread per (head, pair i < D/2)
template <typename T>
__global__ void rope_kv_append_kernel(T* __restrict__ qkv, T* __restrict__ kcache, T* __restrict__ vcache,
                                      const int* __restrict__ pos0, const T* __restrict__ cos_tab,
                                      const T* __restrict__ sin_tab, int Tn, int Hq, int Hkv, int D, int max_pos) {
  const int t = blockIdx.x, b = blockIdx.y;
  const int half = D >> 1;
  const int pos = (pos0 ? pos0[b] : 0) + t;
  T* row = qkv + ((size_t)b * Tn + t) * (size_t)(Hq + 2 * Hkv) * D;
  const int nrot = (Hq + Hkv) * half;
  for (int w = threadIdx.x; w < nrot; w += blockDim.x) {
    const int h = w / half, i = w - h * half;
    const float c = to_f(cos_tab[(size_t)pos * half + i]), s = to_f(sin_tab[(size_t)pos * half + i]);
    T* p = row + (size_t)h * D;
    const float x1 = to_f(p[i]), x2 = to_f(p[i + half]);
    // q*cos + rotate_half(q)*sin with every intermediate materialised in T
    const float o1 = rnd<T>(rnd<T>(x1 * c) + rnd<T>(-x2 * s));
    const float o2 = rnd<T>(rnd<T>(x2 * c) + rnd<T>(x1 * s));
    if (h < Hq) {
      p[i] = from_f<T>(o1);
      p[i + half] = from_f<T>(o2);
    } else {
      T* kc = kcache + (((size_t)b * Hkv + (h - Hq)) * max_pos + pos) * D;
      kc[i] = from_f<T>(o1);
      kc[i + half] = from_f<T>(o2);
    }
  }
  const T* vsrc = row + (size_t)(Hq + Hkv) * D;
  for (int w = threadIdx.x; w < Hkv * D; w += blockDim.x) {
    const int h = w / D, i = w - h * D;
    vcache[(((size_t)b * Hkv + h) * max_pos + pos) * D + i] = vsrc[w];
  }
}


// Beam search: the live KV-cache rows follow the beams the search kept (HF `_reorder_cache`, which the reference inherits through
// GenerationMixin.beam_search / beam_sample at llava_llama.py:212).  In place, ONE launch for K and V: a thread owns 16 bytes of one
// cached position of one (layer, kv head) and moves them for ALL beams of a batch item -- every source row is in its registers
// before any destination row is written, and beam indices never leave their batch item, so no block depends on another.
template <int NB>
__global__ __launch_bounds__(256) void kv_beam_reorder_kernel(unsigned char* __restrict__ kc, unsigned char* __restrict__ vc,
                                                              const int64_t* __restrict__ beam_idx, int rows, int Hkv, int max_pos,
                                                              int row16 /* 16-byte pieces per cached position */, int live) {
  const int item = blockIdx.y, hl = blockIdx.z;  // batch item; (layer * Hkv + head)
  const int layer = hl / Hkv, head = hl - layer * Hkv;
  const long long piece = (long long)blockIdx.x * 256 + threadIdx.x;  // over live * row16
  if (piece >= (long long)live * row16) return;
  int src[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) src[j] = (int)beam_idx[item * NB + j];
  const size_t row_stride = (size_t)Hkv * max_pos * row16, layer_stride = (size_t)rows * row_stride;
  const size_t base = (size_t)layer * layer_stride + (size_t)head * max_pos * row16 + (size_t)piece;
  u32x4* k16 = reinterpret_cast<u32x4*>(kc);
  u32x4* v16 = reinterpret_cast<u32x4*>(vc);
  u32x4 kk[NB], vv[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    kk[j] = k16[base + (size_t)src[j] * row_stride];
    vv[j] = v16[base + (size_t)src[j] * row_stride];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (src[j] != item * NB + j) {  // (a beam that continues its own row has nothing to move)
      k16[base + (size_t)(item * NB + j) * row_stride] = kk[j];
      v16[base + (size_t)(item * NB + j) * row_stride] = vv[j];
    }
  }
}

}  // namespace

#define DISPATCH_T(dtype, NAME, ...)                   \
  if ((dtype) == SRGPT_BF16) {                         \
    using T = bf16_t;                                  \
    NAME(__VA_ARGS__);                                 \
  } else if ((dtype) == SRGPT_F32) {                   \
    using T = float;                                   \
    NAME(__VA_ARGS__);                                 \
  } else {                                             \
    srgpt_set_error("bad dtype %d", (int)(dtype));     \
    return SRGPT_ERR_ARG;                              \
  }

extern "C" int srgpt_embed_rows(const void* table, const int64_t* ids, void* out, int n, int cols, int dtype,
                                srgpt_stream_t stream) {
  SRGPT_CHECK(table && ids && out && n > 0 && cols > 0, SRGPT_ERR_ARG, "srgpt_embed_rows: bad args");
  SRGPT_CHECK(cols % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_embed_rows: cols not 16-byte multiple");
#define L() hipLaunchKernelGGL(embed_rows_kernel<T>, dim3(n), dim3(256), 0, as_stream(stream), (const T*)table, ids, (T*)out, cols)
  DISPATCH_T(dtype, L);
#undef L
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_kv_beam_reorder(void* kcache, void* vcache, const int64_t* beam_idx, int layers, int batch, int num_beams,
                                     int kv_heads, int max_pos, int head_dim, int live, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(kcache && vcache && beam_idx, SRGPT_ERR_ARG, "srgpt_kv_beam_reorder: null pointer");
  SRGPT_CHECK(layers > 0 && batch > 0 && kv_heads > 0 && max_pos > 0 && head_dim > 0 && live >= 0 && live <= max_pos, SRGPT_ERR_ARG,
              "srgpt_kv_beam_reorder: bad shape");
  const int eb = dtype == SRGPT_BF16 ? 2 : dtype == SRGPT_F32 ? 4 : 0;
  SRGPT_CHECK(eb != 0 && (head_dim * eb) % 16 == 0, SRGPT_ERR_ARG, "srgpt_kv_beam_reorder: rows must be whole 16-byte pieces");
  SRGPT_CHECK(num_beams >= 2 && num_beams <= 8, SRGPT_ERR_UNSUPPORTED, "srgpt_kv_beam_reorder: %d beams (2 ... 8)", num_beams);
  if (live == 0) return SRGPT_OK;
  const int row16 = head_dim * eb / 16;
  const dim3 grid((unsigned)(((long long)live * row16 + 255) / 256), (unsigned)batch, (unsigned)(layers * kv_heads));
#define L(NB)                                                                                                                    \
  hipLaunchKernelGGL(kv_beam_reorder_kernel<NB>, grid, dim3(256), 0, as_stream(stream), (unsigned char*)kcache, (unsigned char*)vcache, \
                     beam_idx, batch * num_beams, kv_heads, max_pos, row16, live)
  switch (num_beams) {
    case 2: L(2); break;
    case 3: L(3); break;
    case 4: L(4); break;
    case 5: L(5); break;
    case 6: L(6); break;
    case 7: L(7); break;
    default: L(8); break;
  }
#undef L
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_scatter_rows(const void* src, const int* src_idx, const int* idx, void* dst, int n, int cols,
                                  int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(src && idx && dst && n > 0 && cols > 0, SRGPT_ERR_ARG, "srgpt_scatter_rows: bad args");
  SRGPT_CHECK(cols % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_scatter_rows: cols not 16-byte multiple");
#define L() hipLaunchKernelGGL(scatter_rows_kernel<T>, dim3(n), dim3(256), 0, as_stream(stream), (const T*)src, src_idx, idx, (T*)dst, cols)
  DISPATCH_T(dtype, L);
#undef L
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_silu_mul(const void* gu, void* out, int rows, int inter, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(gu && out && rows > 0 && inter > 0, SRGPT_ERR_ARG, "srgpt_silu_mul: bad args");
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(inter % vec == 0, SRGPT_ERR_ARG, "srgpt_silu_mul: inter not 16-byte multiple");
  const size_t chunks = (size_t)rows * (inter / vec);
  int grid = (int)((chunks + 255) / 256);
  if (grid > 4096) grid = 4096;
#define L() hipLaunchKernelGGL(silu_mul_kernel<T>, dim3(grid), dim3(256), 0, as_stream(stream), (const T*)gu, (T*)out, inter, chunks)
  DISPATCH_T(dtype, L);
#undef L
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

// ------------------------------------------------------------------------------------------------
// causal-LM loss (LlamaForCausalLM.forward with labels, modeling_llama.py:1047-1058): per target row
// logsumexp(logits) - logits[label], then the mean over the rows whose label is not ignore_index -- two deterministic
// kernels (row losses, fixed-order sum), no float atomics.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     float* __restrict__ row_loss, int V, int64_t ignore_index) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const int64_t y = labels[r];
  if (y == ignore_index || y < 0 || y >= V) {  // uniform per block
    if (threadIdx.x == 0) row_loss[r] = 0.f;
    return;
  }
  const float* p = logits + (size_t)r * V;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < V; i += 256) mx = fmaxf(mx, p[i]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) sum += expf(p[i] - mx);
  sum = block_sum(sum, red + 8);
  if (threadIdx.x == 0) row_loss[r] = logf(sum) + mx - p[y];
}
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                     float* __restrict__ out, int rows, int V, int64_t ignore_index) {
  __shared__ float red[16];
  float s = 0.f, n = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) {
    const int64_t y = labels[i];
    if (y != ignore_index && y >= 0 && y < V) {
      s += row_loss[i];
      n += 1.f;
    }
  }
  s = block_sum(s, red);
  n = block_sum(n, red + 8);
  if (threadIdx.x == 0) {
    out[0] = n > 0.f ? s / n : NAN;  // torch's mean over zero targets is nan
    out[1] = n;
  }
}
}  // namespace

extern "C" int srgpt_cross_entropy(const float* logits, const int64_t* labels, float* row_loss, float* out, int rows, int V,
                                   int64_t ignore_index, srgpt_stream_t stream) {
  SRGPT_CHECK(logits && labels && row_loss && out, SRGPT_ERR_ARG, "srgpt_cross_entropy: null pointer");
  SRGPT_CHECK(rows > 0 && V > 0, SRGPT_ERR_ARG, "srgpt_cross_entropy: bad shape");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(ce_rows_kernel, dim3(rows), dim3(256), 0, s, logits, labels, row_loss, V, ignore_index);
  SRGPT_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, s, row_loss, labels, out, rows, V, ignore_index);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_argmax(const float* logits, int64_t* ids_out, int B, int V, srgpt_stream_t stream) {
  SRGPT_CHECK(logits && ids_out && B > 0 && V > 0, SRGPT_ERR_ARG, "srgpt_argmax: bad args");
  hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(1024), 0, as_stream(stream), logits, ids_out, V);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_rope_kv_append(void* qkv, void* kcache, void* vcache, const int* pos0, const void* cos_tab,
                                    const void* sin_tab, int B, int T_, int Hq, int Hkv, int D, int max_pos,
                                    int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(qkv && kcache && vcache && cos_tab && sin_tab, SRGPT_ERR_ARG, "srgpt_rope_kv_append: null pointer");
  SRGPT_CHECK(B > 0 && T_ > 0 && Hq > 0 && Hkv > 0 && D > 0 && (D & 1) == 0 && T_ <= max_pos, SRGPT_ERR_ARG,
              "srgpt_rope_kv_append: bad shape");
#define L() hipLaunchKernelGGL(rope_kv_append_kernel<T>, dim3(T_, B), dim3(256), 0, as_stream(stream), (T*)qkv, (T*)kcache, (T*)vcache, pos0, (const T*)cos_tab, (const T*)sin_tab, T_, Hq, Hkv, D, max_pos)
  DISPATCH_T(dtype, L);
#undef L
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
