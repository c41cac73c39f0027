// Host-side composites over the per-op C ABI: vision tower forward, Llama prefill, device-side greedy
// decode step and its hipGraph.  Pure launch sequencing in C++ so one request costs a handful of
// FFI crossings instead of ~10^4 (SURVEY 3.1 hot loops); all state lives in caller-owned buffers.
#include <vector>

#include "common.h"

int srgpt_decode_attention_pf(const void* qkv, void* kcache, void* vcache, const int* pos, const void* cos_tab,
                              const void* sin_tab, void* out, float* ws, int B, int Hq, int Hkv, int D, int max_pos, int dtype,
                              const void* next_w, int next_n, int next_k, int next_fp8, int next_packed_rows, srgpt_stream_t stream);  // attn.hip
void* srgpt_decode_attn_sync_words(float* ws, int B, int Hq, int D, size_t* bytes);  // attn.hip
int srgpt_sample_launch(const float* logits, const srgpt_sampling* sp, int64_t* tok, void* ws, float* pv, int* pi, int* err, int B, int V,
                        hipStream_t s);  // sample.hip
extern "C" __attribute__((visibility("hidden"))) int srgpt_sample_slices(void);  // sample.hip (internal: not part of the C ABI)
namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  void* take(size_t bytes) {
    void* r = base ? base + off : nullptr;
    off += align256(bytes);
    return r;
  }
};

// ---- ViT workspace ----
struct VitWs {
  void *col, *h, *qkv, *mlp, *gws;
  size_t gws_bytes, total;
};
VitWs carve_vit(const srgpt_vit_weights* w, int n_img, void* ws) {
  const size_t es = dtype_size(w->dtype);
  const int g = w->image_size / w->patch;
  const size_t rows = (size_t)n_img * (g * g + (w->cls_emb ? 1 : 0));
  Carver c(ws);
  VitWs v;
  v.col = c.take(rows * w->kp * es);
  v.h = c.take(rows * w->hidden * es);
  v.qkv = c.take(rows * 3 * w->hidden * es);
  v.mlp = c.take(rows * w->inter_pad * es);
  v.gws_bytes = (size_t)srgpt_gemm_ws_bytes((int)rows, w->hidden);  // split-K only pays on the narrow (N = hidden) GEMMs
  v.gws = c.take(v.gws_bytes);
  v.total = c.off;
  return v;
}

// ---- LLM workspace ----
struct LlmWs {
  void *x, *h, *qkv, *attn, *gu, *act, *last;  // prefill
  void *xd, *qkvd, *attnd, *actd;              // decode
  float* dws;
  void* gws;  // split-K workspace of the prefill GEMMs
  size_t gws_bytes;
  float* amax_v;  // [batch, ARGMAX_BLOCKS] partial maxima
  int* amax_i;    // [batch, ARGMAX_BLOCKS] their indices
  int64_t* tok_emb;  // [batch] the token whose embedding row currently sits in xd (-1: none)
  void* smp;         // sampling: the slices' top-k candidates (srgpt_sample_ws_bytes)
  int* err;          // sticky error word of the decode step (bit 0: a caller-written st->tok outside the table), read by srgpt_llm_decode_sync_state
  float* rowss;   // decode, 2+ rows: two row-statistics tables [batch][SRGPT_ROWSS_STRIDE] (o_proj's and down_proj's output rows)
  void* a8;       // fp8_act: the e4m3 bytes of the current GEMM input [rows, max K]
  float* a8s;     // fp8_act: their per-row scales [rows]
  size_t total;
};
constexpr int ARGMAX_BLOCKS = 128;
LlmWs carve_llm(const srgpt_llm_weights* w, int batch, int max_tokens, void* ws) {
  const size_t es = dtype_size(w->dtype);
  const size_t rows = (size_t)batch * max_tokens;
  const size_t qkvw = (size_t)(w->heads + 2 * w->kv_heads) * w->head_dim;
  Carver c(ws);
  LlmWs l;
  l.x = c.take(rows * w->hidden * es);
  l.h = c.take(rows * w->hidden * es);
  l.qkv = c.take(rows * qkvw * es);
  l.attn = c.take(rows * (size_t)w->heads * w->head_dim * es);
  l.gu = c.take(rows * 2 * (size_t)w->inter * es);
  l.act = c.take(rows * (size_t)w->inter * es);
  l.last = c.take((size_t)batch * w->hidden * es);
  l.xd = c.take((size_t)batch * w->hidden * es);
  l.qkvd = c.take((size_t)batch * qkvw * es);
  l.attnd = c.take((size_t)batch * w->heads * w->head_dim * es);
  l.actd = c.take((size_t)batch * w->inter * es);
  l.dws = reinterpret_cast<float*>(c.take((size_t)srgpt_decode_attn_ws_floats(batch, w->heads, w->head_dim) * 4));
  l.gws_bytes = (size_t)srgpt_gemm_ws_bytes((int)rows, (int)(qkvw > (size_t)w->hidden ? qkvw : (size_t)w->hidden));
  l.gws = c.take(l.gws_bytes);
  l.amax_v = reinterpret_cast<float*>(c.take((size_t)batch * ARGMAX_BLOCKS * 4));
  l.amax_i = reinterpret_cast<int*>(c.take((size_t)batch * ARGMAX_BLOCKS * 4));
  l.tok_emb = reinterpret_cast<int64_t*>(c.take((size_t)batch * 8));
  l.err = reinterpret_cast<int*>(c.take(sizeof(int)));
  l.smp = c.take((size_t)srgpt_sample_ws_bytes(batch));
  l.rowss = reinterpret_cast<float*>(c.take((size_t)2 * batch * SRGPT_ROWSS_STRIDE * sizeof(float)));
  l.a8 = nullptr;
  l.a8s = nullptr;
  if (w->fp8_act) {
    size_t kmax = (size_t)w->hidden;
    if ((size_t)w->inter > kmax) kmax = (size_t)w->inter;
    if ((size_t)w->heads * w->head_dim > kmax) kmax = (size_t)w->heads * w->head_dim;
    l.a8 = c.take(rows * kmax);
    l.a8s = reinterpret_cast<float*>(c.take(rows * 4));
  }
  l.total = c.off;
  return l;
}

__global__ void set_int_kernel(int* p, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// greedy pick, stage 1: each block scans a contiguous slice of one row (first max wins, like torch.argmax)
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float* __restrict__ logits, float* __restrict__ pv,
                                                             int* __restrict__ pi, int V) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.y, nb = gridDim.x;
  const int per = (V + nb - 1) / nb;
  const int lo = blockIdx.x * per, hi = min(lo + per, V);
  const float* row = logits + (size_t)b * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = row[i];
    if (v > best) {  // ascending scan per thread: the first maximum is kept
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
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = best;
    si[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (sv[i] > best || (sv[i] == best && si[i] < bi)) {
        best = sv[i];
        bi = si[i];
      }
    pv[(size_t)b * nb + blockIdx.x] = best;
    pi[(size_t)b * nb + blockIdx.x] = bi;
  }
}

// stage 2 + bookkeeping: one wave per batch row merges the partials -> tok / out_ids[:, step] / pos / step, and (embed != NULL)
// the embedding rows of the picked tokens go straight into the residual-stream buffer of the next step (one launch fewer per token)
constexpr int ADVANCE_MAXB = 256;
__global__ __launch_bounds__(1024) void advance_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int nb, int64_t* tok,
                               int64_t* out_ids, int* pos, int* step, int B, int max_new, int bump_pos,
                               const unsigned char* __restrict__ embed, unsigned char* __restrict__ xd, int row_bytes,
                               int64_t* __restrict__ tok_emb, srgpt_sampling* __restrict__ sp) {
  __shared__ int picked[ADVANCE_MAXB];
  const int s = *step;
  const int lane = threadIdx.x & 63;
  // sampling with a top-k filter: sample_select_kernel already drew tok[b]; greedy and Gumbel-max sampling (top_k = 0): the slices'
  // maxima are merged here
  const bool drawn = sp != nullptr && sp->top_k > 0;
  for (int b = threadIdx.x >> 6; b < B; b += blockDim.x >> 6) {
    float best = -INFINITY;
    int bi = drawn ? (int)tok[b] : 0x7fffffff;
    for (int i = lane; i < (drawn ? 0 : nb); i += 64) {
      const float v = pv[(size_t)b * nb + i];
      const int ix = pi[(size_t)b * nb + i];
      if (v > best || (v == best && ix < bi)) {
        best = v;
        bi = ix;
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
    if (lane == 0) {
      const int64_t t = bi == 0x7fffffff ? 0 : bi;
      tok[b] = t;
      if (embed && b < ADVANCE_MAXB) {
        picked[b] = (int)t;
        tok_emb[b] = t;  // the row written below belongs to this token (embed_sync_kernel checks it)
      }
      if (s < max_new) out_ids[(size_t)b * max_new + s] = t;
      if (bump_pos) pos[b] += 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *step = s + 1;
    if (sp) sp->counter += 1;  // the next step draws from a fresh Philox counter
  }
  if (embed) {
    // the rows of the picked tokens, four independent 16-byte loads in flight per thread (one at a time, a dependent load -> store
    // chain per chunk, this copy was 8 of the kernel's 13 us at 8 sequences)
    const int per_row = row_bytes >> 4;  // 16-byte chunks per embedding row
    const int total = B * per_row, nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * nt) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * nt, total - 1);
        const int b = i / per_row, c = i - b * per_row;
        v[u] = *reinterpret_cast<const u32x4*>(embed + (size_t)picked[b] * row_bytes + (size_t)c * 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * nt;
        if (i < total) {
          const int b = i / per_row, c = i - b * per_row;
          *reinterpret_cast<u32x4*>(xd + (size_t)b * row_bytes + (size_t)c * 16) = v[u];
        }
      }
    }
  }
}

// First node of the CAPTURED decode step: xd must hold the embedding rows of st->tok.  Normally advance_kernel of the step before
// left exactly those rows (tok_emb == tok: B compares, nothing copied); a caller that wrote its own token into st->tok between
// replays -- valid under ABI 1/2, silently ignored in ABI 3 -- gets its row re-embedded here.  A caller-written id
// outside the table cannot be embedded: the step would run on the previous token's row and return plausible ids, so it sets a
// sticky error bit that srgpt_llm_decode_sync_state reports.
__global__ void embed_sync_kernel(const int64_t* __restrict__ tok, int64_t* __restrict__ tok_emb,
                                  const unsigned char* __restrict__ embed, unsigned char* __restrict__ xd, int B, int row_bytes,
                                  int64_t vocab, int* __restrict__ err) {
  const int per_row = row_bytes >> 4;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int64_t t = tok[b];
    if (t == tok_emb[b]) continue;  // uniform per block
    if (t < 0 || t >= vocab) {
      if (threadIdx.x == 0) atomicOr(err, 1);
      continue;
    }
    for (int c = threadIdx.x; c < per_row; c += blockDim.x)
      *reinterpret_cast<u32x4*>(xd + (size_t)b * row_bytes + (size_t)c * 16) =
          *reinterpret_cast<const u32x4*>(embed + (size_t)t * row_bytes + (size_t)c * 16);
    __syncthreads();
    if (threadIdx.x == 0) tok_emb[b] = t;
  }
}

// the eager step embeds st->tok itself (srgpt_embed_rows): record whose rows are in place, so that the invariant "xd holds the
// rows of tok_emb" does not depend on a greedy_pick always following
__global__ void record_embedded_kernel(const int64_t* __restrict__ tok, int64_t* __restrict__ tok_emb, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) tok_emb[b] = tok[b];
}

// the next step's embedding rows can come from advance_kernel when they are whole 16-byte chunks and the batch fits its LDS list
static inline bool advance_embeds(const srgpt_llm_weights* w, const srgpt_llm_state* st) {
  return st->batch <= ADVANCE_MAXB && ((size_t)w->hidden * dtype_size(w->dtype)) % 16 == 0;
}

// the next token of every sequence from st->logits: argmax (st->sampling == NULL) or a draw (sample.hip), then the bookkeeping
static int greedy_pick(const srgpt_llm_weights* w, srgpt_llm_state* st, const LlmWs& d, int bump_pos, hipStream_t s) {
  const int B = st->batch;
  const bool emb = advance_embeds(w, st);
  if (st->sampling) {
    SRGPT_CHECK(srgpt_sample_slices() == ARGMAX_BLOCKS, SRGPT_ERR_STATE, "sampling: slice count differs from the argmax merge's");
    SRGPT_TRY(srgpt_sample_launch(st->logits, st->sampling, st->tok, d.smp, d.amax_v, d.amax_i, d.err, B, w->vocab, s));
  } else {
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(ARGMAX_BLOCKS, B), dim3(256), 0, s, st->logits, d.amax_v, d.amax_i, w->vocab);
  }
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(B > 4 ? 1024 : 256), 0, s, d.amax_v, d.amax_i, ARGMAX_BLOCKS, st->tok, st->out_ids,
                     st->pos, st->step, B, st->max_new, bump_pos, emb ? reinterpret_cast<const unsigned char*>(w->embed) : nullptr,
                     reinterpret_cast<unsigned char*>(d.xd), (int)((size_t)w->hidden * dtype_size(w->dtype)), d.tok_emb, st->sampling);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

}  // namespace

struct srgpt_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

// ================================================================================================
// vision tower
// ================================================================================================
extern "C" int64_t srgpt_vit_ws_bytes(const srgpt_vit_weights* w, int n_img) {
  if (!w || n_img <= 0) return -1;
  return (int64_t)carve_vit(w, n_img, nullptr).total;
}

extern "C" int srgpt_vit_forward(const srgpt_vit_weights* w, const void* images, void* out, void* ws, int n_img,
                                 srgpt_stream_t stream) {
  SRGPT_CHECK(w && images && out && ws && n_img > 0, SRGPT_ERR_ARG, "srgpt_vit_forward: bad args");
  // image_size need not be a multiple of patch: a 'valid' conv drops the remainder (384 px / 14 -> 27 patches)
  SRGPT_CHECK(w->hidden % w->heads == 0 && w->image_size >= w->patch, SRGPT_ERR_ARG, "srgpt_vit_forward: bad config");
  SRGPT_CHECK(w->inter_pad >= w->inter && w->inter_pad % 8 == 0, SRGPT_ERR_ARG, "srgpt_vit_forward: inter_pad %d < inter %d",
              w->inter_pad, w->inter);
  const int dt = w->dtype, C = w->hidden, I = w->inter, IP = w->inter_pad, H = w->heads, hd = C / H;
  const int g = w->image_size / w->patch, gg = g * g, L = gg + (w->cls_emb ? 1 : 0), rows = n_img * L;
  const VitWs v = carve_vit(w, n_img, ws);
  void* x = out;  // residual stream lives in the output buffer
  SRGPT_TRY(srgpt_im2col(images, v.col, n_img, w->image_size, w->patch, w->kp, dt, stream));
  if (!w->cls_emb) {
    // SigLIP: conv-as-GEMM + bias, then + position embedding (row m uses pos_emb[m % L])
    SRGPT_TRY(srgpt_gemm(v.col, w->patch_w, w->patch_b, w->pos_emb, x, rows, C, w->kp, w->kp, C, SRGPT_ACT_NONE, 0, L, 0,
                         SRGPT_OUT_PLAIN, 0, v.gws, (int64_t)v.gws_bytes, dt, stream));
  } else {
    // CLIP: patch GEMM (no bias) -> [cls | patches] + position embedding -> pre_layrnorm
    SRGPT_TRY(srgpt_gemm(v.col, w->patch_w, w->patch_b, nullptr, v.h, n_img * gg, C, w->kp, w->kp, C, SRGPT_ACT_NONE, 0, 0, 0,
                         SRGPT_OUT_PLAIN, 0, v.gws, (int64_t)v.gws_bytes, dt, stream));
    SRGPT_TRY(srgpt_vit_assemble_cls(v.h, w->cls_emb, w->pos_emb, x, n_img, gg, C, dt, stream));
    if (w->pre_ln_w) SRGPT_TRY(srgpt_layernorm(x, w->pre_ln_w, w->pre_ln_b, x, rows, C, w->eps, SRGPT_ACT_NONE, dt, stream));
  }
  const float scale = 1.0f / sqrtf((float)hd);
  const char* qkv = reinterpret_cast<const char*>(v.qkv);
  const size_t es = dtype_size(dt);
  if (IP > I)  // the pad columns of the MLP activation buffer: zero once per forward (fc1 never writes them, w2 is zero there)
    SRGPT_HIP_TRY(hipMemset2DAsync(reinterpret_cast<char*>(v.mlp) + (size_t)I * es, (size_t)IP * es, 0, (size_t)(IP - I) * es,
                                   (size_t)rows, as_stream(stream)),
                  "srgpt_vit_forward: zeroing the activation pad");
  // The LayerNorm that follows out_proj / fc2 rides in the product's split-K reduction when there is one (srgpt_gemm_norm: bit-identical
  // to the two launches); ln1 of the first layer is the only stand-alone norm.
  for (int l = 0; l < w->n_layers_run; ++l) {
    if (l == 0) SRGPT_TRY(srgpt_layernorm(x, w->ln1_w[l], w->ln1_b[l], v.h, rows, C, w->eps, SRGPT_ACT_NONE, dt, stream));
    SRGPT_TRY(srgpt_gemm(v.h, w->wqkv[l], w->bqkv[l], nullptr, v.qkv, rows, 3 * C, C, C, 3 * C, SRGPT_ACT_NONE, 0, 0, 0,
                         SRGPT_OUT_PLAIN, 0, v.gws, (int64_t)v.gws_bytes, dt, stream));
    SRGPT_TRY(srgpt_attention(qkv, qkv + (size_t)C * es, qkv + (size_t)2 * C * es, v.h, n_img, L, L, H, H, hd,
                              (int64_t)L * 3 * C, 3 * C, hd, (int64_t)L * 3 * C, 3 * C, hd, (int64_t)L * 3 * C, 3 * C, hd,
                              scale, 0, nullptr, dt, stream));
    SRGPT_TRY(srgpt_gemm_norm(v.h, w->wo[l], w->bo[l], x, x, rows, C, C, v.gws, (int64_t)v.gws_bytes, SRGPT_NORM_LAYER, w->ln2_w[l],
                              w->ln2_b[l], v.h, w->eps, dt, stream));
    SRGPT_TRY(srgpt_gemm(v.h, w->w1[l], w->b1[l], nullptr, v.mlp, rows, I, C, C, IP, w->act, 0, 0, 0,
                         SRGPT_OUT_PLAIN, 0, v.gws, (int64_t)v.gws_bytes, dt, stream));
    if (l + 1 < w->n_layers_run)
      SRGPT_TRY(srgpt_gemm_norm(v.mlp, w->w2[l], w->b2[l], x, x, rows, C, IP, v.gws, (int64_t)v.gws_bytes, SRGPT_NORM_LAYER,
                                w->ln1_w[l + 1], w->ln1_b[l + 1], v.h, w->eps, dt, stream));
    else
      SRGPT_TRY(srgpt_gemm(v.mlp, w->w2[l], w->b2[l], x, x, rows, C, IP, IP, C, SRGPT_ACT_NONE, 0, 0, 0, SRGPT_OUT_PLAIN, 0, v.gws,
                           (int64_t)v.gws_bytes, dt, stream));
  }
  return SRGPT_OK;
}

// ================================================================================================
// Llama decoder
// ================================================================================================
extern "C" int64_t srgpt_llm_ws_bytes(const srgpt_llm_weights* w, int batch, int max_tokens) {
  if (!w || batch <= 0 || max_tokens <= 0) return -1;
  return (int64_t)carve_llm(w, batch, max_tokens, nullptr).total;
}

static int check_llm(const srgpt_llm_weights* w, const srgpt_llm_state* st) {
  SRGPT_CHECK(w && st, SRGPT_ERR_ARG, "llm: null weights/state");
  SRGPT_CHECK(st->kcache && st->vcache && st->pos && st->tok && st->out_ids && st->step && st->ws && st->logits,
              SRGPT_ERR_ARG, "llm: state has null buffers");
  SRGPT_CHECK(w->heads % w->kv_heads == 0 && w->head_dim % 2 == 0, SRGPT_ERR_ARG, "llm: bad head config");
  SRGPT_CHECK(st->batch > 0 && st->max_pos > 0 && st->ws_tokens > 0 && st->max_new > 0, SRGPT_ERR_ARG, "llm: bad state sizes");
  return SRGPT_OK;
}

// out[b, :] = x[b, lens[b] - 1, :] (16-byte chunks) and pos[b] = lens[b]
template <typename T>
__global__ void gather_last_rows_kernel(const T* __restrict__ x, const int* __restrict__ lens, T* __restrict__ out,
                                        int* __restrict__ pos, int Tlen, int Hd) {
  const int b = blockIdx.x;
  const int len = min(max(lens[b], 1), Tlen);
  constexpr int VEC = Vec16<T>::N;
  const Vec16<T>* src = reinterpret_cast<const Vec16<T>*>(x + ((size_t)b * Tlen + (len - 1)) * Hd);
  Vec16<T>* dst = reinterpret_cast<Vec16<T>*>(out + (size_t)b * Hd);
  for (int c = threadIdx.x; c < Hd / VEC; c += blockDim.x) dst[c] = src[c];
  if (threadIdx.x == 0) pos[b] = len;
}

static int prefill_impl(const srgpt_llm_weights* w, srgpt_llm_state* st, const void* inputs_embeds, int T, const int* lens,
                        float* all_logits, void* hidden_out, srgpt_stream_t stream) {
  SRGPT_TRY(check_llm(w, st));
  SRGPT_CHECK(inputs_embeds && T > 0 && T <= st->max_pos, SRGPT_ERR_ARG, "srgpt_llm_prefill: T=%d exceeds max_pos=%d", T,
              st->max_pos);
  const int dt = w->dtype, Hd = w->hidden, I = w->inter, Hq = w->heads, Hkv = w->kv_heads, D = w->head_dim;
  const int B = st->batch, rows = B * T, QW = (Hq + 2 * Hkv) * D;
  const size_t es = dtype_size(dt);
  hipStream_t s = as_stream(stream);
  SRGPT_CHECK(T <= st->ws_tokens, SRGPT_ERR_ARG, "srgpt_llm_prefill: T=%d exceeds ws_tokens=%d", T, st->ws_tokens);
  const LlmWs l = carve_llm(w, B, st->ws_tokens, st->ws);
  const size_t layer_kv = (size_t)B * Hkv * st->max_pos * D * es;
  const size_t hid_bytes = (size_t)rows * Hd * es;
  {  // the decode attention's arrival tickets / sync words must be zero before the first step: every prefill re-arms them, so a
     // caller-allocated (never zeroed) workspace or an aborted launch cannot leave the decode steps merging nothing
    size_t sync_bytes = 0;
    void* sync = srgpt_decode_attn_sync_words(l.dws, B, Hq, D, &sync_bytes);
    SRGPT_HIP_TRY(hipMemsetAsync(sync, 0, sync_bytes, s), "srgpt_llm_prefill: re-arming the decode tickets");
    // no embedding row is in place for the new sequences (-1 matches no token id)
    SRGPT_HIP_TRY(hipMemsetAsync(l.tok_emb, 0xFF, (size_t)B * sizeof(int64_t), s), "srgpt_llm_prefill: resetting the embedded-token record");
    SRGPT_HIP_TRY(hipMemsetAsync(l.err, 0, sizeof(int), s), "srgpt_llm_prefill: clearing the decode error word");
  }
  if (hipMemcpyAsync(l.x, inputs_embeds, hid_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    srgpt_set_error("srgpt_llm_prefill: memcpy failed");
    return SRGPT_ERR_LAUNCH;
  }
  if (hidden_out) SRGPT_HIP_TRY(hipMemcpyAsync(hidden_out, l.x, hid_bytes, hipMemcpyDeviceToDevice, s), "srgpt_llm_prefill: hidden-state copy");
  const float scale = 1.0f / sqrtf((float)D);
  // fp8 copies present -> they are the weights (srgpt_gemm_w8 / srgpt_gemv_w8); the dtype matrices are not touched
  const bool w8 = w->wqkv8 != nullptr;
  if (w8)
    SRGPT_CHECK(dt == SRGPT_BF16 && w->wo8 && w->wgu8 && w->wdown8 && w->lm_head8 && w->wqkv_scale && w->wo_scale &&
                    w->wgu_scale && w->wdown_scale && w->lm_head_scale,
                SRGPT_ERR_ARG, "srgpt_llm_prefill: fp8 weights need bf16 activations and all five matrices + scales");
  const bool a8 = w8 && w->fp8_act != 0;  // W8A8 prefill: per-token e4m3 activations on the fp8 matrix pipe (include/srgpt.h)
  if (a8)
    SRGPT_CHECK(Hd % 128 == 0 && I % 128 == 0 && (Hq * D) % 128 == 0 && Hd >= 256 && I >= 256 && Hq * D >= 256,
                SRGPT_ERR_UNSUPPORTED, "srgpt_llm_prefill: fp8_act needs hidden, inter and heads * head_dim to be multiples of 128");
  auto mm = [&](const void* a, const void* Wd, const void* W8p, const float* sc, const void* res, void* out, int N, int K,
                int f32, void* gws, int64_t gws_bytes) -> int {
    if (a8 && !f32) {  // the layers' four products; the all-position lm_head (fp32 logits, parity hook) stays W8A16
      SRGPT_TRY(srgpt_quant_rows_e4m3(a, l.a8, l.a8s, rows, K, K, stream));
      return srgpt_gemm_w8a8(l.a8, l.a8s, W8p, sc, nullptr, res, out, rows, N, K, K, N, 0, gws, gws_bytes, stream);
    }
    if (w8) return srgpt_gemm_w8(a, W8p, sc, nullptr, res, out, rows, N, K, K, N, SRGPT_ACT_NONE, f32, gws, gws_bytes, stream);
    return srgpt_gemm(a, Wd, nullptr, res, out, rows, N, K, K, N, SRGPT_ACT_NONE, 0, 0, f32, SRGPT_OUT_PLAIN, 0, gws, gws_bytes, dt,
                      stream);
  };
  // W8A8: the producer of a GEMM's input rows (RMSNorm, SwiGLU) is fused into their per-token quantisation -- the bf16
  // intermediate is neither written nor read (bit-identical to the two launches: tests/test_gpu_fp8_mfma.py)
  auto mm8 = [&](const void* W8p, const float* sc, const void* res, void* out, int N, int K) -> int {
    return srgpt_gemm_w8a8(l.a8, l.a8s, W8p, sc, nullptr, res, out, rows, N, K, K, N, 0, l.gws, (int64_t)l.gws_bytes, stream);
  };
  const bool fuse8 = a8 && Hd <= 16384 && I <= 16384;
  const bool plain = !w8;  // dtype matrices: srgpt_gemm
  bool h_ready = false;   // l.h already holds RMSNorm(l.x) under the coming layer's attn_norm
  for (int i = 0; i < w->layers; ++i) {
    char* kc = reinterpret_cast<char*>(st->kcache) + (size_t)i * layer_kv;
    char* vc = reinterpret_cast<char*>(st->vcache) + (size_t)i * layer_kv;
    if (fuse8) {
      SRGPT_TRY(srgpt_quant_rows_e4m3_rmsnorm(l.x, w->attn_norm[i], w->rms_eps, l.a8, l.a8s, rows, Hd, Hd, stream));
      SRGPT_TRY(mm8(w->wqkv8[i], w->wqkv_scale[i], nullptr, l.qkv, QW, Hd));
    } else {
      if (!h_ready) SRGPT_TRY(srgpt_rmsnorm(l.x, w->attn_norm[i], l.h, rows, Hd, w->rms_eps, dt, stream));
      if (!plain)
        SRGPT_TRY(mm(l.h, w->wqkv[i], w8 ? w->wqkv8[i] : nullptr, w8 ? w->wqkv_scale[i] : nullptr, nullptr, l.qkv, QW, Hd, 0, l.gws,
                     (int64_t)l.gws_bytes));
    }
    if (plain && !fuse8)  // projection + RoPE + cache append: the rotation rides in the split-K reduction
      SRGPT_TRY(srgpt_gemm_rope_kv_append(l.h, w->wqkv[i], l.qkv, Hd, l.gws, (int64_t)l.gws_bytes, kc, vc, nullptr, w->rope_cos,
                                          w->rope_sin, B, T, Hq, Hkv, D, st->max_pos, dt, stream));
    else
      SRGPT_TRY(srgpt_rope_kv_append(l.qkv, kc, vc, nullptr, w->rope_cos, w->rope_sin, B, T, Hq, Hkv, D, st->max_pos, dt,
                                     stream));
    SRGPT_TRY(srgpt_attention(l.qkv, kc, vc, l.attn, B, T, T, Hq, Hkv, D, (int64_t)T * QW, QW, D,
                              (int64_t)Hkv * st->max_pos * D, D, (int64_t)st->max_pos * D,
                              (int64_t)Hkv * st->max_pos * D, D, (int64_t)st->max_pos * D, scale, 1, nullptr, dt, stream));
    // plain weights: the RMSNorm that follows o_proj / down_proj rides in the product's split-K reduction (srgpt_gemm_norm:
    // bit-identical to the two launches, one launch and one pass over the rows less per norm)
    if (plain)
      SRGPT_TRY(srgpt_gemm_norm(l.attn, w->wo[i], nullptr, l.x, l.x, (int)rows, Hd, Hq * D, l.gws, (int64_t)l.gws_bytes, SRGPT_NORM_RMS,
                                w->mlp_norm[i], nullptr, l.h, w->rms_eps, dt, stream));
    else
      SRGPT_TRY(mm(l.attn, w->wo[i], w8 ? w->wo8[i] : nullptr, w8 ? w->wo_scale[i] : nullptr, l.x, l.x, Hd, Hq * D, 0, l.gws,
                   (int64_t)l.gws_bytes));
    if (fuse8) {
      SRGPT_TRY(srgpt_quant_rows_e4m3_rmsnorm(l.x, w->mlp_norm[i], w->rms_eps, l.a8, l.a8s, rows, Hd, Hd, stream));
      SRGPT_TRY(mm8(w->wgu8[i], w->wgu_scale[i], nullptr, l.gu, 2 * I, Hd));
      SRGPT_TRY(srgpt_quant_rows_e4m3_swiglu(l.gu, l.a8, l.a8s, rows, I, stream));
      SRGPT_TRY(mm8(w->wdown8[i], w->wdown_scale[i], l.x, l.x, Hd, I));
    } else {
      if (!plain) SRGPT_TRY(srgpt_rmsnorm(l.x, w->mlp_norm[i], l.h, rows, Hd, w->rms_eps, dt, stream));
      if (plain) {  // gate / up + SiLU * up: the activation is the product's epilogue on the whole-M kernel
        SRGPT_TRY(srgpt_gemm_swiglu(l.h, w->wgu[i], l.act, (int)rows, I, Hd, l.gu, l.gws, (int64_t)l.gws_bytes, dt, stream));
      } else {
        SRGPT_TRY(mm(l.h, w->wgu[i], w8 ? w->wgu8[i] : nullptr, w8 ? w->wgu_scale[i] : nullptr, nullptr, l.gu, 2 * I, Hd, 0, l.gws,
                     (int64_t)l.gws_bytes));
        SRGPT_TRY(srgpt_silu_mul(l.gu, l.act, rows, I, dt, stream));
      }
      h_ready = plain && i + 1 < w->layers;
      if (h_ready)
        SRGPT_TRY(srgpt_gemm_norm(l.act, w->wdown[i], nullptr, l.x, l.x, (int)rows, Hd, I, l.gws, (int64_t)l.gws_bytes, SRGPT_NORM_RMS,
                                  w->attn_norm[i + 1], nullptr, l.h, w->rms_eps, dt, stream));
      else
        SRGPT_TRY(mm(l.act, w->wdown[i], w8 ? w->wdown8[i] : nullptr, w8 ? w->wdown_scale[i] : nullptr, l.x, l.x, Hd, I, 0, l.gws,
                     (int64_t)l.gws_bytes));
    }
    if (hidden_out)
      SRGPT_HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(hidden_out) + (size_t)(i + 1) * hid_bytes, l.x, hid_bytes,
                                   hipMemcpyDeviceToDevice, s),
                    "srgpt_llm_prefill: hidden-state copy");
  }
  if (all_logits) {
    SRGPT_TRY(srgpt_rmsnorm(l.x, w->final_norm, l.h, rows, Hd, w->rms_eps, dt, stream));
    SRGPT_TRY(mm(l.h, w->lm_head, w->lm_head8, w->lm_head_scale, nullptr, all_logits, w->vocab, Hd, 1, nullptr, 0));
  }
  // last position of every sequence -> logits (final norm fused into the GEMV prologue)
  if (lens) {  // right-padded ragged batch: row b ends at lens[b] - 1 and decoding continues from position lens[b]
    SRGPT_CHECK(Hd % (16 / (int)es) == 0, SRGPT_ERR_ARG, "srgpt_llm_prefill_ragged: hidden %d not a multiple of 16 bytes", Hd);
    if (dt == SRGPT_BF16)
      hipLaunchKernelGGL(gather_last_rows_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)l.x, lens, (bf16_t*)l.last,
                         st->pos, T, Hd);
    else
      hipLaunchKernelGGL(gather_last_rows_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)l.x, lens, (float*)l.last,
                         st->pos, T, Hd);
    SRGPT_LAUNCH_CHECK();
  } else {
    if (hipMemcpy2DAsync(l.last, (size_t)Hd * es, reinterpret_cast<char*>(l.x) + (size_t)(T - 1) * Hd * es,
                         (size_t)T * Hd * es, (size_t)Hd * es, B, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      srgpt_set_error("srgpt_llm_prefill: gather of last rows failed");
      return SRGPT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(set_int_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, st->pos, B, T);
    SRGPT_LAUNCH_CHECK();
  }
  if (w8)
    SRGPT_TRY(srgpt_gemv_w8(l.last, w->lm_head8, w->lm_head_scale, w->final_norm, w->rms_eps, nullptr, st->logits, B, w->vocab, Hd,
                            0, 1, stream));
  else
    SRGPT_TRY(srgpt_gemv(l.last, w->lm_head, w->final_norm, w->rms_eps, nullptr, st->logits, B, w->vocab, Hd, 0, 1, dt, stream));
  return SRGPT_OK;
}

extern "C" int srgpt_llm_prefill(const srgpt_llm_weights* w, srgpt_llm_state* st, const void* inputs_embeds, int T,
                                 float* all_logits, void* hidden_out, srgpt_stream_t stream) {
  return prefill_impl(w, st, inputs_embeds, T, nullptr, all_logits, hidden_out, stream);
}

extern "C" int srgpt_llm_prefill_ragged(const srgpt_llm_weights* w, srgpt_llm_state* st, const void* inputs_embeds, int T,
                                        const int* lens, float* all_logits, void* hidden_out, srgpt_stream_t stream) {
  SRGPT_CHECK(lens, SRGPT_ERR_ARG, "srgpt_llm_prefill_ragged: lens is null");
  return prefill_impl(w, st, inputs_embeds, T, lens, all_logits, hidden_out, stream);
}

extern "C" int srgpt_llm_sample_first(const srgpt_llm_weights* w, srgpt_llm_state* st, srgpt_stream_t stream) {
  SRGPT_TRY(check_llm(w, st));
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(64), 0, s, st->step, 1, 0);
  const LlmWs d = carve_llm(w, st->batch, st->ws_tokens, st->ws);
  return greedy_pick(w, st, d, 0, s);
}

// embed_first = false: the residual-stream buffer already holds the embeddings of st->tok (written by the advance_kernel of the
// step before: the graph-captured greedy loop); the public entry always embeds (st->tok may have been set by the caller)
static int decode_step_impl(const srgpt_llm_weights* w, srgpt_llm_state* st, srgpt_stream_t stream, bool embed_first) {
  SRGPT_TRY(check_llm(w, st));
  const int dt = w->dtype, Hd = w->hidden, I = w->inter, Hq = w->heads, Hkv = w->kv_heads, D = w->head_dim;
  const int B = st->batch, QW = (Hq + 2 * Hkv) * D;
  const size_t es = dtype_size(dt);
  hipStream_t s = as_stream(stream);
  const LlmWs d = carve_llm(w, B, st->ws_tokens, st->ws);  // same carve as prefill (sized by ws_tokens)
  const size_t layer_kv = (size_t)B * Hkv * st->max_pos * D * es;
  if (embed_first) {
    SRGPT_TRY(srgpt_embed_rows(w->embed, st->tok, d.xd, B, Hd, dt, stream));
    hipLaunchKernelGGL(record_embedded_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, st->tok, d.tok_emb, B);
    SRGPT_LAUNCH_CHECK();
  } else {  // captured step: one tiny launch that copies nothing unless the caller changed st->tok behind the graph's back
    hipLaunchKernelGGL(embed_sync_kernel, dim3(B < 64 ? B : 64), dim3(256), 0, s, st->tok, d.tok_emb,
                       reinterpret_cast<const unsigned char*>(w->embed), reinterpret_cast<unsigned char*>(d.xd), B, (int)((size_t)Hd * es),
                       (int64_t)w->vocab, d.err);
    SRGPT_LAUNCH_CHECK();
  }
  // fp8 copies present -> the decode step streams them (W8A16, half the bytes per token)
  const bool w8 = w->wqkv8 != nullptr;
  if (w8)
    SRGPT_CHECK(dt == SRGPT_BF16 && w->wo8 && w->wgu8 && w->wdown8 && w->lm_head8 && w->wqkv_scale && w->wo_scale &&
                    w->wgu_scale && w->wdown_scale && w->lm_head_scale,
                SRGPT_ERR_ARG, "srgpt_llm_decode_step: fp8 weights need bf16 activations and all five matrices + scales");
  // 2+ bf16 rows (the MFMA kernel): o_proj / down_proj publish the sum of squares of the rows they write, the RMSNorm of the next
  // product reads 512 partial sums per row instead of re-reading every row in every block (skinny.hip; layer 0's q/k/v normalises
  // the embedding rows itself).  The per-op form of exactly this sequence is srgpt_gemv_rowss.
  const bool pub = srgpt_gemv_rowss_supported(B, dt, w8 ? 1 : 0) != 0 && SRGPT_KNOB("SRGPT_DECODE_ROWSS", 1) != 0;
  float* const ss_attn = d.rowss;                                       // rows after the attention block's residual add
  float* const ss_mlp = d.rowss + (size_t)B * SRGPT_ROWSS_STRIDE;       // rows after the MLP block's
  // packed copy of layer i's matrix `which` (0 wqkv, 1 wo, 2 wgu, 3 wdown) for the MFMA kernel, or NULL: stream the row-major one
  auto pk = [&](const void* const* arr, int i) -> const void* { return pub && w8 && arr != nullptr ? arr[i] : nullptr; };
  auto mv = [&](const void* x, const void* Wd, const void* W8p, const float* sc, const void* norm, const void* res, void* out,
                int N, int K, int swiglu, int f32, const float* ss_in, float* ss_out, const void* W8pk = nullptr, int pk_rows = 0) -> int {
    if (pub && W8pk != nullptr)
      return srgpt_gemv_rowss(x, nullptr, W8pk, sc, norm, w->rms_eps, res, out, B, N, K, swiglu, f32, ss_in, ss_out, pk_rows, stream);
    if (pub) return srgpt_gemv_rowss(x, Wd, W8p, sc, norm, w->rms_eps, res, out, B, N, K, swiglu, f32, ss_in, ss_out, 0, stream);
    if (w8) return srgpt_gemv_w8(x, W8p, sc, norm, w->rms_eps, res, out, B, N, K, swiglu, f32, stream);
    return srgpt_gemv(x, Wd, norm, w->rms_eps, res, out, B, N, K, swiglu, f32, dt, stream);
  };
  for (int i = 0; i < w->layers; ++i) {
    char* kc = reinterpret_cast<char*>(st->kcache) + (size_t)i * layer_kv;
    char* vc = reinterpret_cast<char*>(st->vcache) + (size_t)i * layer_kv;
    SRGPT_TRY(mv(d.xd, w->wqkv[i], w8 ? w->wqkv8[i] : nullptr, w8 ? w->wqkv_scale[i] : nullptr, w->attn_norm[i], nullptr,
                 d.qkvd, QW, Hd, 0, 0, i > 0 ? ss_mlp : nullptr, nullptr, pk(w->wqkv8p, i), w->pk_rows_qkv));
    // the attention launch also pulls o_proj's weights into L2 (HBM is idle while it runs).  Round 3 built the next step -- o_proj
    // itself inside this launch, weights in registers, agent-scope hand-off -- bit-exact and 3 us per layer SLOWER
    // (profiles/r03_fused_attention_oproj.txt)
    SRGPT_TRY(srgpt_decode_attention_pf(d.qkvd, kc, vc, st->pos, w->rope_cos, w->rope_sin, d.attnd, d.dws, B, Hq, Hkv, D,
                                        st->max_pos, dt, pk(w->wo8p, i) ? pk(w->wo8p, i) : (w8 ? w->wo8[i] : w->wo[i]), Hd, Hq * D,
                                        w8 ? 1 : 0, pk(w->wo8p, i) ? w->pk_rows_o : 0, stream));
    SRGPT_TRY(mv(d.attnd, w->wo[i], w8 ? w->wo8[i] : nullptr, w8 ? w->wo_scale[i] : nullptr, nullptr, d.xd, d.xd, Hd,
                 Hq * D, 0, 0, nullptr, ss_attn, pk(w->wo8p, i), w->pk_rows_o));
    SRGPT_TRY(mv(d.xd, w->wgu[i], w8 ? w->wgu8[i] : nullptr, w8 ? w->wgu_scale[i] : nullptr, w->mlp_norm[i], nullptr,
                 d.actd, I, Hd, 1, 0, ss_attn, nullptr, pk(w->wgu8p, i), w->pk_rows_gu));
    SRGPT_TRY(mv(d.actd, w->wdown[i], w8 ? w->wdown8[i] : nullptr, w8 ? w->wdown_scale[i] : nullptr, nullptr, d.xd, d.xd,
                 Hd, I, 0, 0, nullptr, ss_mlp, pk(w->wdown8p, i), w->pk_rows_down));
  }
  SRGPT_TRY(mv(d.xd, w->lm_head, w->lm_head8, w->lm_head_scale, w->final_norm, nullptr, st->logits, w->vocab, Hd, 0, 1,
               w->layers > 0 ? ss_mlp : nullptr, nullptr));
  return greedy_pick(w, st, d, 1, s);
}

extern "C" int srgpt_llm_decode_step(const srgpt_llm_weights* w, srgpt_llm_state* st, srgpt_stream_t stream) {
  return decode_step_impl(w, st, stream, true);
}

// Health of the decode steps since the last prefill (synchronises `stream` ONCE: both read-backs are queued, then one wait):
// (1) between steps every arrival ticket of the decode attention (the split that draws the last ticket merges and re-arms it) must
// be zero again -- a non-zero ticket means a launch was aborted or merged nothing, and every later step would silently use stale
// attention output; (2) the sticky error word of the captured step (a caller-written token id outside the embedding table).
extern "C" int srgpt_llm_decode_sync_state(const srgpt_llm_weights* w, const srgpt_llm_state* st, srgpt_stream_t stream) {
  SRGPT_TRY(check_llm(w, st));
  const LlmWs d = carve_llm(w, st->batch, st->ws_tokens, st->ws);
  size_t bytes = 0;
  void* sync = srgpt_decode_attn_sync_words(d.dws, st->batch, w->heads, w->head_dim, &bytes);
  std::vector<int> host(bytes / sizeof(int) + 1);
  SRGPT_HIP_TRY(hipMemcpyAsync(host.data(), sync, bytes, hipMemcpyDeviceToHost, as_stream(stream)), "srgpt_llm_decode_sync_state: copy");
  SRGPT_HIP_TRY(hipMemcpyAsync(&host.back(), d.err, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)), "srgpt_llm_decode_sync_state: copy");
  SRGPT_HIP_TRY(hipStreamSynchronize(as_stream(stream)), "srgpt_llm_decode_sync_state: synchronize");
  const size_t n = host.size() - 1;
  for (size_t i = 0; i < n; ++i)
    SRGPT_CHECK(host[i] == 0, SRGPT_ERR_STATE, "decode step: arrival ticket %zu of %zu is %d between steps (expected 0)", i, n, host[i]);
  SRGPT_CHECK((host.back() & 4) == 0, SRGPT_ERR_UNSUPPORTED,
              "decode step: sampling parameters the device sampler does not serve (top_k > 64, or top_p < 1 without top_k): the draws "
              "used another distribution");
  SRGPT_CHECK((host.back() & 2) == 0, SRGPT_ERR_STATE,
              "decode step: more than 256 vocabulary entries tie at the top-k threshold of a sampling step (kept set truncated)");
  SRGPT_CHECK((host.back() & 1) == 0, SRGPT_ERR_STATE,
              "decode step: a token id written into st->tok lies outside the embedding table [0, %d); the step ran on a stale row",
              w->vocab);
  return SRGPT_OK;
}

// ================================================================================================
// hipGraph of one decode step
// ================================================================================================
extern "C" int srgpt_llm_decode_graph_create(const srgpt_llm_weights* w, srgpt_llm_state* st, srgpt_stream_t stream,
                                             srgpt_graph** out) {
  SRGPT_CHECK(out, SRGPT_ERR_ARG, "srgpt_llm_decode_graph_create: null out");
  SRGPT_TRY(check_llm(w, st));
  hipStream_t s = as_stream(stream);
  (void)srgpt_device_cus();  // make sure no device query happens inside the capture
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    srgpt_set_error("hipStreamBeginCapture failed");
    return SRGPT_ERR_STATE;
  }
  // replays continue from the token the previous step (or srgpt_llm_sample_first) picked: its embedding is already in place
  const int rc = decode_step_impl(w, st, stream, !advance_embeds(w, st));
  const hipError_t ee = hipStreamEndCapture(s, &graph);
  if (rc != SRGPT_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (ee != hipSuccess || !graph) {
    srgpt_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ee));
    return SRGPT_ERR_STATE;
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (ei != hipSuccess) {
    (void)hipGraphDestroy(graph);
    srgpt_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ei));
    return SRGPT_ERR_STATE;
  }
  *out = new srgpt_graph{graph, exec};
  return SRGPT_OK;
}

extern "C" int srgpt_graph_launch(srgpt_graph* g, int times, srgpt_stream_t stream) {
  SRGPT_CHECK(g && times >= 0, SRGPT_ERR_ARG, "srgpt_graph_launch: bad args");
  for (int i = 0; i < times; ++i) {
    const hipError_t e = hipGraphLaunch(g->exec, as_stream(stream));
    if (e != hipSuccess) {
      srgpt_set_error("hipGraphLaunch failed: %s", hipGetErrorString(e));
      return SRGPT_ERR_LAUNCH;
    }
  }
  return SRGPT_OK;
}

extern "C" int srgpt_graph_destroy(srgpt_graph* g) {
  if (!g) return SRGPT_OK;
  (void)hipGraphExecDestroy(g->exec);  // best effort: nothing useful to report from a destructor path
  (void)hipGraphDestroy(g->graph);
  delete g;
  return SRGPT_OK;
}
