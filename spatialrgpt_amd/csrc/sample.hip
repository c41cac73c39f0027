// Device-side sampling of the next token: temperature -> top-k -> top-p -> categorical draw, the warper chain HF's
// GenerationMixin.sample applies for the reference's interactive callers (demo/gradio_web_server_multi.py:202-213: do_sample,
// temperature 0.2, top_k = the transformers==4.37.2 default 50; llava/eval/model_vqa.py:66-80: temperature / top_p / num_beams).
// Round 3 ran this as torch ops per token outside the captured step; here it is three launches INSIDE the hipGraph of the decode
// step, driven by a device-resident parameter block (one graph serves every temperature / top-k / top-p / seed).
//
//   launch 1  sample_partial_kernel   grid (128 vocabulary slices, batch): the slice's scores logits / T as order-preserving
//             integer keys in LDS, the slice's top-k by a 4-pass radix select (LDS histograms, one wave scans the 256 bins),
//             <= 64 candidates (key, index) per slice (ties at the slice's k-th score included while the slots last).  top_k = 0 (no filter, top_p off): the Gumbel-max trick -- argmax of
//             score + Gumbel noise IS a draw from softmax(score) -- as per-slice maxima for the greedy merge kernel.
//   launch 2  sample_select_kernel    one block per sequence: the 128 x 64 candidates in LDS, the global k-th largest score by the
//             same radix select, kept = every candidate >= it (ties kept, like TopKLogitsWarper's `scores < kth` mask), rank-sorted
//             by (score desc, index desc: torch.sort's order among ties, reversed) -- deterministic whatever order the LDS atomics
//             appended them in; then, on one thread over
//             <= 256 entries: top-p exactly as TopPLogitsWarper (ascending cumulative softmax, remove <= 1 - top_p, never the
//             largest), and the draw by inverse CDF on one Philox4x32-10 uniform.
//   launch 3  advance_kernel (model.hip) books the token like the greedy path and advances the Philox counter.
// Randomness: Philox4x32-10 keyed by the caller's seed, counter = (step counter, sequence, vocabulary index | ~0): reproducible for
// a given seed, independent across steps / sequences / slices; torch's generator cannot be matched (HF itself draws differently
// on CPU and GPU) -- parity is the KEPT SET (bit-equal to HF's warpers on the same logits) and the drawn DISTRIBUTION (chi-square).
#include "common.h"

namespace {

constexpr int SMP_NB = 128;          // vocabulary slices per sequence
constexpr int SMP_K = 64;            // candidates per slice = the largest top_k served on the device
constexpr int SMP_SLICE_MAX = 2048;  // vocabulary entries of a slice (LDS): V <= 128 * 2048
constexpr int SMP_LIST = 256;        // kept-set capacity: top_k plus ties at the k-th score

__device__ __forceinline__ unsigned key_of(float s) {
  const unsigned u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // unsigned order == float order (-inf .. +inf)
}
__device__ __forceinline__ float score_of(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

struct U4 {
  unsigned x, y, z, w;
};
__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// k-th largest (k = 1 .. n) of keys[0 .. n) held in LDS; every thread of the block calls it (barriers inside).
// Returns the key; *need_eq = how many entries EQUAL to it belong to the top k, *count_eq = how many such entries exist.
struct SelectLds {
  unsigned hist[256];
  unsigned bin, above, cnt;
};
__device__ unsigned block_select_kth(const unsigned* __restrict__ keys, int n, int k, SelectLds& L, int* need_eq, int* count_eq) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  unsigned prefix = 0, mask = 0;
  int remaining = k;
  unsigned last_cnt = 0;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) L.hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
      const unsigned key = keys[i];
      if ((key & mask) == prefix) atomicAdd(&L.hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // one wave: suffix sums over the 256 bins, 4 bins per lane
      const unsigned c0 = L.hist[4 * lane], c1 = L.hist[4 * lane + 1], c2 = L.hist[4 * lane + 2], c3 = L.hist[4 * lane + 3];
      const unsigned tot = c0 + c1 + c2 + c3;
      unsigned v = tot;  // inclusive suffix over lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(v, o);
        if (lane + o < 64) v += t;
      }
      const unsigned e = v - tot;  // entries in bins above this lane's four
      const unsigned s3 = e + c3, s2 = s3 + c2, s1 = s2 + c1, s0 = s1 + c0;
      const unsigned need = (unsigned)remaining;
      // exactly one (lane, j): S[bin] >= need > S[bin + 1]
      if (s3 >= need && e < need) { L.bin = 4 * lane + 3; L.above = e; L.cnt = c3; }
      else if (s2 >= need && s3 < need) { L.bin = 4 * lane + 2; L.above = s3; L.cnt = c2; }
      else if (s1 >= need && s2 < need) { L.bin = 4 * lane + 1; L.above = s2; L.cnt = c1; }
      else if (s0 >= need && s1 < need) { L.bin = 4 * lane; L.above = s1; L.cnt = c0; }
    }
    __syncthreads();
    prefix |= L.bin << shift;
    mask |= 0xFFu << shift;
    remaining -= (int)L.above;
    last_cnt = L.cnt;
    __syncthreads();
  }
  *need_eq = remaining;
  *count_eq = (int)last_cnt;
  return prefix;
}

__global__ __launch_bounds__(256) void sample_partial_kernel(const float* __restrict__ logits, const srgpt_sampling* __restrict__ sp,
                                                             unsigned* __restrict__ cand_key, int* __restrict__ cand_idx,
                                                             float* __restrict__ pv, int* __restrict__ pi, int V) {
  __shared__ unsigned keys[SMP_SLICE_MAX];
  __shared__ SelectLds L;
  __shared__ unsigned cnt;
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.y, nb = gridDim.x, blk = blockIdx.x, tid = threadIdx.x;
  const int per = (V + nb - 1) / nb;
  const int lo = blk * per, hi = min(lo + per, V), n = max(hi - lo, 0);
  const float* row = logits + (size_t)b * V;
  const float T = sp->temperature;
  const int topk = min(sp->top_k, SMP_K);
  if (topk <= 0) {
    // ---- Gumbel-max: argmax_i (logit_i / T + g_i), g_i = -log(-log(u_i)), is a draw from softmax(logits / T) ----
    const unsigned long long ctr = sp->counter, seed = sp->seed;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lo + tid; i < hi; i += 256) {
      const U4 r = philox4x32_10(U4{(unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)b, (unsigned)i}, (unsigned)seed, (unsigned)(seed >> 32));
      const float u = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
      const float v = row[i] / T - logf(-logf(u));
      if (v > best) {
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
    if ((tid & 63) == 0) {
      sv[tid >> 6] = best;
      si[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < 4; ++i)
        if (sv[i] > best || (sv[i] == best && si[i] < bi)) {
          best = sv[i];
          bi = si[i];
        }
      pv[(size_t)b * nb + blk] = best;
      pi[(size_t)b * nb + blk] = bi;
    }
    return;
  }
  // ---- top-k of the slice ----
  const size_t base = ((size_t)b * nb + blk) * SMP_K;
  if (tid < SMP_K) {
    cand_key[base + tid] = 0u;  // below every real key
    cand_idx[base + tid] = -1;
  }
  if (tid == 0) cnt = 0;
  for (int i = tid; i < n; i += 256) keys[i] = key_of(row[lo + i] / T);
  __syncthreads();
  if (n == 0) return;  // uniform
  const int kk = min(topk, n);
  int need_eq, count_eq;
  const unsigned thr = block_select_kth(keys, n, kk, L, &need_eq, &count_eq);
  // everything above the slice's k-th score, and EVERY entry equal to it while the 64 slots last: TopKLogitsWarper keeps ties at the
  // global k-th score, so a slice must not drop its share of them (three equal maxima in one slice under top_k = 1 are three kept
  // tokens).  Only a slice with more than 64 - (entries above) equal scores truncates -- lowest indices first, a fixed set.
  const int room = SMP_K - (kk - need_eq);
  for (int i = tid; i < n; i += 256) {
    const unsigned key = keys[i];
    bool take = key > thr;
    if (key == thr) {
      if (count_eq <= room) take = true;
      else {
        int r = 0;
        for (int j = 0; j < i; ++j) r += keys[j] == thr;
        take = r < room;
      }
    }
    if (take) {
      const unsigned slot = atomicAdd(&cnt, 1u);
      if (slot < (unsigned)SMP_K) {
        cand_key[base + slot] = key;
        cand_idx[base + slot] = lo + i;
      }
    }
  }
}

__global__ __launch_bounds__(1024) void sample_select_kernel(const srgpt_sampling* __restrict__ sp, const unsigned* __restrict__ cand_key,
                                                             const int* __restrict__ cand_idx, int64_t* __restrict__ tok, int nb,
                                                             int* __restrict__ err) {
  constexpr int NC = SMP_NB * SMP_K;
  __shared__ unsigned keys[NC];
  __shared__ int idxs[NC];
  __shared__ SelectLds L;
  __shared__ unsigned lkey[SMP_LIST];
  __shared__ int lidx[SMP_LIST];
  __shared__ float ss[SMP_LIST];
  __shared__ int sidx[SMP_LIST];
  __shared__ unsigned n_valid, n_list;
  // settings the device sampler does not serve must not draw from a silently different distribution: top_k beyond the
  // candidate slots would be clamped, and a top-p filter without top-k would be ignored by the Gumbel path -- error bit 4, reported
  // by srgpt_llm_decode_sync_state (the state's word: sticky) / srgpt_sample_status (the workspace's word: the last srgpt_sample call)
  if (blockIdx.x == 0 && threadIdx.x == 0 && (sp->top_k > SMP_K || (sp->top_k <= 0 && sp->top_p < 1.f))) atomicOr(err, 4);
  const int topk = min(sp->top_k, SMP_K);
  if (topk <= 0) return;  // Gumbel-max mode: the greedy merge (advance_kernel) picks
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int n = nb * SMP_K;
  if (tid == 0) {
    n_valid = 0;
    n_list = 0;
  }
  __syncthreads();
  unsigned mine = 0;
  for (int i = tid; i < n; i += nt) {
    const int ix = cand_idx[(size_t)b * n + i];
    keys[i] = cand_key[(size_t)b * n + i];
    idxs[i] = ix;
    mine += ix >= 0;
  }
  if (mine) atomicAdd(&n_valid, mine);
  __syncthreads();
  const int kk = min(topk, (int)n_valid);
  if (kk <= 0) {  // no vocabulary at all: cannot happen for V >= 1
    if (tid == 0) tok[b] = 0;
    return;
  }
  int need_eq, count_eq;
  const unsigned thr = block_select_kth(keys, n, kk, L, &need_eq, &count_eq);
  for (int i = tid; i < n; i += nt)
    if (idxs[i] >= 0 && keys[i] >= thr) {
      const unsigned slot = atomicAdd(&n_list, 1u);
      if (slot < (unsigned)SMP_LIST) {
        lkey[slot] = keys[i];
        lidx[slot] = idxs[i];
      }
    }
  __syncthreads();
  if (n_list > (unsigned)SMP_LIST && tid == 0) atomicOr(err, 2);  // > 256 - k entries tie at the k-th score: reported, not silent
  const int nk = min((int)n_list, SMP_LIST);
  // rank by (score desc, index DESC): TopPLogitsWarper sorts ascending and cuts from the small end, never the last entry.  Where the
  // cut runs through a group of exactly equal scores, which members survive is torch.sort's tie order -- index order for short
  // rows (then the HIGHEST index survives longest: what this order reproduces), unspecified for long ones -- so only the kept
  // VALUES are defined there.  A total order: the list does not depend on the append order above.
  for (int e = tid; e < nk; e += nt) {
    const unsigned ke = lkey[e];
    const int ie = lidx[e];
    int r = 0;
    for (int j = 0; j < nk; ++j) r += (lkey[j] > ke) || (lkey[j] == ke && lidx[j] > ie);
    ss[r] = score_of(ke);
    sidx[r] = ie;
  }
  __syncthreads();
  if (tid == 0) {
    const float m = ss[0];
    int n2 = nk;
    const float top_p = sp->top_p;
    if (top_p < 1.0f) {
      // TopPLogitsWarper: sort ascending, softmax, cumulative sum; remove while cumsum <= 1 - top_p, never the last (largest)
      float Z = 0.f;
      for (int r = nk - 1; r >= 0; --r) Z += __expf(ss[r] - m);
      float c = 0.f;
      for (int r = nk - 1; r >= 1; --r) {
        c += __expf(ss[r] - m) / Z;
        if (c <= sp->top_p_rm) n2 = r;
        else break;
      }
    }
    float Z2 = 0.f;
    for (int r = 0; r < n2; ++r) Z2 += __expf(ss[r] - m);
    const unsigned long long ctr = sp->counter, seed = sp->seed;
    const U4 rr = philox4x32_10(U4{(unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)b, 0xFFFFFFFFu}, (unsigned)seed, (unsigned)(seed >> 32));
    const float u = (float)(rr.x >> 8) * (1.0f / 16777216.0f);  // [0, 1)
    const float target = u * Z2;
    int pick = n2 - 1;
    float c = 0.f;
    for (int r = 0; r < n2; ++r) {
      c += __expf(ss[r] - m);
      if (target < c) {
        pick = r;
        break;
      }
    }
    tok[b] = sidx[pick];
    int* ko = sp->kept_out;
    if (ko) {  // parity hook: the kept set of this step, best first
      ko += (size_t)b * (SMP_LIST + 1);
      ko[0] = n2;
      for (int r = 0; r < n2; ++r) ko[1 + r] = sidx[r];
    }
  }
}

__global__ void sample_bump_kernel(srgpt_sampling* sp, const float* __restrict__ pv, const int* __restrict__ pi, int nb,
                                   int64_t* __restrict__ tok, int B) {
  // stand-alone op only (the decode step's advance_kernel does both): Gumbel mode merges the slice maxima; every mode advances the counter
  if (sp->top_k <= 0)
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int i = 0; i < nb; ++i) {
        const float v = pv[(size_t)b * nb + i];
        const int ix = pi[(size_t)b * nb + i];
        if (v > best || (v == best && ix < bi)) {
          best = v;
          bi = ix;
        }
      }
      tok[b] = bi == 0x7fffffff ? 0 : bi;
    }
  __syncthreads();
  if (threadIdx.x == 0) sp->counter += 1;
}

}  // namespace

// workspace: candidates (key + index) of every slice, then the Gumbel-mode slice maxima
extern "C" int64_t srgpt_sample_ws_bytes(int B) {
  if (B <= 0) return -1;
  return (int64_t)B * SMP_NB * SMP_K * 8 + (int64_t)B * SMP_NB * 8 + 256;
}

// internal (model.hip): launches 1 and 2; the caller books the token (advance_kernel) and advances the counter
int srgpt_sample_launch(const float* logits, const srgpt_sampling* sp, int64_t* tok, void* ws, float* pv, int* pi, int* err, int B, int V,
                        hipStream_t s) {
  SRGPT_CHECK(logits && sp && tok && ws && pv && pi && err, SRGPT_ERR_ARG, "srgpt_sample: null pointer");
  SRGPT_CHECK(B > 0 && V > 0 && V <= SMP_NB * SMP_SLICE_MAX, SRGPT_ERR_UNSUPPORTED, "srgpt_sample: vocabulary %d exceeds %d", V,
              SMP_NB * SMP_SLICE_MAX);
  unsigned* ck = reinterpret_cast<unsigned*>(ws);
  int* ci = reinterpret_cast<int*>(ck + (size_t)B * SMP_NB * SMP_K);
  hipLaunchKernelGGL(sample_partial_kernel, dim3(SMP_NB, B), dim3(256), 0, s, logits, sp, ck, ci, pv, pi, V);
  hipLaunchKernelGGL(sample_select_kernel, dim3(B), dim3(1024), 0, s, sp, ck, ci, tok, SMP_NB, err);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_sample(const float* logits, srgpt_sampling* sp, int64_t* tok_out, void* ws, int B, int V, srgpt_stream_t stream) {
  SRGPT_CHECK(logits && sp && tok_out && ws && B > 0 && V > 0, SRGPT_ERR_ARG, "srgpt_sample: null pointer or empty shape");
  SRGPT_CHECK(V <= SMP_NB * SMP_SLICE_MAX, SRGPT_ERR_UNSUPPORTED, "srgpt_sample: vocabulary %d exceeds %d", V, SMP_NB * SMP_SLICE_MAX);
  char* tail = reinterpret_cast<char*>(ws) + (size_t)B * SMP_NB * SMP_K * 8;
  float* pv = reinterpret_cast<float*>(tail);
  int* pi = reinterpret_cast<int*>(tail + (size_t)B * SMP_NB * 4);
  int* err = reinterpret_cast<int*>(tail + (size_t)B * SMP_NB * 8);
  hipStream_t s = as_stream(stream);
  SRGPT_HIP_TRY(hipMemsetAsync(err, 0, sizeof(int), s), "srgpt_sample: clearing the error word");
  SRGPT_TRY(srgpt_sample_launch(logits, sp, tok_out, ws, pv, pi, err, B, V, s));
  hipLaunchKernelGGL(sample_bump_kernel, dim3(1), dim3(256), 0, s, sp, pv, pi, SMP_NB, tok_out, B);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

// health of the last srgpt_sample call(s) on this workspace (synchronises `stream`): the sticky error word at the tail of ws
extern "C" int srgpt_sample_status(const void* ws, int B, srgpt_stream_t stream) {
  SRGPT_CHECK(ws && B > 0, SRGPT_ERR_ARG, "srgpt_sample_status: bad args");
  const char* tail = reinterpret_cast<const char*>(ws) + (size_t)B * SMP_NB * SMP_K * 8;
  int host = 0;
  SRGPT_HIP_TRY(hipMemcpyAsync(&host, tail + (size_t)B * SMP_NB * 8, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)), "srgpt_sample_status: copy");
  SRGPT_HIP_TRY(hipStreamSynchronize(as_stream(stream)), "srgpt_sample_status: synchronize");
  SRGPT_CHECK((host & 4) == 0, SRGPT_ERR_UNSUPPORTED,
              "sampling: top_k > %d, or a top-p filter without top-k, is not served by the device sampler (the draw used another distribution)", SMP_K);
  SRGPT_CHECK((host & 2) == 0, SRGPT_ERR_STATE, "sampling: more than %d vocabulary entries tie at the top-k threshold (kept set truncated)", SMP_LIST);
  return SRGPT_OK;
}

extern "C" __attribute__((visibility("hidden"))) int srgpt_sample_slices(void) { return SMP_NB; }  // cross-file helper, not exported
