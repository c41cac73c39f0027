// Token-stream splice on the device (reference: the per-token Python loop of LlavaMetaForCausalLM.
// prepare_inputs_labels_for_multimodal, llava_arch.py:420-611).
//
// The reference walks every prompt with `.item()`-style host reads: it drops the positions the attention mask hides, replaces each
// <image> sentinel (-200) by the projector's rows of the next image, replaces the <mask> / <depth> ids of a prompt that owns images
// by the region embeddings of the prompt's FIRST image in order of appearance (llava_arch.py:470-505), embeds everything else,
// truncates to tokenizer_model_max_length and pads the batch on `padding_side`.  Rounds 1-4 kept that index arithmetic as a host
// loop over per-row tuples -- O(batch x length) interpreter work inside every request, behind a device->host copy of the ids.
//
// Here it is two launches and one small read-back:
//   srgpt_splice_plan    one block per prompt: four block-wide exclusive scans over the prompt's ids (kept positions' output width,
//                        image ordinal, <mask> ordinal, <depth> ordinal) and a row -> source descriptor table [B][Tcap]; the image index
//                        a prompt starts at is the number of sentinels in the prompts before it (re-counted per block: B x P ids, tiny);
//                        per-prompt facts the host must judge (lengths, counts, id range) go to `stats` -- the ONLY thing read back,
//                        and it is read before the vision tower is launched, so the host never waits behind the GPU.
//   srgpt_splice_gather  one block per output row: descriptor -> source row (embedding table / image features / mask embeddings /
//                        depth embeddings / zeros for padding), 16-byte copies; also writes the spliced labels and attention mask.
// Integer and byte work only: results are identical to the reference loop (tests/test_gpu_splice.py checks against the host loop).
#include "common.h"

namespace {

constexpr int SPL_TEXT = 0, SPL_IMAGE = 1, SPL_MASK = 2, SPL_DEPTH = 3;
constexpr int SPL_NT = 1024;
constexpr int64_t SPL_IMAGE_TOKEN = -200;  // IMAGE_TOKEN_INDEX (llava/constants.py)

// exclusive scan of 4 counters over the block (values per thread -> offsets), totals returned through `tot`
__device__ __forceinline__ void block_scan4(int (&v)[4], int (&tot)[4], int* lds /* [4][SPL_NT / 64 + 1] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NWV = SPL_NT / 64;
  int incl[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    int x = v[c];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    incl[c] = x;
    if (lane == 63) lds[c * (NWV + 1) + wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    int run = 0;
    for (int w = 0; w < NWV; ++w) {
      const int t = lds[threadIdx.x * (NWV + 1) + w];
      lds[threadIdx.x * (NWV + 1) + w] = run;
      run += t;
    }
    lds[threadIdx.x * (NWV + 1) + NWV] = run;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v[c] = incl[c] - v[c] + lds[c * (NWV + 1) + wave];
    tot[c] = lds[c * (NWV + 1) + NWV];
  }
  __syncthreads();
}

// stats row of a prompt (SRGPT_SPLICE_STATS ints)
enum { ST_LEN = 0, ST_LEN_RAW, ST_NIMG, ST_NMASK, ST_NDEPTH, ST_FIRST_IMG, ST_MIN_ID, ST_MAX_ID };

__global__ __launch_bounds__(SPL_NT) void splice_plan_kernel(const int64_t* __restrict__ ids, const unsigned char* __restrict__ am, int B,
                                                             int P, int nimg_feat, int n_images_total, const int* __restrict__ img_info,
                                                             int use_masks, int use_depths, int64_t mask_id, int64_t depth_id, int max_len,
                                                             int Tcap, int* __restrict__ desc, int* __restrict__ stats,
                                                             int* __restrict__ scratch) {
  __shared__ int lds[4 * (SPL_NT / 64 + 1)];
  __shared__ int red[SPL_NT / 64];
  __shared__ int s_first;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* row = ids + (size_t)b * P;
  const unsigned char* arow = am ? am + (size_t)b * P : nullptr;

  // ---- image index this prompt starts at: sentinels of the prompts before it (llava_arch.py: cur_image_idx runs over the batch)
  {
    int cnt = 0;
    const size_t n = (size_t)b * P;
    for (size_t i = tid; i < n; i += SPL_NT) cnt += (ids[i] == SPL_IMAGE_TOKEN && (!am || am[i])) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < SPL_NT / 64; ++w) t += red[w];
      s_first = t;
    }
    __syncthreads();
  }
  const int first_img = s_first;

  // ---- pass 1: per-thread contiguous chunk of positions -> local counters
  const int per = (P + SPL_NT - 1) / SPL_NT;
  const int p0 = min(tid * per, P), p1 = min(p0 + per, P);
  int v[4] = {0, 0, 0, 0};  // output width, images, <mask> ids, <depth> ids
  long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
  for (int p = p0; p < p1; ++p) {
    if (arow && !arow[p]) continue;
    const int64_t t = row[p];
    const bool img = t == SPL_IMAGE_TOKEN;
    v[0] += img ? nimg_feat : 1;
    v[1] += img ? 1 : 0;
    v[2] += t == mask_id ? 1 : 0;
    v[3] += t == depth_id ? 1 : 0;
  }
  int tot[4];
  block_scan4(v, tot, lds);
  const int len_raw = tot[0], n_img = tot[1];
  const bool has_img = n_img > 0;
  const bool img_ok = has_img && first_img < n_images_total;
  const int me_cnt = img_ok ? img_info[2 * first_img] : -1;  // -1: this image has no region embeddings (masks[i] is None)
  const int me_off = img_ok ? img_info[2 * first_img + 1] : 0;
  const bool me_on = has_img && use_masks && me_cnt >= 0;
  const bool de_on = has_img && use_depths && me_cnt >= 0;
  const int len = max_len > 0 ? min(len_raw, max_len) : len_raw;

  // ---- pass 2: each kept position writes its row descriptor(s); image positions only record where their rows start (scratch),
  // the rows themselves are filled by the whole block below
  int off = v[0], io = v[1], mo = v[2], dd = v[3];
  int* d = desc + (size_t)b * Tcap * 2;
  int* img_start = scratch + (size_t)b * (n_images_total + 1);  // output row at which the prompt's i-th image begins
  for (int p = p0; p < p1; ++p) {
    if (arow && !arow[p]) continue;
    const int64_t t = row[p];
    if (t == SPL_IMAGE_TOKEN) {
      if (io < n_images_total) img_start[io] = off;
      off += nimg_feat;
      io += 1;
      continue;
    }
    int kind = SPL_TEXT, src = (int)t;
    if (me_on && t == mask_id) {
      kind = SPL_MASK;
      src = me_off + mo;
    } else if (de_on && t == depth_id) {
      kind = SPL_DEPTH;
      src = me_off + dd;
    } else {
      lo = t < lo ? t : lo;
      hi = t > hi ? t : hi;
    }
    mo += t == mask_id ? 1 : 0;
    dd += t == depth_id ? 1 : 0;
    if (off < len && off < Tcap) {
      d[2 * off] = kind | (src << 2);
      d[2 * off + 1] = p;
    }
    off += 1;
  }
  __syncthreads();
  const int n_fill = min(n_img, n_images_total);
  for (int i = 0; i < n_fill; ++i) {
    const int r0 = img_start[i];
    for (int r = tid; r < nimg_feat; r += SPL_NT) {
      const int o = r0 + r;
      if (o < len && o < Tcap) {
        d[2 * o] = SPL_IMAGE | (((first_img + i) * nimg_feat + r) << 2);
        d[2 * o + 1] = -1;
      }
    }
  }
  // ---- id range of the rows that go through the embedding table (nn.Embedding raises IndexError outside [0, vocab))
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  __shared__ long long rlo[SPL_NT / 64], rhi[SPL_NT / 64];
  if ((tid & 63) == 0) {
    rlo[tid >> 6] = lo;
    rhi[tid >> 6] = hi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < SPL_NT / 64; ++w) {
      lo = rlo[w] < lo ? rlo[w] : lo;
      hi = rhi[w] > hi ? rhi[w] : hi;
    }
    int* s = stats + (size_t)b * SRGPT_SPLICE_STATS;
    s[ST_LEN] = len;
    s[ST_LEN_RAW] = len_raw;
    s[ST_NIMG] = n_img;
    s[ST_NMASK] = tot[2];
    s[ST_NDEPTH] = tot[3];
    s[ST_FIRST_IMG] = first_img;
    // clamp to int32 (ids are int64; anything outside int32 is out of every vocabulary anyway)
    s[ST_MIN_ID] = lo > hi ? 0 : (int)(lo < -0x7fffffffLL ? -0x7fffffffLL : (lo > 0x7fffffffLL ? 0x7fffffffLL : lo));
    s[ST_MAX_ID] = lo > hi ? 0 : (int)(hi < -0x7fffffffLL ? -0x7fffffffLL : (hi > 0x7fffffffLL ? 0x7fffffffLL : hi));
  }
}

template <typename E>
__global__ __launch_bounds__(256) void splice_gather_kernel(const int* __restrict__ desc, const int* __restrict__ stats, int Tcap, int T,
                                                            int left_pad, int cols, const E* __restrict__ embed,
                                                            const E* __restrict__ image_features, const E* __restrict__ mask_embeds,
                                                            const E* __restrict__ depth_embeds, const int64_t* __restrict__ labels, int P,
                                                            int64_t ignore_index, E* __restrict__ out, int64_t* __restrict__ labels_out,
                                                            unsigned char* __restrict__ am_out) {
  constexpr int VEC = Vec16<E>::N;
  const int b = blockIdx.x / T, jp = blockIdx.x - b * T;
  const int len = stats[(size_t)b * SRGPT_SPLICE_STATS + ST_LEN];
  const int j = jp - (left_pad ? T - len : 0);
  E* dst = out + (size_t)blockIdx.x * cols;
  const bool valid = j >= 0 && j < len;
  if (!valid) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int c = threadIdx.x; c < cols / VEC; c += blockDim.x) *reinterpret_cast<u32x4*>(dst + c * VEC) = z;
    if (threadIdx.x == 0) {
      if (labels_out) labels_out[blockIdx.x] = ignore_index;
      if (am_out) am_out[blockIdx.x] = 0;
    }
    return;
  }
  const int* d = desc + ((size_t)b * Tcap + j) * 2;
  const int code = d[0], srcpos = d[1];
  const int kind = code & 3, src = code >> 2;
  const E* table = kind == SPL_TEXT ? embed : kind == SPL_IMAGE ? image_features : kind == SPL_MASK ? mask_embeds : depth_embeds;
  const E* s = table + (size_t)src * cols;
  for (int c = threadIdx.x; c < cols / VEC; c += blockDim.x)
    *reinterpret_cast<u32x4*>(dst + c * VEC) = *reinterpret_cast<const u32x4*>(s + c * VEC);
  if (threadIdx.x == 0) {
    if (labels_out) labels_out[blockIdx.x] = (kind == SPL_IMAGE || !labels) ? ignore_index : labels[(size_t)b * P + srcpos];
    if (am_out) am_out[blockIdx.x] = 1;
  }
}

}  // namespace

extern "C" int64_t srgpt_splice_scratch_ints(int B, int n_images_total) { return (int64_t)B * (n_images_total + 1); }

extern "C" int srgpt_splice_plan(const int64_t* ids, const unsigned char* attn_mask, int B, int P, int nimg_feat, int n_images_total,
                                 const int* img_info, int use_masks, int use_depths, int64_t mask_id, int64_t depth_id, int max_len, int Tcap,
                                 int* desc, int* stats, int* scratch, srgpt_stream_t stream) {
  SRGPT_CHECK(ids && desc && stats && scratch, SRGPT_ERR_ARG, "srgpt_splice_plan: null pointer");
  SRGPT_CHECK(B > 0 && P > 0 && nimg_feat >= 0 && n_images_total >= 0 && Tcap > 0, SRGPT_ERR_ARG, "srgpt_splice_plan: bad shape");
  SRGPT_CHECK(n_images_total == 0 || img_info, SRGPT_ERR_ARG, "srgpt_splice_plan: null pointer (img_info)");
  SRGPT_CHECK((long long)n_images_total * nimg_feat < (1LL << 29) && (long long)Tcap < (1LL << 29), SRGPT_ERR_UNSUPPORTED,
              "srgpt_splice_plan: source rows do not fit the descriptor");
  hipLaunchKernelGGL(splice_plan_kernel, dim3(B), dim3(SPL_NT), 0, as_stream(stream), ids, attn_mask, B, P, nimg_feat, n_images_total,
                     img_info, use_masks, use_depths, mask_id, depth_id, max_len, Tcap, desc, stats, scratch);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_splice_gather(const int* desc, const int* stats, int B, int Tcap, int T, int left_pad, int cols, int dtype,
                                   const void* embed, const void* image_features, const void* mask_embeds, const void* depth_embeds,
                                   const int64_t* labels, int P, int64_t ignore_index, void* out, int64_t* labels_out,
                                   unsigned char* attn_mask_out, srgpt_stream_t stream) {
  SRGPT_CHECK(desc && stats && embed && out, SRGPT_ERR_ARG, "srgpt_splice_gather: null pointer");
  SRGPT_CHECK(B > 0 && T > 0 && T <= Tcap && cols > 0, SRGPT_ERR_ARG, "srgpt_splice_gather: bad shape");
  SRGPT_CHECK(dtype == SRGPT_F32 || dtype == SRGPT_BF16, SRGPT_ERR_ARG, "srgpt_splice_gather: bad dtype %d", dtype);
  SRGPT_CHECK(cols % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_splice_gather: cols not 16-byte multiple");
  SRGPT_CHECK((long long)B * T < (1LL << 31), SRGPT_ERR_UNSUPPORTED, "srgpt_splice_gather: too many rows");
#define L(T_)                                                                                                                          \
  hipLaunchKernelGGL(splice_gather_kernel<T_>, dim3(B * T), dim3(256), 0, as_stream(stream), desc, stats, Tcap, T, left_pad, cols,     \
                     (const T_*)embed, (const T_*)image_features, (const T_*)mask_embeds, (const T_*)depth_embeds, labels, P,          \
                     ignore_index, (T_*)out, labels_out, attn_mask_out)
  if (dtype == SRGPT_BF16) L(bf16_t);
  else L(float);
#undef L
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
