// Region extractor kernels (SpatialRGPT specific, reference: llava/model/region_extractor/base_extractor.py)
// plus the layout kernels either side of it (projector space-to-depth, adaptive average pool, patch im2col).
//
// Masked region pooling is HBM-bound: the refined feature map (108*108 x C bf16 = 26.9 MB at C = 1152) is
// read exactly once for ALL M masks.  Three deterministic stages (no float atomics):
//   1. region_weights_kernel : one block per mask; bilinear resample to the feature grid (PyTorch's
//      upsample_bilinear2d index math, align_corners = False, no antialias), round to the feature dtype,
//      fixed-order sum, L1 normalise  -> w[M][L] fp32 holding dtype-representable values.
//   2. region_partial_kernel : grid (row slabs x channel slabs); each thread streams 16-byte channel
//      chunks of its rows and keeps M x 8 fp32 accumulators; in-block fixed-order reduce.
//   3. region_final_kernel   : fixed-order sum over slabs, round to dtype.
#include "common.h"

namespace {

constexpr int RP_ROWS = 128;   // feature rows per block
constexpr int RP_CHUNKS = 48;  // 16-byte channel chunks per block
constexpr int RP_RG = 5;       // row groups per block (48 * 5 = 240 of 256 threads)
constexpr int RP_MAXM = 16;    // masks per launch

struct Idx {
  int i0, i1;
  float l0, l1;
};
// PyTorch area_pixel_compute_source_index + guard_index_and_lambda (ATen/native/UpSample.h)
__device__ __forceinline__ Idx src_index(float rscale, int dst, int n_in) {
  float src = rscale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Idx r;
  r.i0 = min((int)src, n_in - 1);
  r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
  r.l0 = 1.f - r.l1;
  r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
  return r;
}

// RAW = true: `masks` are the caller's raw uint8 [M, rh, rw] masks and (ys, xs) the cv2.INTER_NEAREST source
// index of every row / column of the mh x mw processor-size mask the reference would have built on the host
// (mm_utils.py:477-532: cv2.resize(..., INTER_NEAREST), then the processor with rescale 1.0 -> float(uint8)); the bilinear taps
// read THROUGH the tables, so the nearest resize, the float conversion and the resample are one pass over the raw bytes.
template <typename T, typename MT, bool RAW>
__global__ __launch_bounds__(1024) void region_weights_kernel(const MT* __restrict__ masks, float* __restrict__ w,
                                                              int mh, int mw, int fw, float rscale_h, float rscale_w,
                                                              const int* __restrict__ ys, const int* __restrict__ xs, int rh,
                                                              int rw) {
  __shared__ float red[16];
  const int m = blockIdx.x;
  const MT* mk = masks + (size_t)m * (RAW ? (size_t)rh * rw : (size_t)mh * mw);
  float* wm = w + (size_t)m * fw * fw;
  const int L = fw * fw;
  float s = 0.f;
  auto at = [&](int yy, int xx) -> float {
    if constexpr (RAW)
      return (float)mk[(size_t)ys[yy] * rw + xs[xx]];  // float(uint8), as the processor with rescale_factor 1.0 produces
    else
      return to_f(mk[(size_t)yy * mw + xx]);
  };
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    const int oy = l / fw, ox = l - oy * fw;
    const Idx y = src_index(rscale_h, oy, mh), x = src_index(rscale_w, ox, mw);
    const float v00 = at(y.i0, x.i0), v01 = at(y.i0, x.i1);
    const float v10 = at(y.i1, x.i0), v11 = at(y.i1, x.i1);
    // explicit fused steps: every instantiation (float / bf16 / raw uint8 masks) rounds the same way, whatever the
    // compiler's contraction choices would have been
    const float top = fmaf(x.l1, v01, x.l0 * v00), bot = fmaf(x.l1, v11, x.l0 * v10);
    const float v = rnd<T>(fmaf(y.l1, bot, y.l0 * top));  // .to(x.dtype)
    wm[l] = v;
    s += v;
  }
  const float denorm = rnd<T>(rnd<T>(block_sum(s, red)) + 1e-8f);  // mask.sum() + 1e-8 in the feature dtype
  for (int l = threadIdx.x; l < L; l += blockDim.x) wm[l] = rnd<T>(wm[l] / denorm);
}

template <typename T, int MM>
__global__ __launch_bounds__(256) void region_partial_kernel(const T* __restrict__ feat, const float* __restrict__ w,
                                                             float* __restrict__ partial, int M, int L, int C) {
  constexpr int VEC = Vec16<T>::N;
  __shared__ float ws[MM][RP_ROWS];
  __shared__ float red[RP_RG][RP_CHUNKS * 8];
  const int slab = blockIdx.x, l0 = slab * RP_ROWS;
  const int nrows = min(RP_ROWS, L - l0);
  const int tid = threadIdx.x;
  for (int i = tid; i < MM * RP_ROWS; i += 256) {
    const int m = i / RP_ROWS, r = i - m * RP_ROWS;
    ws[m][r] = (m < M && r < nrows) ? w[(size_t)m * L + l0 + r] : 0.f;
  }
  __syncthreads();
  const int cl = tid % RP_CHUNKS, rg = tid / RP_CHUNKS;
  const int chunk = blockIdx.y * RP_CHUNKS + cl;
  const bool active = rg < RP_RG && chunk * VEC < C;
  float acc[MM][VEC];
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[m][i] = 0.f;
  if (active) {
    for (int r = rg; r < nrows; r += RP_RG) {
      const Vec16<T> f = *reinterpret_cast<const Vec16<T>*>(feat + (size_t)(l0 + r) * C + (size_t)chunk * VEC);
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const float wv = ws[m][r];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[m][i] = fmaf(wv, f.get(i), acc[m][i]);
      }
    }
  }
  // fixed-order reduce over the row groups, one mask at a time
  for (int m = 0; m < MM; ++m) {
    __syncthreads();
    if (rg < RP_RG)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[rg][cl * VEC + i] = acc[m][i];
    __syncthreads();
    if (m < M)
      for (int j = tid; j < RP_CHUNKS * VEC; j += 256) {
        const int c = blockIdx.y * RP_CHUNKS * VEC + j;
        if (c < C) {
          float t = 0.f;
#pragma unroll
          for (int g = 0; g < RP_RG; ++g) t += red[g][j];
          partial[((size_t)slab * M + m) * C + c] = t;
        }
      }
  }
}

template <typename T>
__global__ void region_final_kernel(const float* __restrict__ partial, T* __restrict__ out, int nslab, int MC) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= MC) return;
  float t = 0.f;
  for (int s = 0; s < nslab; ++s) t += partial[(size_t)s * MC + i];
  out[i] = from_f<T>(t);
}

// AdaptiveAvgPool2d(out_w) on a channels-last [n, in_w, in_w, C] map -> [n, out_w*out_w, C]
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int in_w, int out_w, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int op = blockIdx.x, img = blockIdx.y;
  const int oy = op / out_w, ox = op - oy * out_w;
  const int y0 = (oy * in_w) / out_w, y1 = ((oy + 1) * in_w + out_w - 1) / out_w;
  const int x0 = (ox * in_w) / out_w, x1 = ((ox + 1) * in_w + out_w - 1) / out_w;
  const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
  const T* xi = x + (size_t)img * in_w * in_w * C;
  for (int c = threadIdx.x; c < C / VEC; c += blockDim.x) {
    float s[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(xi + ((size_t)yy * in_w + xx) * C + (size_t)c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] += v.get(i);
      }
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, s[i] * inv);
    *reinterpret_cast<Vec16<T>*>(y + ((size_t)img * out_w * out_w + op) * C + (size_t)c * VEC) = o;
  }
}

// DownSampleBlock: out token o = bc * hb + br, channels [x(2br,2bc) | x(2br,2bc+1) | x(2br+1,2bc) | x(2br+1,2bc+1)],
// x(r,c) = input token r*g + c, zero outside the g x g grid (verified against the reference).
template <typename T>
__global__ void s2d_kernel(const T* __restrict__ x, T* __restrict__ y, int g, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int hb = (g + 1) >> 1;
  const int o = blockIdx.x, img = blockIdx.y;
  const int br = o % hb, bc = o / hb;
  const int cpq = C / VEC;  // chunks per quadrant
  for (int c = threadIdx.x; c < 4 * cpq; c += blockDim.x) {
    const int quad = c / cpq, cc = c - quad * cpq;
    const int r = 2 * br + (quad >> 1), col = 2 * bc + (quad & 1);
    u32x4 v = {0, 0, 0, 0};
    if (r < g && col < g)
      v = *reinterpret_cast<const u32x4*>(x + ((size_t)img * g * g + (size_t)r * g + col) * C + (size_t)cc * VEC);
    *reinterpret_cast<u32x4*>(y + ((size_t)img * hb * hb + o) * 4 * C + (size_t)c * VEC) = v;
  }
}

// patch im2col for Conv2d(3, C, k = s = p, 'valid'): row (img, py, px) <- pixels ordered (c, ky, kx), zero pad to kp
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ img, T* __restrict__ out, int S, int p, int kp) {
  const int gw = S / p;
  const int patch = blockIdx.x, n = blockIdx.y;
  const int py = patch / gw, px = patch - py * gw;
  T* o = out + ((size_t)n * gw * gw + patch) * kp;
  const int kk = 3 * p * p;
  for (int i = threadIdx.x; i < kp; i += blockDim.x) {
    T v = from_f<T>(0.f);
    if (i < kk) {
      const int c = i / (p * p), rem = i - c * p * p, ky = rem / p, kx = rem - ky * p;
      v = img[(((size_t)n * 3 + c) * S + (py * p + ky)) * S + (px * p + kx)];
    }
    o[i] = v;
  }
}

// CLIP embeddings: [cls | patches] + position embedding, one block per output token
template <typename T>
__global__ void vit_assemble_cls_kernel(const T* __restrict__ patches, const T* __restrict__ cls, const T* __restrict__ pos,
                                        T* __restrict__ x, int gg, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int t = blockIdx.x, img = blockIdx.y;
  const T* src = (t == 0) ? cls : patches + ((size_t)img * gg + (t - 1)) * C;
  const T* pe = pos + (size_t)t * C;
  T* dst = x + ((size_t)img * (gg + 1) + t) * C;
  for (int c = threadIdx.x; c < C / VEC; c += blockDim.x) {
    const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(src + c * VEC);
    const Vec16<T> b = *reinterpret_cast<const Vec16<T>*>(pe + c * VEC);
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, a.get(i) + b.get(i));
    *reinterpret_cast<Vec16<T>*>(dst + c * VEC) = o;
  }
}

}  // namespace

extern "C" int64_t srgpt_region_pool_ws_floats(int M, int fw, int C) {
  const int64_t L = (int64_t)fw * fw;
  const int64_t nslab = (L + RP_ROWS - 1) / RP_ROWS;
  return (int64_t)M * L + nslab * M * C;
}

static int region_pool_impl(const void* feat, const void* masks, void* out, float* ws, int M, int mh, int mw, int fw, int C,
                            float rscale_h, float rscale_w, int mask_dtype, int dtype, const int* ys, const int* xs, int rh, int rw,
                            srgpt_stream_t stream) {
  const bool raw = ys != nullptr;
  SRGPT_CHECK(raw || mask_dtype == SRGPT_BF16 || mask_dtype == SRGPT_F32, SRGPT_ERR_ARG, "srgpt_region_pool: bad mask dtype");
  SRGPT_CHECK(feat && masks && out && ws, SRGPT_ERR_ARG, "srgpt_region_pool: null pointer");
  SRGPT_CHECK(M > 0 && M <= RP_MAXM, SRGPT_ERR_ARG, "srgpt_region_pool: M=%d must be in 1..%d per call", M, RP_MAXM);
  SRGPT_CHECK(mh > 0 && mw > 0 && fw > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_region_pool: bad shape");
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(C % vec == 0, SRGPT_ERR_ARG, "srgpt_region_pool: C=%d must be a multiple of %d", C, vec);
  SRGPT_CHECK(dtype == SRGPT_BF16 || dtype == SRGPT_F32, SRGPT_ERR_ARG, "srgpt_region_pool: bad dtype");
  hipStream_t s = as_stream(stream);
  const int L = fw * fw, nslab = cdiv(L, RP_ROWS);
  float* w = ws;
  float* partial = ws + (size_t)M * L;
  dim3 pgrid(nslab, cdiv(C / vec, RP_CHUNKS));
#define RW(TT, MT, RAWV)                                                                                                   \
  hipLaunchKernelGGL((region_weights_kernel<TT, MT, RAWV>), dim3(M), dim3(1024), 0, s, (const MT*)masks, w, mh, mw, fw,    \
                     rscale_h, rscale_w, ys, xs, rh, rw)
  if (dtype == SRGPT_BF16) {
    if (raw) RW(bf16_t, unsigned char, true);
    else if (mask_dtype == SRGPT_BF16) RW(bf16_t, bf16_t, false);
    else RW(bf16_t, float, false);
    if (M <= 8)
      hipLaunchKernelGGL((region_partial_kernel<bf16_t, 8>), pgrid, dim3(256), 0, s, (const bf16_t*)feat, w, partial, M, L, C);
    else
      hipLaunchKernelGGL((region_partial_kernel<bf16_t, 16>), pgrid, dim3(256), 0, s, (const bf16_t*)feat, w, partial, M, L, C);
    hipLaunchKernelGGL(region_final_kernel<bf16_t>, dim3(cdiv(M * C, 256)), dim3(256), 0, s, partial, (bf16_t*)out, nslab, M * C);
  } else {
    if (raw) RW(float, unsigned char, true);
    else if (mask_dtype == SRGPT_BF16) RW(float, bf16_t, false);
    else RW(float, float, false);
    if (M <= 8)
      hipLaunchKernelGGL((region_partial_kernel<float, 8>), pgrid, dim3(256), 0, s, (const float*)feat, w, partial, M, L, C);
    else
      hipLaunchKernelGGL((region_partial_kernel<float, 16>), pgrid, dim3(256), 0, s, (const float*)feat, w, partial, M, L, C);
    hipLaunchKernelGGL(region_final_kernel<float>, dim3(cdiv(M * C, 256)), dim3(256), 0, s, partial, (float*)out, nslab, M * C);
  }
#undef RW
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_region_pool(const void* feat, const void* masks, void* out, float* ws, int M, int mh, int mw,
                                 int fw, int C, float rscale_h, float rscale_w, int mask_dtype, int dtype,
                                 srgpt_stream_t stream) {
  return region_pool_impl(feat, masks, out, ws, M, mh, mw, fw, C, rscale_h, rscale_w, mask_dtype, dtype, nullptr, nullptr, 0, 0,
                          stream);
}

extern "C" int srgpt_region_pool_u8(const void* feat, const void* masks_u8, const int* ys, const int* xs, void* out, float* ws,
                                    int M, int rh, int rw, int mh, int mw, int fw, int C, float rscale_h, float rscale_w,
                                    int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(ys && xs && rh > 0 && rw > 0, SRGPT_ERR_ARG, "srgpt_region_pool_u8: null index tables / bad raw size");
  return region_pool_impl(feat, masks_u8, out, ws, M, mh, mw, fw, C, rscale_h, rscale_w, SRGPT_F32, dtype, ys, xs, rh, rw, stream);
}

#define RDISPATCH(dtype, ...)                      \
  if ((dtype) == SRGPT_BF16) {                     \
    using T = bf16_t;                              \
    __VA_ARGS__;                                   \
  } else if ((dtype) == SRGPT_F32) {               \
    using T = float;                               \
    __VA_ARGS__;                                   \
  } else {                                         \
    srgpt_set_error("bad dtype %d", (int)(dtype)); \
    return SRGPT_ERR_ARG;                          \
  }

extern "C" int srgpt_avgpool(const void* x, void* y, int n_img, int in_w, int out_w, int C, int dtype,
                             srgpt_stream_t stream) {
  SRGPT_CHECK(x && y && n_img > 0 && in_w > 0 && out_w > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_avgpool: bad args");
  SRGPT_CHECK(C % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_avgpool: C not a 16-byte multiple");
  RDISPATCH(dtype, hipLaunchKernelGGL(avgpool_kernel<T>, dim3(out_w * out_w, n_img), dim3(256), 0, as_stream(stream),
                                      (const T*)x, (T*)y, in_w, out_w, C));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_s2d(const void* x, void* y, int n_img, int g, int C, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(x && y && n_img > 0 && g > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_s2d: bad args");
  SRGPT_CHECK(C % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_s2d: C not a 16-byte multiple");
  const int hb = (g + 1) / 2;
  RDISPATCH(dtype, hipLaunchKernelGGL(s2d_kernel<T>, dim3(hb * hb, n_img), dim3(256), 0, as_stream(stream), (const T*)x,
                                      (T*)y, g, C));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_im2col(const void* images, void* out, int n_img, int S, int patch, int kp, int dtype,
                            srgpt_stream_t stream) {
  SRGPT_CHECK(images && out && n_img > 0 && S > 0 && patch > 0, SRGPT_ERR_ARG, "srgpt_im2col: bad args");
  SRGPT_CHECK(kp >= 3 * patch * patch, SRGPT_ERR_ARG, "srgpt_im2col: kp < 3*patch*patch");
  const int gw = S / patch;
  RDISPATCH(dtype, hipLaunchKernelGGL(im2col_kernel<T>, dim3(gw * gw, n_img), dim3(256), 0, as_stream(stream),
                                      (const T*)images, (T*)out, S, patch, kp));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_vit_assemble_cls(const void* patches, const void* cls_emb, const void* pos_emb, void* x, int n_img,
                                      int gg, int C, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(patches && cls_emb && pos_emb && x && n_img > 0 && gg > 0 && C > 0, SRGPT_ERR_ARG, "srgpt_vit_assemble_cls: bad args");
  SRGPT_CHECK(C % (dtype == SRGPT_BF16 ? 8 : 4) == 0, SRGPT_ERR_ARG, "srgpt_vit_assemble_cls: C not a 16-byte multiple");
  RDISPATCH(dtype, hipLaunchKernelGGL(vit_assemble_cls_kernel<T>, dim3(gg + 1, n_img), dim3(128), 0, as_stream(stream),
                                      (const T*)patches, (const T*)cls_emb, (const T*)pos_emb, (T*)x, gg, C));
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
