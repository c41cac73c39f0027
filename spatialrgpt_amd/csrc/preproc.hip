// Request preprocessing on the device: what llava/mm_utils.py:421-532 does on the host with PIL / cv2 / the HF
// image processor, for raw uint8 inputs already uploaded:
//   * RGB / depth image: Pillow's two-pass bicubic resize on uint8 (ImagingResample, 8bpc fixed point: 22-bit
//     coefficients, round-half-up, clip to uint8 after EACH pass) -> rescale -> normalise -> dtype, HWC -> CHW.
//     The coefficient tables are computed on the host in double exactly like Pillow's precompute_coeffs (mm_utils.py);
//     the kernels are the integer convolution, so the result is bit-identical to PIL + the HF processor's float32 math.
//   * region masks: cv2.INTER_NEAREST gather (index tables from the host, computed in double like cv2) -> dtype.
// HBM-bound byte work: one read of the source, one write of the result; no MFMA reshaping.
#include "common.h"

namespace {

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass: src [H, W, C] -> dst [H, Wout, C]
__global__ void resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const int* __restrict__ bounds,
                                const int* __restrict__ coef, int ksize, int H, int W, int Wout, int C) {
  const int64_t total = (int64_t)H * Wout * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int xx = (int)((i / C) % Wout);
    const int y = (int)(i / ((int64_t)C * Wout));
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const unsigned char* p = src + ((int64_t)y * W + x0) * C + c;
    const int* k = coef + (int64_t)xx * ksize;
    int ss = 1 << 21;
    for (int x = 0; x < n; ++x) ss += (int)p[(int64_t)x * C] * k[x];
    dst[i] = (unsigned char)clip8(ss >> 22);
  }
}

// vertical pass + rescale/normalise: src [H, Wout, C] -> out [C, Hout, Wout] (dtype T)
template <typename T>
__global__ void resize_v_norm_kernel(const unsigned char* __restrict__ src, T* __restrict__ out, const int* __restrict__ bounds,
                                     const int* __restrict__ coef, int ksize, int H, int Hout, int Wout, int C,
                                     const float* __restrict__ mean, const float* __restrict__ stdv, float rescale,
                                     int do_normalize) {
  const int64_t total = (int64_t)C * Hout * Wout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wout);
    const int yy = (int)((i / Wout) % Hout);
    const int c = (int)(i / ((int64_t)Wout * Hout));
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const unsigned char* p = src + ((int64_t)y0 * Wout + xx) * C + c;
    const int* k = coef + (int64_t)yy * ksize;
    int ss = 1 << 21;
    for (int y = 0; y < n; ++y) ss += (int)p[(int64_t)y * Wout * C] * k[y];
    float v = (float)clip8(ss >> 22) * rescale;  // HF rescale: image * scale in float32
    if (do_normalize) v = (v - mean[c]) / stdv[c];  // HF normalize: (image - mean) / std in float32 (true division)
    out[i] = from_f<T>(v);
  }
}

// nearest-neighbour gather: src [K, H, W] uint8 -> out [K, Hout, Wout] (dtype T)
template <typename T>
__global__ void nearest_u8_kernel(const unsigned char* __restrict__ src, T* __restrict__ out, const int* __restrict__ ys,
                                  const int* __restrict__ xs, int K, int H, int W, int Hout, int Wout) {
  const int64_t total = (int64_t)K * Hout * Wout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wout);
    const int yy = (int)((i / Wout) % Hout);
    const int m = (int)(i / ((int64_t)Wout * Hout));
    out[i] = from_f<T>((float)src[((int64_t)m * H + ys[yy]) * W + xs[xx]]);
  }
}

// pad-mode masks (process_regions with image_aspect_ratio == "pad", mm_utils.py:505-531): the mask is centred on a zero square of
// side max(H, W) and the HF processor then resizes that square with Pillow's bicubic filter (uint8, one channel).  The square is
// never materialised: the horizontal pass reads through the padding offsets (outside the mask = 0).
// src [K, H, W] -> tmp [K, side, Wout]
__global__ void mask_pad_resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                         const int* __restrict__ bounds, const int* __restrict__ coef, int ksize, int K, int H, int W,
                                         int side, int pad_top, int pad_left, int Wout) {
  const int64_t total = (int64_t)K * side * Wout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wout);
    const int y = (int)((i / Wout) % side);
    const int m = (int)(i / ((int64_t)Wout * side));
    const int py = y - pad_top;
    int ss = 1 << 21;
    if (py >= 0 && py < H) {
      const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
      const unsigned char* row = src + ((int64_t)m * H + py) * W;
      const int* k = coef + (int64_t)xx * ksize;
      for (int x = 0; x < n; ++x) {
        const int px = x0 + x - pad_left;
        if (px >= 0 && px < W) ss += (int)row[px] * k[x];
      }
    }
    dst[i] = (unsigned char)clip8(ss >> 22);
  }
}

// tmp [K, side, Wout] -> out [K, Hout, Wout] (dtype T): the vertical pass; rescale_factor 1.0, no normalisation (process_regions)
template <typename T>
__global__ void mask_resize_v_kernel(const unsigned char* __restrict__ src, T* __restrict__ out, const int* __restrict__ bounds,
                                     const int* __restrict__ coef, int ksize, int K, int side, int Hout, int Wout) {
  const int64_t total = (int64_t)K * Hout * Wout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wout);
    const int yy = (int)((i / Wout) % Hout);
    const int m = (int)(i / ((int64_t)Wout * Hout));
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const unsigned char* p = src + ((int64_t)m * side + y0) * Wout + xx;
    const int* k = coef + (int64_t)yy * ksize;
    int ss = 1 << 21;
    for (int y = 0; y < n; ++y) ss += (int)p[(int64_t)y * Wout] * k[y];
    out[i] = from_f<T>((float)clip8(ss >> 22));
  }
}

inline int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  const int64_t cap = (int64_t)srgpt_device_cus() * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int srgpt_image_resize_normalize(const void* src_u8, int H, int W, int C, const int* hbounds, const int* hcoef,
                                            int hk, const int* vbounds, const int* vcoef, int vk, int Hout, int Wout,
                                            void* tmp_u8, void* out, const float* mean, const float* stdv, float rescale,
                                            int do_normalize, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(src_u8 && hbounds && hcoef && vbounds && vcoef && tmp_u8 && out, SRGPT_ERR_ARG,
              "srgpt_image_resize_normalize: null pointer");
  SRGPT_CHECK(H > 0 && W > 0 && C > 0 && Hout > 0 && Wout > 0 && hk > 0 && vk > 0, SRGPT_ERR_ARG,
              "srgpt_image_resize_normalize: bad shape");
  SRGPT_CHECK(!do_normalize || (mean && stdv), SRGPT_ERR_ARG, "srgpt_image_resize_normalize: normalise needs mean/std");
  SRGPT_CHECK(dtype == SRGPT_F32 || dtype == SRGPT_BF16, SRGPT_ERR_ARG, "srgpt_image_resize_normalize: bad dtype %d", dtype);
  hipStream_t s = as_stream(stream);
  const int64_t t1 = (int64_t)H * Wout * C, t2 = (int64_t)C * Hout * Wout;
  hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for(t1)), dim3(256), 0, s, (const unsigned char*)src_u8, (unsigned char*)tmp_u8,
                     hbounds, hcoef, hk, H, W, Wout, C);
  SRGPT_LAUNCH_CHECK();
  if (dtype == SRGPT_BF16)
    hipLaunchKernelGGL(resize_v_norm_kernel<bf16_t>, dim3(grid_for(t2)), dim3(256), 0, s, (const unsigned char*)tmp_u8, (bf16_t*)out,
                       vbounds, vcoef, vk, H, Hout, Wout, C, mean, stdv, rescale, do_normalize);
  else
    hipLaunchKernelGGL(resize_v_norm_kernel<float>, dim3(grid_for(t2)), dim3(256), 0, s, (const unsigned char*)tmp_u8, (float*)out,
                       vbounds, vcoef, vk, H, Hout, Wout, C, mean, stdv, rescale, do_normalize);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_mask_resize_nearest(const void* src_u8, int K, int H, int W, const int* ys, const int* xs, int Hout,
                                         int Wout, void* out, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(src_u8 && ys && xs && out, SRGPT_ERR_ARG, "srgpt_mask_resize_nearest: null pointer");
  SRGPT_CHECK(K > 0 && H > 0 && W > 0 && Hout > 0 && Wout > 0, SRGPT_ERR_ARG, "srgpt_mask_resize_nearest: bad shape");
  SRGPT_CHECK(dtype == SRGPT_F32 || dtype == SRGPT_BF16, SRGPT_ERR_ARG, "srgpt_mask_resize_nearest: bad dtype %d", dtype);
  hipStream_t s = as_stream(stream);
  const int64_t t = (int64_t)K * Hout * Wout;
  if (dtype == SRGPT_BF16)
    hipLaunchKernelGGL(nearest_u8_kernel<bf16_t>, dim3(grid_for(t)), dim3(256), 0, s, (const unsigned char*)src_u8, (bf16_t*)out, ys,
                       xs, K, H, W, Hout, Wout);
  else
    hipLaunchKernelGGL(nearest_u8_kernel<float>, dim3(grid_for(t)), dim3(256), 0, s, (const unsigned char*)src_u8, (float*)out, ys, xs,
                       K, H, W, Hout, Wout);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_mask_pad_resize(const void* src_u8, int K, int H, int W, const int* hbounds, const int* hcoef, int hk,
                                     const int* vbounds, const int* vcoef, int vk, int Hout, int Wout, void* tmp_u8, void* out,
                                     int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(src_u8 && hbounds && hcoef && vbounds && vcoef && tmp_u8 && out, SRGPT_ERR_ARG, "srgpt_mask_pad_resize: null pointer");
  SRGPT_CHECK(K > 0 && H > 0 && W > 0 && Hout > 0 && Wout > 0 && hk > 0 && vk > 0, SRGPT_ERR_ARG, "srgpt_mask_pad_resize: bad shape");
  SRGPT_CHECK(dtype == SRGPT_F32 || dtype == SRGPT_BF16, SRGPT_ERR_ARG, "srgpt_mask_pad_resize: bad dtype %d", dtype);
  hipStream_t s = as_stream(stream);
  const int side = H > W ? H : W, pad_top = (side - H) / 2, pad_left = (side - W) / 2;  // pad_to_square, mm_utils.py:505-514
  const int64_t t1 = (int64_t)K * side * Wout, t2 = (int64_t)K * Hout * Wout;
  hipLaunchKernelGGL(mask_pad_resize_h_kernel, dim3(grid_for(t1)), dim3(256), 0, s, (const unsigned char*)src_u8, (unsigned char*)tmp_u8,
                     hbounds, hcoef, hk, K, H, W, side, pad_top, pad_left, Wout);
  SRGPT_LAUNCH_CHECK();
  if (dtype == SRGPT_BF16)
    hipLaunchKernelGGL(mask_resize_v_kernel<bf16_t>, dim3(grid_for(t2)), dim3(256), 0, s, (const unsigned char*)tmp_u8, (bf16_t*)out,
                       vbounds, vcoef, vk, K, side, Hout, Wout);
  else
    hipLaunchKernelGGL(mask_resize_v_kernel<float>, dim3(grid_for(t2)), dim3(256), 0, s, (const unsigned char*)tmp_u8, (float*)out,
                       vbounds, vcoef, vk, K, side, Hout, Wout);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
