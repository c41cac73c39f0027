// LayerNorm (+ optional activation) and RMSNorm: HBM-bound row kernels, 16-byte vector loads,
// fp32 statistics, one block (256 threads) per row.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ b, T* __restrict__ y, int cols,
                                                        float eps, int act) {
  constexpr int VEC = Vec16<T>::N;
  __shared__ float red[16];
  const int row = blockIdx.x;
  const T* xr = x + (size_t)row * cols;
  T* yr = y + (size_t)row * cols;
  const int nch = cols / VEC;
  float s = 0.f;
  for (int c = threadIdx.x; c < nch; c += 256) {
    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(xr + c * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += v.get(i);
  }
  const float mean = block_sum(s, red) / (float)cols;
  float q = 0.f;
  for (int c = threadIdx.x; c < nch; c += 256) {
    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(xr + c * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float d = v.get(i) - mean;
      q += d * d;
    }
  }
  const float rstd = rsqrtf(block_sum(q, red) / (float)cols + eps);
  for (int c = threadIdx.x; c < nch; c += 256) {
    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(xr + c * VEC);
    const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(w + c * VEC);
    const Vec16<T> be = *reinterpret_cast<const Vec16<T>*>(b + c * VEC);
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float t = (v.get(i) - mean) * rstd * g.get(i) + be.get(i);
      if (act != SRGPT_ACT_NONE) t = apply_act<T>(rnd<T>(t), act);
      o.set(i, t);
    }
    *reinterpret_cast<Vec16<T>*>(yr + c * VEC) = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ y, int cols, float eps) {
  constexpr int VEC = Vec16<T>::N;
  __shared__ float red[16];
  const int row = blockIdx.x;
  const T* xr = x + (size_t)row * cols;
  T* yr = y + (size_t)row * cols;
  const int nch = cols / VEC;
  float s = 0.f;
  for (int c = threadIdx.x; c < nch; c += 256) {
    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(xr + c * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += v.get(i) * v.get(i);
  }
  const float r = rsqrtf(block_sum(s, red) / (float)cols + eps);
  for (int c = threadIdx.x; c < nch; c += 256) {
    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(xr + c * VEC);
    const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(w + c * VEC);
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, g.get(i) * rnd<T>(v.get(i) * r));  // weight * h.to(input_dtype)
    *reinterpret_cast<Vec16<T>*>(yr + c * VEC) = o;
  }
}

}  // namespace

extern "C" int srgpt_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int cols, float eps,
                               int act, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(x && w && b && y, SRGPT_ERR_ARG, "srgpt_layernorm: null pointer");
  SRGPT_CHECK(rows > 0 && cols > 0, SRGPT_ERR_ARG, "srgpt_layernorm: bad shape");
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(cols % vec == 0, SRGPT_ERR_ARG, "srgpt_layernorm: cols=%d must be a multiple of %d", cols, vec);
  if (dtype == SRGPT_BF16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, cols, eps, act);
  else if (dtype == SRGPT_F32)
    hipLaunchKernelGGL(layernorm_kernel<float>, dim3(rows), dim3(256), 0, as_stream(stream), (const float*)x,
                       (const float*)w, (const float*)b, (float*)y, cols, eps, act);
  else {
    srgpt_set_error("srgpt_layernorm: bad dtype %d", dtype);
    return SRGPT_ERR_ARG;
  }
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

extern "C" int srgpt_rmsnorm(const void* x, const void* w, void* y, int rows, int cols, float eps, int dtype,
                             srgpt_stream_t stream) {
  SRGPT_CHECK(x && w && y, SRGPT_ERR_ARG, "srgpt_rmsnorm: null pointer");
  SRGPT_CHECK(rows > 0 && cols > 0, SRGPT_ERR_ARG, "srgpt_rmsnorm: bad shape");
  const int vec = dtype == SRGPT_BF16 ? 8 : 4;
  SRGPT_CHECK(cols % vec == 0, SRGPT_ERR_ARG, "srgpt_rmsnorm: cols=%d must be a multiple of %d", cols, vec);
  if (dtype == SRGPT_BF16)
    hipLaunchKernelGGL(rmsnorm_kernel<bf16_t>, dim3(rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                       (const bf16_t*)w, (bf16_t*)y, cols, eps);
  else if (dtype == SRGPT_F32)
    hipLaunchKernelGGL(rmsnorm_kernel<float>, dim3(rows), dim3(256), 0, as_stream(stream), (const float*)x,
                       (const float*)w, (float*)y, cols, eps);
  else {
    srgpt_set_error("srgpt_rmsnorm: bad dtype %d", dtype);
    return SRGPT_ERR_ARG;
  }
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
