// checks the DPP / permlane cross-lane helpers of csrc/common.h against host sums (integer-valued floats: exact)
#include "../spatialrgpt_amd/csrc/common.h"
#include <vector>
void srgpt_set_error(const char*, ...) {}
template <int N> __global__ void k_sum(const float* in, float* out) { out[threadIdx.x] = lanes_sum<N>(in[threadIdx.x]); }
template <int N> __global__ void k_max(const float* in, float* out) { out[threadIdx.x] = lanes_max<N>(in[threadIdx.x]); }
template <int S> __global__ void k_str(const float* in, float* out) { out[threadIdx.x] = strided_sum<S>(in[threadIdx.x]); }
__global__ void k_swap(const float* in, float* out) {
  const float v = in[threadIdx.x];
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  out[threadIdx.x] = __builtin_bit_cast(float, r[0]); out[64 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  out[128 + threadIdx.x] = __builtin_bit_cast(float, q[0]); out[192 + threadIdx.x] = __builtin_bit_cast(float, q[1]);
}
int main() {
  std::vector<float> h(64), o(256);
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37 + 11) % 101);
  float *d, *e; hipMalloc(&d, 256); hipMalloc(&e, 1024); hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o.data(), e, 1024, hipMemcpyDeviceToHost);
  printf("permlane16_swap(v,v) first : "); for (int i = 0; i < 64; i += 8) printf("%g<-lane? ", o[i]); printf("\n");
  auto src_of = [&](float x) { for (int i = 0; i < 64; ++i) if (h[i] == x) return i; return -1; };
  printf("lane->source lane, permlane16 first : "); for (int i = 0; i < 64; ++i) printf("%d ", src_of(o[i])); printf("\n");
  printf("lane->source lane, permlane16 second: "); for (int i = 0; i < 64; ++i) printf("%d ", src_of(o[64 + i])); printf("\n");
  printf("lane->source lane, permlane32 first : "); for (int i = 0; i < 64; ++i) printf("%d ", src_of(o[128 + i])); printf("\n");
  printf("lane->source lane, permlane32 second: "); for (int i = 0; i < 64; ++i) printf("%d ", src_of(o[192 + i])); printf("\n");
#define CHECK_SUM(N) { hipLaunchKernelGGL(k_sum<N>, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o.data(), e, 256, hipMemcpyDeviceToHost); int bad = 0; \
    for (int i = 0; i < 64; ++i) { float r = 0; for (int j = 0; j < N; ++j) r += h[(i / N) * N + j]; bad += (r != o[i]); } printf("lanes_sum<%d>: %d wrong lanes\n", N, bad); }
#define CHECK_MAX(N) { hipLaunchKernelGGL(k_max<N>, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o.data(), e, 256, hipMemcpyDeviceToHost); int bad = 0; \
    for (int i = 0; i < 64; ++i) { float r = -1; for (int j = 0; j < N; ++j) r = fmaxf(r, h[(i / N) * N + j]); bad += (r != o[i]); } printf("lanes_max<%d>: %d wrong lanes\n", N, bad); }
#define CHECK_STR(S) { hipLaunchKernelGGL(k_str<S>, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o.data(), e, 256, hipMemcpyDeviceToHost); int bad = 0; \
    for (int i = 0; i < 64; ++i) { float r = 0; for (int j = i % S; j < 64; j += S) r += h[j]; bad += (r != o[i]); } printf("strided_sum<%d>: %d wrong lanes\n", S, bad); }
  CHECK_SUM(2) CHECK_SUM(4) CHECK_SUM(8) CHECK_SUM(16) CHECK_SUM(32) CHECK_SUM(64)
  CHECK_MAX(16) CHECK_MAX(64)
  CHECK_STR(1) CHECK_STR(2) CHECK_STR(4) CHECK_STR(8) CHECK_STR(16) CHECK_STR(32)
  return 0;
}
