// Fused GEMM epilogue shared by the GEMM kernels (gemm.hip, gemm256.hip): bias -> round -> activation -> round -> residual ->
// round, plain or deconv pixel-shuffle addressing, bf16 / fp32 store.  Rounding points mirror PyTorch materialising each
// intermediate in the storage dtype.
#pragma once
#include "common.h"

struct SrgptGemmEpilogue {
  const void* bias;
  const void* residual;
  void* C;
  int M, N, ldc, act, bias_mod, res_mod, out_f32, out_mode, gw;
  float* partial;  // split-K: fp32 slabs [splits][M][N] (deterministic: reduced in slab order by splitk_reduce_kernel)
  int splits, tiles_per_split;
  const float* wscale;  // fp8 weights (srgpt_gemm_w8): one fp32 scale per output column, applied to the accumulator; else NULL
  // RMSNorm / LayerNorm of the OUTPUT rows, written beside C (srgpt_gemm_norm: the residual stream and the next block's normalised
  // input from one pass): fused into the split-K reduction when there is one, a separate norm launch otherwise.  norm_y NULL = none.
  const void* norm_w;
  const void* norm_b;  // LayerNorm bias (norm_kind SRGPT_NORM_LAYER)
  void* norm_y;
  float norm_eps;
  int norm_kind;
  // RoPE + KV-cache append of the OUTPUT rows (srgpt_gemm_rope_kv_append: the q/k/v projection of the prefill): rope_k NULL = none
  void* rope_k;
  void* rope_v;
  const int* rope_pos0;
  const void* rope_cos;
  const void* rope_sin;
  int rope_T, rope_Hq, rope_Hkv, rope_D, rope_max_pos;
  // SwiGLU of a stacked [gate; up] weight (srgpt_gemm_swiglu, whole-M kernel only): W is [2 * swiglu_inter, K], a block multiplies 64
  // gate and the 64 matching up columns, C is [M, swiglu_inter] = rnd(rnd(silu(gate)) * up).  0 = none.
  int swiglu_inter;
};
typedef SrgptGemmEpilogue Epilogue;

namespace {

template <typename T>
__device__ __forceinline__ void epilogue_store(const Epilogue& e, int m, int n, float acc) {
  if (m >= e.M || n >= e.N) return;
  float v = e.wscale ? acc * e.wscale[n] : acc;
  if (e.bias) {
    const int bi = e.bias_mod > 0 ? n % e.bias_mod : n;
    v += to_f(reinterpret_cast<const T*>(e.bias)[bi]);
  }
  v = rnd<T>(v);  // nn.Linear / conv output is materialised in T
  if (e.act != SRGPT_ACT_NONE) v = rnd<T>(apply_act<T>(v, e.act));
  if (e.residual) {
    const int rm = e.res_mod > 0 ? m % e.res_mod : m;
    v = rnd<T>(v + to_f(reinterpret_cast<const T*>(e.residual)[(size_t)rm * e.N + n]));
  }
  size_t off;
  if (e.out_mode == SRGPT_OUT_DECONV2X) {
    const int cout = e.N >> 2, gg = e.gw * e.gw;
    const int img = m / gg, rem = m - img * gg, i = rem / e.gw, j = rem - i * e.gw;
    const int tap = n / cout, co = n - tap * cout, a = tap >> 1, b = tap & 1;
    const int ow = 2 * e.gw;
    off = ((size_t)img * ow * ow + (size_t)(2 * i + a) * ow + (2 * j + b)) * cout + co;
  } else {
    off = (size_t)m * e.ldc + n;
  }
  if (e.out_f32)
    reinterpret_cast<float*>(e.C)[off] = v;
  else
    reinterpret_cast<T*>(e.C)[off] = from_f<T>(v);
}

// The same epilogue for one 32x32 MFMA accumulator tile of this lane: 16 values at rows mb + (r & 3) + 8 * (r >> 2), column n
// (D layout of v_mfma_f32_32x32x16_bf16 with mb = tile row + 4 * (lane >> 5)).  Per-element epilogue_store() serialises a
// bias load, a residual load and their waits 16 times per tile; here the bias is ONE load per tile (the column is fixed per
// lane), the 16 residual loads are issued back to back before anything consumes them, and the uniform switches (bias /
// activation / residual / output mode) are taken once per tile instead of once per element.  Same arithmetic, same roundings.
template <typename T>
__device__ __forceinline__ void epilogue_tile32(const Epilogue& e, int mb, int n, const f32x16& a) {
  if (n >= e.N || mb >= e.M) return;
  const T* bias = reinterpret_cast<const T*>(e.bias);
  const T* resid = reinterpret_cast<const T*>(e.residual);
  float b = 0.f;
  if (bias) b = to_f(bias[e.bias_mod > 0 ? n % e.bias_mod : n]);
  const float sc = e.wscale ? e.wscale[n] : 1.f;
  float res[16];
  if (resid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(mb + (r & 3) + 8 * (r >> 2), e.M - 1);
      const int rm = e.res_mod > 0 ? m % e.res_mod : m;
      res[r] = to_f(resid[(size_t)rm * e.N + n]);
    }
  }
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = rnd<T>(a[r] * sc + b);  // nn.Linear / conv output is materialised in T
  switch (e.act) {
    case SRGPT_ACT_GELU_ERF:
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = rnd<T>(gelu_erf(v[r]));
      break;
    case SRGPT_ACT_GELU_TANH:
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = rnd<T>(gelu_tanh(v[r]));
      break;
    case SRGPT_ACT_SILU:
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = rnd<T>(silu(v[r]));
      break;
    case SRGPT_ACT_QUICK_GELU:
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = rnd<T>(v[r] / (1.f + __expf(-1.702f * v[r])));
      break;
    default: break;
  }
  if (resid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = rnd<T>(v[r] + res[r]);
  }
  if (e.out_mode == SRGPT_OUT_DECONV2X) {
    const int cout = e.N >> 2, gg = e.gw * e.gw, ow = 2 * e.gw;
    const int tap = n / cout, co = n - tap * cout, ta = tap >> 1, tb = tap & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb + (r & 3) + 8 * (r >> 2);
      if (m < e.M) {
        const int img = m / gg, rem = m - img * gg, i = rem / e.gw, j = rem - i * e.gw;
        const size_t off = ((size_t)img * ow * ow + (size_t)(2 * i + ta) * ow + (2 * j + tb)) * cout + co;
        if (e.out_f32)
          reinterpret_cast<float*>(e.C)[off] = v[r];
        else
          reinterpret_cast<T*>(e.C)[off] = from_f<T>(v[r]);
      }
    }
  } else if (e.out_f32) {
    float* c = reinterpret_cast<float*>(e.C) + (size_t)mb * e.ldc + n;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (mb + (r & 3) + 8 * (r >> 2) < e.M) c[(size_t)((r & 3) + 8 * (r >> 2)) * e.ldc] = v[r];
  } else {
    T* c = reinterpret_cast<T*>(e.C) + (size_t)mb * e.ldc + n;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (mb + (r & 3) + 8 * (r >> 2) < e.M) c[(size_t)((r & 3) + 8 * (r >> 2)) * e.ldc] = from_f<T>(v[r]);
  }
}

}  // namespace
