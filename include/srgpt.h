/*
 * srgpt.h -- C ABI of libsrgpt_hip.so: the MI355X (gfx950) kernels behind SpatialRGPT's region-grounded
 * multimodal forward/generate path.
 *
 * The reference (AnjieCheng/SpatialRGPT @ 2024-12-18) is pure Python; it has no FFI/plugin seam of its
 * own.  Each entry point below therefore names the reference *function* whose
 * arithmetic it replaces (file:line relative to the reference root).  The Python host
 * (spatialrgpt_amd/) keeps the reference's call signatures and binds these symbols with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name starts with `h_`; tensors are dense row-major
 *   - `dtype` selects the storage type of activations AND weights: SRGPT_F32 or SRGPT_BF16;
 *     accumulation is always fp32.  bf16 is the parity/benchmark dtype, fp32 exists for tight parity
 *     checks of the same code path
 *   - every function only enqueues work on `stream` (a hipStream_t) and returns immediately;
 *     nothing allocates, nothing synchronises (except srgpt_graph_* creation), no globals
 *   - return value: 0 on success, negative SRGPT_ERR_* otherwise; srgpt_last_error() gives the text
 *     (thread-local).  The Python host maps codes to the exceptions the reference raises.
 */
#ifndef SRGPT_H_
#define SRGPT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* srgpt_stream_t; /* hipStream_t */

enum { SRGPT_F32 = 0, SRGPT_BF16 = 1 };
enum { SRGPT_ACT_NONE = 0, SRGPT_ACT_GELU_ERF = 1, SRGPT_ACT_GELU_TANH = 2, SRGPT_ACT_SILU = 3, SRGPT_ACT_QUICK_GELU = 4 };
enum {
  SRGPT_OK = 0,
  SRGPT_ERR_ARG = -1,      /* -> ValueError */
  SRGPT_ERR_UNSUPPORTED = -2, /* -> NotImplementedError */
  SRGPT_ERR_LAUNCH = -3,   /* -> RuntimeError */
  SRGPT_ERR_STATE = -4     /* -> RuntimeError */
};
/* output-row mapping of srgpt_gemm */
enum {
  SRGPT_OUT_PLAIN = 0,     /* C[m, n]                                                       */
  SRGPT_OUT_DECONV2X = 1   /* ConvTranspose2d(k=2,s=2) pixel shuffle, channels-last (see gemm) */
};

const char* srgpt_last_error(void);
int srgpt_abi_version(void);
/* number of compute units of the current device (host query, used to size persistent grids) */
int srgpt_device_cus(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM family: every nn.Linear / conv-as-GEMM on the path.
 *   C[M,N] = act(A[M,K] @ W[N,K]^T + bias[N]) + residual
 * replaces: HF SiglipAttention/SiglipMLP Linear layers (third-party, called from
 * multimodal_encoder/vision_encoder.py:119-129), region_extractor/base_extractor.py:93-96,125-126
 * (ConvTranspose2d + connectors), multimodal_projector/base_projector.py:76-79,
 * transformers_replace/models/llama/modeling_llama.py:300-336 (q/k/v/o), :194-223 (MLP), :1044 (lm_head).
 *   - A row stride lda, C row stride ldc (elements); W dense [N,K]; K % 8 == 0 (bf16) / K % 4 == 0 (f32)
 *   - bias may be NULL; bias index is n % bias_mod when bias_mod > 0 (deconv: bias per channel)
 *   - residual may be NULL; residual row is (m % res_mod) when res_mod > 0 (position embeddings), ld = N
 *   - out_f32 != 0: C is written as fp32 regardless of dtype (logits.float(), modeling_llama.py:1045)
 *   - out_mode SRGPT_OUT_DECONV2X: A rows are pixels (img, i, j) of a [n_img, gw, gw] grid, N = 4*Cout
 *     ordered n = (a*2+b)*Cout + co; C is the channels-last [n_img, 2gw, 2gw, Cout] map and
 *     element (m,n) lands at pixel (2i+a, 2j+b), channel co.  `gw` passes the grid width.
 *   - ws / ws_bytes: optional fp32 workspace for deterministic split-K (small-M shapes whose tile grid cannot fill
 *     256 CUs: K is cut into <= 8 slabs written to ws and reduced in slab order); NULL disables split-K.
 *     srgpt_gemm_ws_bytes(M, N) is always sufficient.
 * --------------------------------------------------------------------------------------------- */
int64_t srgpt_gemm_ws_bytes(int M, int N);
int srgpt_gemm(const void* A, const void* W, const void* bias, const void* residual, void* C,
               int M, int N, int K, int lda, int ldc, int act, int bias_mod, int res_mod,
               int out_f32, int out_mode, int gw, void* ws, int64_t ws_bytes, int dtype, srgpt_stream_t stream);

/* C[M,N] = A[M,K] @ W[N,K]^T + bias[n] + residual[M,N] and Y[M,N] = norm(C) from one call -- the tail of an attention / MLP block
 * followed by the next block's input norm: LlamaDecoderLayer.forward modeling_llama.py:611-684 (o_proj or down_proj, the residual
 * add, then LlamaRMSNorm :61-75; norm_kind SRGPT_NORM_RMS, bias and norm_b NULL) and the HF SigLIP / CLIP encoder layer (out_proj
 * or fc2 + bias, the residual add, then nn.LayerNorm; SRGPT_NORM_LAYER).  Dense rows (lda = K, ldc = N), no activation.  When the
 * product is split over K the norm rides in the slab reduction (one launch and one pass over the rows less); otherwise it is the
 * srgpt_rmsnorm / srgpt_layernorm launch.  Either way C and Y are bit-identical to srgpt_gemm followed by the norm.  C may alias
 * residual, Y may alias A (not C).  ws as for srgpt_gemm. */
#define SRGPT_NORM_RMS 1
#define SRGPT_NORM_LAYER 2
int srgpt_gemm_norm(const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K, void* ws,
                    int64_t ws_bytes, int norm_kind, const void* norm_w, const void* norm_b, void* Y, float norm_eps, int dtype,
                    srgpt_stream_t stream);

/* out[M, I] = silu(A @ Wg^T) * (A @ Wu^T) for the stacked weight Wgu = [Wg; Wu] ([2 I, K], the layout srgpt_gemv's swiglu mode
 * reads) -- LlamaMLP's gate / up products and srgpt_silu_mul (below) in one call (modeling_llama.py:194-223).  At 225 .. 272 rows
 * (the bs = 1 prefill) the activation is the epilogue of the product and the [M, 2 I] intermediate is never written; every other
 * shape runs srgpt_gemm into gu_scratch ([M, 2 I], may be NULL only when the fused form applies) and srgpt_silu_mul.  Either way
 * the result is srgpt_gemm + srgpt_silu_mul to the last bit. */
int srgpt_gemm_swiglu(const void* A, const void* Wgu, void* out, int M, int I, int K, void* gu_scratch, void* ws,
                      int64_t ws_bytes, int dtype, srgpt_stream_t stream);

/* qkv[B*T, (Hq + 2 Hkv) D] = A[B*T, K] @ W^T, then RoPE on the q and k heads and the append of k / v to the caches -- the q/k/v
 * projection of a prefill and srgpt_rope_kv_append (below: same arguments, same arithmetic) in one call
 * (LlamaFlashAttention2.forward modeling_llama.py:398-456).  When the product is split over K the rotation rides in the slab
 * reduction; otherwise it is the srgpt_rope_kv_append launch.  Every byte written (qkv, kcache, vcache) is the same either way. */
int srgpt_gemm_rope_kv_append(const void* A, const void* W, void* qkv, int K, void* ws, int64_t ws_bytes, void* kcache,
                              void* vcache, const int* pos0, const void* cos_tab, const void* sin_tab, int B, int T, int Hq,
                              int Hkv, int D, int max_pos, int dtype, srgpt_stream_t stream);

/* GEMM with weight-only fp8 quantisation (the prefill-side companion of srgpt_gemv_w8, BASELINE config 5):
 *   C[M,N] = act( (A[M,K] @ fp8(W8[N,K])^T) * wscale[n] + bias[n] ) + residual[M,N]
 * A / bias / residual / C bf16 (C fp32 if out_f32), W8 = OCP e4m3fn bytes, wscale = one fp32 scale per weight row, fp32
 * accumulation over the fp8 values widened exactly to bf16; with power-of-two scales this equals srgpt_gemm on the dequantised
 * bf16 weights bit for bit.  K % 64 == 0 takes the 256 x 256 MFMA kernel, any other K a scalar fallback.  ws: optional fp32
 * split-K workspace (srgpt_gemm_ws_bytes). */
int srgpt_gemm_w8(const void* A, const void* W8, const float* wscale, const void* bias, const void* residual, void* C,
                  int M, int N, int K, int lda, int ldc, int act, int out_f32, void* ws, int64_t ws_bytes,
                  srgpt_stream_t stream);

/* W8A8 on the fp8 matrix pipe (v_mfma_scale_f32_16x16x128_f8f6f4) -- the opt-in prefill mode of BASELINE configs[4]; the
 * reference has no counterpart (it offers bitsandbytes 8/4-bit loading, llava/model/builder.py:51-60).
 * srgpt_quant_rows_e4m3: x [M, K] bf16 (row stride ldx) -> q [M, K] OCP e4m3fn bytes + scale [M] fp32, per row (token):
 *   scale = the smallest power of two with max|x[m,:]| / scale <= 448, q = e4m3fn(x / scale) round-to-nearest-even (the rule the
 *   weights are quantised with).  K % 8 == 0, 16-byte aligned rows.
 * srgpt_gemm_w8a8: C[M,N] = ( (A8[M,K] @ W8[N,K]^T) * ascale[m] * wscale[n] + bias[n] ) + residual[M,N], fp32 accumulation of
 *   exact e4m3 x e4m3 products, bias / residual / C bf16 (C fp32 if out_f32) with srgpt_gemm's rounding points (the product is
 *   rounded to bf16 before the residual is added): the GEMM of the dequantised operands up to the order of the fp32 sum.  K % 128 == 0, K >= 256, lda % 16 == 0, 16-byte aligned operands; anything else
 *   returns SRGPT_ERR_UNSUPPORTED (no fallback).  ws: optional fp32 split-K workspace (srgpt_gemm_ws_bytes).
 *   Scales: srgpt_quant_rows_e4m3 and the weight quantiser produce POWERS OF TWO, for which scaling commutes exactly with the
 *   fp32 sums; arbitrary fp32 scales are accepted, but the split-K path then scales the per-split slabs by ascale before the
 *   reduction and the un-split path after the whole sum -- results stay deterministic for a given (shape, device, ws_bytes),
 *   yet differ by fp32 roundings between the two paths.  Callers that need path-independent bits pass power-of-two scales. */
int srgpt_quant_rows_e4m3(const void* x, void* q, float* scale, int M, int K, int ldx, srgpt_stream_t stream);
/* ABI 5: the same quantisation with the producer of the rows fused in -- what the W8A8 prefill runs, so the bf16 intermediate is
 * neither written nor read.  Bit-identical to srgpt_rmsnorm / srgpt_silu_mul followed by srgpt_quant_rows_e4m3 (tested); K (inter)
 * up to 16384 columns, else SRGPT_ERR_UNSUPPORTED.
 *   _rmsnorm: rows of LlamaRMSNorm(x; norm_w, eps) (modeling_llama.py:61-75), x [M, K] bf16 with row stride ldx
 *   _swiglu:  rows of silu(gate) * up (LlamaMLP.forward, modeling_llama.py:194-223) of gate_up [M, 2 * inter] = [gate | up] */
int srgpt_quant_rows_e4m3_rmsnorm(const void* x, const void* norm_w, float eps, void* q, float* scale, int M, int K, int ldx,
                                  srgpt_stream_t stream);
int srgpt_quant_rows_e4m3_swiglu(const void* gate_up, void* q, float* scale, int M, int inter, srgpt_stream_t stream);
int srgpt_gemm_w8a8(const void* A8, const float* ascale, const void* W8, const float* wscale, const void* bias,
                    const void* residual, void* C, int M, int N, int K, int lda, int ldc, int out_f32, void* ws,
                    int64_t ws_bytes, srgpt_stream_t stream);

/* Decode-only fused GEMVs (any batch; one bf16 / fp8 row: VALU kernel, 2+ rows: MFMA kernel, 16 rows per weight pass -- all rows of
 * one call on the same kernel), weights streamed once per pass from HBM:
 *   norm_w != NULL : x <- RMSNorm(x) * norm_w first (modeling_llama.py:61-75) (rounded to dtype like torch)
 *   swiglu != 0    : W = [gate rows(N); up rows(N)], out[n] = silu(gate.x) * (up.x) (modeling_llama.py:221)
 *   residual       : out = residual + (W x)   (decoder layer residual adds, modeling_llama.py:650-684)
 */
int srgpt_gemv(const void* x, const void* W, const void* norm_w, float norm_eps, const void* residual,
               void* out, int batch, int N, int K, int swiglu, int out_f32, int dtype, srgpt_stream_t stream);
/* Same product with weight-only fp8 quantisation (W8A16, BASELINE config 5): W8 = OCP e4m3fn bytes [N (2N if swiglu), K],
 * wscale = one fp32 scale per weight row, x / norm_w / residual / out bf16 (out fp32 if out_f32):
 *   out[b, n] = round_bf16( (sum_k x[b,k] * fp8(W8[n,k])) * wscale[n] ), fp32 accumulation; same fusions as srgpt_gemv.
 * The reference has no fp8 path (it offers bitsandbytes 8/4-bit loading, builder.py:51-60); this replaces that option. */
int srgpt_gemv_w8(const void* x, const void* W8, const float* wscale, const void* norm_w, float norm_eps,
                  const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32,
                  srgpt_stream_t stream);
/* ABI 8: the two products above with the ROW-STATISTICS HAND-OFF the batched decode step runs on (2+ bf16 rows: the MFMA kernel).
 * LlamaRMSNorm (modeling_llama.py:61-75) needs mean(x^2) of every row of the residual stream; the product that WRITES that row
 * (o_proj / down_proj with the residual add, modeling_llama.py:650-684) has each element in a register, so it publishes per-block
 * partial sums and the product that normalises (q/k/v, gate/up, lm_head) adds them in a fixed order instead of re-reading the
 * rows in every block.  A table is [batch][SRGPT_ROWSS_STRIDE] fp32.
 *   W8 != NULL      : fp8 weights (+ wscale), W ignored; else W bf16
 *   rowss_out       : this product's bf16 output rows get their table written (not with swiglu / out_f32)
 *   rowss_in        : the table of x's rows, published by the call that wrote x (requires norm_w); NULL: computed from x
 * srgpt_gemv_rowss_supported: 1 when (batch, dtype, fp8) takes the kernel that can do this (else call srgpt_gemv / _w8).
 * The statistics' summation order differs from srgpt_gemv's own, so the normalised rows may differ from it by one bf16 ulp;
 * srgpt_llm_decode_step == the composition of THESE entries bit for bit (tests/test_gpu_kernels.py). */
#define SRGPT_ROWSS_STRIDE 512
int srgpt_gemv_rowss_supported(int batch, int dtype, int fp8);
int srgpt_gemv_rowss(const void* x, const void* W, const void* W8, const float* wscale, const void* norm_w, float norm_eps,
                     const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32, const float* rowss_in,
                     float* rowss_out, int packed_rows, srgpt_stream_t stream);
/* ABI 9: the PACKED decode layout of a streamed matrix.  The MFMA kernel behind srgpt_gemv_rowss multiplies 16 weight rows x 32 k
 * per instruction and wants lane (c = lane & 15, g = lane >> 4) to hold row c, k = 32 s + 8 g .. + 8 of the step.  A row-major
 * matrix delivers that through an LDS transpose per 8-KiB stage; the packed array stores the SAME bytes in operand order, so one
 * coalesced 16-byte load per lane is the fragment (fp8 weights, 8 rows per GPU: the four layer products 56 -> 52 us,
 * profiles/r06_skinny_packed.txt).  Layout, `rows` = 4, 8 or 16 rows per granule, e = bytes per element (1: fp8, 2: bf16),
 * kb = 16 / e ... a k block is 64 k (fp8) or 32 k (bf16):
 *   out[ ((n / rows) * (K / kblock) + k / kblock) * rows * 64  +  ((k % 32) / 8 * rows + n % rows) * 16  +  byte ]
 *   byte = (k % 8) * 2 (bf16)   |   ((k % 64) / 32) * 8 + k % 8 (fp8: the lane's 8 k of two consecutive MFMA steps)
 * N is padded to whole granules with zeros (srgpt_packed_bytes); K %% 64 == 0 (fp8) / K %% 32 == 0 (bf16).  A SwiGLU matrix
 * [gate rows (N); up rows (N)] is packed as ONE matrix of 2 N rows (N %% rows == 0).  srgpt_gemv_rowss takes such an array as
 * W / W8 with packed_rows = rows (0: row-major) and returns bit for bit what it returns for the row-major matrix. */
size_t srgpt_packed_bytes(int N, int K, int elem_bytes, int rows);
int srgpt_pack_decode_weights(const void* W, void* out, int N, int K, int elem_bytes, int rows, srgpt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisations.
 * srgpt_layernorm: nn.LayerNorm over the last dim (+ optional activation) -- HF SigLIP layer_norm1/2,
 *   base_projector.py:75 LayerNorm(4C), and LayerNorm2d over channels of a channels-last map followed by
 *   GELU (base_extractor.py:12-24,94-95).
 * srgpt_rmsnorm: LlamaRMSNorm (modeling_llama.py:61-75).
 * --------------------------------------------------------------------------------------------- */
int srgpt_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int cols, float eps,
                    int act, int dtype, srgpt_stream_t stream);
int srgpt_rmsnorm(const void* x, const void* w, void* y, int rows, int cols, float eps, int dtype,
                  srgpt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Attention.
 * srgpt_attention: softmax(scale * Q K^T [+ causal]) V with fp32 softmax, GQA by head index division.
 *   replaces HF SiglipAttention eager/sdpa (non-causal, called via vision_encoder.py:119-129) and
 *   LlamaFlashAttention2 / flash_attn_func (modeling_llama.py:398-566) for prefill.
 *   Q[b, t, h, :]  at q  + b*q_bs  + t*q_ts  + h*q_hs ; K/V[b, s, hk, :] likewise; O dense [b, t, h, d].
 *   causal: key s visible to query t iff s <= t + (Tk - Tq).
 *   kv_len (device int[B], may be NULL): per-row number of valid keys (right-padded batches).
 * --------------------------------------------------------------------------------------------- */
int srgpt_attention(const void* q, const void* k, const void* v, void* o, int B, int Tq, int Tk, int Hq,
                    int Hkv, int D, int64_t q_bs, int64_t q_ts, int64_t q_hs, int64_t k_bs, int64_t k_ts,
                    int64_t k_hs, int64_t v_bs, int64_t v_ts, int64_t v_hs, float scale, int causal,
                    const int* kv_len, int dtype, srgpt_stream_t stream);

/* RoPE + KV-cache append (modeling_llama.py:81-191 rotary, :451-456 cache growth -- here a static cache).
 *   qkv: [B*T, (Hq+2Hkv)*D] packed projections; q is rotated in place; rotated k and v are written to
 *   kcache/vcache [B, Hkv, max_pos, D] at positions pos0[b] + t  (pos0: device int[B], NULL = 0).
 *   cos_tab/sin_tab: [max_pos, D/2] in `dtype`, built ONCE at load time by the host exactly as
 *   LlamaRotaryEmbedding does (fp32 angle = position * inv_freq, cos/sin, cast to dtype; linear
 *   scaling of language_model/builder.py:31-38 folded into inv_freq) -- a table, not on-device trig. */
int srgpt_rope_kv_append(void* qkv, void* kcache, void* vcache, const int* pos0, const void* cos_tab,
                         const void* sin_tab, int B, int T, int Hq, int Hkv, int D, int max_pos, int dtype,
                         srgpt_stream_t stream);

/* Decode attention for one new token per sequence against the static cache, fused with RoPE of the new
 * q/k and the cache append (same reference lines as above + flash-attn decode, modeling_llama.py:540-566).
 *   qkv [B, (Hq+2Hkv)*D] raw projections of the new token; pos (device int[B]) = tokens already cached.
 *   ws: fp32 workspace of srgpt_decode_attn_ws_floats(B,Hq,D) floats: per-split partials followed by one arrival ticket per
 *   (sequence, kv head).  The tickets must be ZERO before the first launch (zero the workspace once when it is allocated);
 *   every launch leaves them zero again.  One kernel: the split that arrives last merges the partials.  out [B, Hq*D].
 *   The key ranges of the splits depend on pos only -- for caches of up to 4096 positions (2048 from two sequences up on the VALU
 *   kernel) the output bits do NOT depend on max_pos (a pooled cache may be larger than the request needs); larger caches change
 *   the split count.  bf16 with D = 128 runs on the matrix pipe (fixed 64-key ranges), everything else on the VALU kernel. */
int64_t srgpt_decode_attn_ws_floats(int B, int Hq, int D);
int srgpt_decode_attention(const void* qkv, void* kcache, void* vcache, const int* pos, const void* cos_tab,
                           const void* sin_tab, void* out, float* ws, int B, int Hq, int Hkv, int D,
                           int max_pos, int dtype, srgpt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Region extractor (SpatialRGPT specific).
 * srgpt_region_pool: MaskPooling.forward (region_extractor/base_extractor.py:32-84) for ONE image:
 *   bilinear resample (align_corners=False, no antialias, scale_factor semantics of F.interpolate) of the
 *   M masks [M, mh, mw] (dtype) to the fw x fw feature grid, cast to dtype, L1-normalise
 *   (sum + 1e-8), weighted reduce over feat [fw*fw, C] -> out [M, C].  Rounding points follow SURVEY 9.2.
 *   rscale_h/w = (float)(1.0 / scale_factor) as ATen's upsample kernel receives it (host computes
 *   scale_factor = sqrt(L / (mh*mw)) in double like base_extractor.py:53-54); M <= 16 per call.
 *   mask_dtype: storage type of `masks` (SRGPT_F32 or SRGPT_BF16) -- the reference does mask.float().
 *   ws: fp32 workspace of srgpt_region_pool_ws_floats(M, fw, C) floats (resampled masks, per-slab sums and partials, arrival
 *   tickets -- it need not be initialised: the first of the two launches re-arms the tickets).  Two launches, no float atomics:
 *   every sum runs in a fixed order, so the result does not depend on scheduling.
 * srgpt_avgpool: AdaptiveAvgPool2d(out_w) on a channels-last map (base_extractor.py:123,145).
 * srgpt_s2d: DownSampleBlock (multimodal_projector/base_projector.py:32-52) -- zero-pad to even, 2x2
 *   space-to-depth in the reference's (column-major block) order; [n, g*g, C] -> [n, ceil(g/2)^2, 4C].
 * srgpt_im2col: patch extraction for the SigLIP patch-embed conv (Conv2d k=s=patch, 'valid'):
 *   images [n,3,S,S] -> [n*g*g, kp] rows ordered (c, ky, kx), zero-padded from 3*p*p to kp columns.
 * --------------------------------------------------------------------------------------------- */
int64_t srgpt_region_pool_ws_floats(int M, int fw, int C);
int srgpt_region_pool(const void* feat, const void* masks, void* out, float* ws, int M, int mh, int mw,
                      int fw, int C, float rscale_h, float rscale_w, int mask_dtype, int dtype,
                      srgpt_stream_t stream);
/* The same pooling straight from the caller's RAW uint8 masks [M, rh, rw]: ys[mh] / xs[mw] (device int32) are the
 * cv2.INTER_NEAREST source row / column of every row / column of the mh x mw processor-size mask the reference builds on the host
 * (mm_utils.py:477-532); the kernel's bilinear taps read through them, so nearest resize, float(uint8) and the resample are one
 * pass over the raw bytes.  Bit-identical to srgpt_mask_resize_nearest followed by srgpt_region_pool. */
int srgpt_region_pool_u8(const void* feat, const void* masks_u8, const int* ys, const int* xs, void* out, float* ws, int M,
                         int rh, int rw, int mh, int mw, int fw, int C, float rscale_h, float rscale_w, int dtype,
                         srgpt_stream_t stream);
int srgpt_avgpool(const void* x, void* y, int n_img, int in_w, int out_w, int C, int dtype,
                  srgpt_stream_t stream);
int srgpt_s2d(const void* x, void* y, int n_img, int g, int C, int dtype, srgpt_stream_t stream);
int srgpt_im2col(const void* images, void* out, int n_img, int S, int patch, int kp, int dtype,
                 srgpt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Token stream (llava/model/llava_arch.py:434-539).
 * srgpt_embed_rows: out[i,:] = table[ids[i],:]   (embed_tokens; ids are int64, -200 already mapped to 0)
 * srgpt_scatter_rows: dst[idx[i],:] = src[src_idx ? src_idx[i] : i,:]  (image / <mask> / <depth> row
 *   placement; int32 indices; idx[i] < 0 skips the row)
 * srgpt_silu_mul: out = silu(gate) * up over [rows, inter] from a packed [rows, 2*inter] buffer
 * srgpt_argmax: ids_out[b] = argmax_v logits[b, v] (fp32 logits; first max wins like torch.argmax)
 * --------------------------------------------------------------------------------------------- */
int srgpt_embed_rows(const void* table, const int64_t* ids, void* out, int n, int cols, int dtype,
                     srgpt_stream_t stream);
/* ABI 8: the whole splice of prepare_inputs_labels_for_multimodal (llava_arch.py:420-611) on the device, ids never leave it.
 * srgpt_splice_plan (one block per prompt): from ids [B, P] int64 (+ optional attention mask [B, P] bytes) builds, per prompt, the
 *   source of every output row -- desc [B][Tcap][2] = {kind | src << 2, prompt position or -1}; kind 0: embedding row `src` (a text
 *   id), 1: image_features row, 2: mask-embedding row, 3: depth-embedding row -- following the reference rule by rule: masked-out
 *   positions dropped (:434-447), each -200 sentinel replaced by the nimg_feat rows of the NEXT image over the batch (:453-469,
 *   :506-511), the <mask> / <depth> ids of a prompt that owns images replaced IN ORDER by the region embeddings of the prompt's first
 *   image (:470-505; img_info[i] = {rows of image i's region embeddings or -1 if it has none, their first row in the concatenated
 *   embedding matrix}), the prompt cut at max_len (> 0; :541-547).  stats [B][SRGPT_SPLICE_STATS] ints = {length, length before the
 *   cut, sentinels, <mask> ids, <depth> ids, first image index, min / max id sent to the embedding table}: what the host reads back
 *   (the batch's T = max length, and the reference's error / warning conditions).  Tcap >= P + n_images_total * (nimg_feat - 1).
 *   scratch: srgpt_splice_scratch_ints(B, n_images_total) ints.
 * srgpt_splice_gather (one block per output row): out [B, T, cols] = the described rows, prompts shorter than T padded with zero
 *   rows on the right (left_pad: on the left, :549-611); labels_out [B, T] (optional) = labels [B, P] at the row's prompt position,
 *   ignore_index on image rows and padding; attn_mask_out [B, T] bytes (optional) = 1 on real rows. */
#define SRGPT_SPLICE_STATS 8
int64_t srgpt_splice_scratch_ints(int B, int n_images_total);
int srgpt_splice_plan(const int64_t* ids, const unsigned char* attn_mask, int B, int P, int nimg_feat, int n_images_total,
                      const int* img_info, int use_masks, int use_depths, int64_t mask_id, int64_t depth_id, int max_len, int Tcap,
                      int* desc, int* stats, int* scratch, srgpt_stream_t stream);
int srgpt_splice_gather(const int* desc, const int* stats, int B, int Tcap, int T, int left_pad, int cols, int dtype,
                        const void* embed, const void* image_features, const void* mask_embeds, const void* depth_embeds,
                        const int64_t* labels, int P, int64_t ignore_index, void* out, int64_t* labels_out,
                        unsigned char* attn_mask_out, srgpt_stream_t stream);
/* ABI 9: beam search's cache permutation on the device (HF `_reorder_cache`): kcache / vcache [layers, batch * num_beams, kv_heads,
 * max_pos, head_dim]; row r of the first `live` positions of every layer := row beam_idx[r] (int64, device; an index never leaves
 * its batch item's num_beams rows).  In place, one launch; 2 ... 8 beams (else SRGPT_ERR_UNSUPPORTED: permute on the caller's side). */
int srgpt_kv_beam_reorder(void* kcache, void* vcache, const int64_t* beam_idx, int layers, int batch, int num_beams, int kv_heads,
                          int max_pos, int head_dim, int live, int dtype, srgpt_stream_t stream);
int srgpt_scatter_rows(const void* src, const int* src_idx, const int* idx, void* dst, int n, int cols,
                       int dtype, srgpt_stream_t stream);
int srgpt_silu_mul(const void* gu, void* out, int rows, int inter, int dtype, srgpt_stream_t stream);
int srgpt_argmax(const float* logits, int64_t* ids_out, int B, int V, srgpt_stream_t stream);
/* Causal-LM loss of LlamaForCausalLM.forward(labels=...) (modeling_llama.py:1047-1058) over already SHIFTED operands:
 * logits fp32 [rows, V] (positions 0..T-2 of every sequence), labels int64 [rows] (positions 1..T-1); rows whose label is
 * ignore_index (-100) do not count.  row_loss: scratch [rows] fp32; out[0] = mean loss (nan if no target), out[1] = targets. */
int srgpt_cross_entropy(const float* logits, const int64_t* labels, float* row_loss, float* out, int rows, int V,
                        int64_t ignore_index, srgpt_stream_t stream);


/* Sampling parameters of the decode step (ABI 6): a block in DEVICE memory, read by the kernels of every step -- one captured graph
 * serves every setting; the host rewrites the block between requests (hipMemcpyAsync on the decode stream).  The warper chain is
 * HF GenerationMixin.sample's, as the reference's interactive callers reach it (demo/gradio_web_server_multi.py:202-213,
 * llava/eval/model_vqa.py:66-80): scores = logits / temperature -> TopKLogitsWarper (everything below the k-th largest score is
 * removed, ties at it kept) -> TopPLogitsWarper (ascending cumulative softmax; entries whose cumulative mass is <= 1 - top_p are
 * removed, never the largest) -> one categorical draw from the softmax of what is left.
 *   top_k 1 .. 64; top_k = 0: no top-k filter -- then top_p must be >= 1 (pure temperature sampling, drawn by Gumbel-max over the
 *   whole vocabulary).  Other settings (top_k > 64; top_p < 1 without top_k) are not served on the device.
 *   Randomness: Philox4x32-10, key = seed, counter = (counter, sequence, ...); every step advances `counter` by one. */
typedef struct {
  float temperature;   /* > 0 */
  int top_k;
  float top_p;         /* (0, 1]; >= 1 switches the top-p filter off */
  float top_p_rm;      /* (float)(1.0 - (double)top_p): the removal threshold exactly as HF's comparison sees it */
  uint64_t seed;
  uint64_t counter;
  int* kept_out;       /* parity hook (device, may be NULL): int[batch][257] = the kept-set size, then its token ids best first */
} srgpt_sampling;

/* Stand-alone draw over fp32 logits [B, V] with the parameter block above (what the decode step runs after its lm_head):
 * tok_out[b] = the drawn id; advances sp->counter.  ws = srgpt_sample_ws_bytes(B) bytes of scratch.  V <= 262144. */
int64_t srgpt_sample_ws_bytes(int B);
int srgpt_sample(const float* logits, srgpt_sampling* sp, int64_t* tok_out, void* ws, int B, int V, srgpt_stream_t stream);
/* ABI 8: the parameter block lives on the device, so settings outside the served range (top_k > 64; top_p < 1 with top_k = 0) cannot
 * be refused at the call: the kernels raise a bit in the workspace's error word instead of drawing from a silently clamped
 * distribution.  srgpt_sample CLEARS that word when it starts, so srgpt_sample_status -- which synchronises `stream` and returns
 * SRGPT_ERR_UNSUPPORTED for the bit (SRGPT_ERR_STATE when more than 256 entries tied at the top-k threshold) -- reports the MOST RECENT
 * srgpt_sample call on that workspace only: check after each draw whose settings may be out of range.  Inside the decode step the
 * word is the state's own and is never cleared once raised (sticky); srgpt_llm_decode_sync_state reports it. */
int srgpt_sample_status(const void* ws, int B, srgpt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Composite: vision tower (VisionTower.forward, multimodal_encoder/vision_encoder.py:115-132 over HF
 * SiglipVisionModel; returns hidden_states[select_layer], i.e. runs `n_layers_run` encoder layers).
 * All weights device pointers in `dtype`; per-layer arrays are HOST arrays of device pointers.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype, hidden, inter, heads, n_layers_run, image_size, patch, kp; /* kp: padded 3*p*p */
  int inter_pad;        /* row length of w2 and of the MLP activation buffer: inter rounded up to a multiple of 64, zero filled
                           (4304 -> 4352 for SigLIP-so400m) so that fc2's K dimension tiles exactly */
  int act;              /* MLP activation: SRGPT_ACT_GELU_TANH (SigLIP) or SRGPT_ACT_QUICK_GELU (CLIP) */
  float eps;
  const void* patch_w;  /* [hidden, kp] (conv weight flattened (c,ky,kx), zero padded) */
  const void* patch_b;  /* [hidden] or NULL (CLIP's patch conv has no bias) */
  const void* pos_emb;  /* [tokens, hidden], tokens = grid*grid (+1 with a class token) */
  const void* cls_emb;  /* [hidden] class embedding prepended to every image (CLIP) or NULL (SigLIP) */
  const void* pre_ln_w; /* CLIP pre_layrnorm (applied to the embeddings) or NULL */
  const void* pre_ln_b;
  const void* const* ln1_w; const void* const* ln1_b;
  const void* const* wqkv;  const void* const* bqkv;   /* [3*hidden, hidden] rows q;k;v */
  const void* const* wo;    const void* const* bo;
  const void* const* ln2_w; const void* const* ln2_b;
  const void* const* w1;    const void* const* b1;     /* [inter, hidden] */
  const void* const* w2;    const void* const* b2;     /* [hidden, inter_pad], columns >= inter are zero */
} srgpt_vit_weights;

/* workspace bytes for n_img images */
int64_t srgpt_vit_ws_bytes(const srgpt_vit_weights* w, int n_img);
/* rows [n_img*gg, C] patch embeddings -> x [n_img, 1+gg, C]: row 0 = cls + pos[0], row 1+p = patch p + pos[1+p]
 * (HF CLIPVisionEmbeddings.forward: cat(class_embeds, patch_embeds) + position_embedding) */
int srgpt_vit_assemble_cls(const void* patches, const void* cls_emb, const void* pos_emb, void* x, int n_img, int gg,
                           int C, int dtype, srgpt_stream_t stream);
/* images [n_img,3,S,S] -> out [n_img, tokens, hidden] */
int srgpt_vit_forward(const srgpt_vit_weights* w, const void* images, void* out, void* ws, int n_img,
                      srgpt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Request preprocessing on the device (host originals: llava/mm_utils.py:421-474 process_image,
 * :477-532 process_regions, which use PIL, cv2 and the HF image processor).
 * srgpt_image_resize_normalize: uint8 image [H, W, C] (HWC) -> Pillow-exact bicubic resize (two uint8 passes, 22-bit
 *   fixed-point coefficients supplied by the host: per output column/row (first tap, tap count) in *bounds and
 *   *coef [out][k]) -> value * rescale -> (v - mean[c]) / std[c] if do_normalize -> out [C, Hout, Wout] in dtype.
 *   tmp_u8 = H * Wout * C bytes of scratch.  A dimension that is not resized gets the identity table (1 tap, 1<<22).
 * srgpt_mask_resize_nearest: uint8 masks [K, H, W] -> out [K, Hout, Wout] in dtype, out[m,y,x] = src[m, ys[y], xs[x]]
 *   (cv2.INTER_NEAREST index tables from the host: OpenCV resizeNN's published arithmetic, min(floor(dst * (1 / (out / in))), in - 1)
 *   in double -- pinned by tests/golden/cv2_nearest_kat.json; cv2 itself is not installed in the build image).
 * --------------------------------------------------------------------------------------------- */
int srgpt_image_resize_normalize(const void* src_u8, int H, int W, int C, const int* hbounds, const int* hcoef, int hk,
                                 const int* vbounds, const int* vcoef, int vk, int Hout, int Wout, void* tmp_u8,
                                 void* out, const float* mean, const float* stdv, float rescale, int do_normalize,
                                 int dtype, srgpt_stream_t stream);
int srgpt_mask_resize_nearest(const void* src_u8, int K, int H, int W, const int* ys, const int* xs, int Hout, int Wout,
                              void* out, int dtype, srgpt_stream_t stream);
/* ABI 6: process_regions with image_aspect_ratio == "pad" (mm_utils.py:505-531): each uint8 mask [H, W] is centred on a zero square
 * of side max(H, W) (pad_to_square: offsets (side - H) / 2, (side - W) / 2) and the HF processor resizes that one-channel square
 * to Hout x Wout with Pillow's bicubic filter, rescale_factor 1.0, no normalisation.  masks [K, H, W] -> out [K, Hout, Wout] in
 * dtype (values 0 .. 255 as the uint8 result holds them); *bounds / *coef are Pillow's tables for side -> Wout (h) and side -> Hout
 * (v) as for srgpt_image_resize_normalize; tmp_u8 = K * side * Wout bytes.  The padded square is never materialised. */
int srgpt_mask_pad_resize(const void* src_u8, int K, int H, int W, const int* hbounds, const int* hcoef, int hk,
                          const int* vbounds, const int* vcoef, int vk, int Hout, int Wout, void* tmp_u8, void* out,
                          int dtype, srgpt_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Composite: Llama decoder (LlamaForCausalLM.forward, modeling_llama.py:972-1110 inference branch,
 * LlamaModel.forward :824-936, LlamaDecoderLayer :611-684) with a static KV cache.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype, hidden, inter, layers, heads, kv_heads, head_dim, vocab;
  float rms_eps;
  const void* rope_cos;    /* [max_pos, head_dim/2] (see srgpt_rope_kv_append) */
  const void* rope_sin;
  const void* embed;       /* [vocab, hidden] */
  const void* final_norm;  /* [hidden] */
  const void* lm_head;     /* [vocab, hidden] */
  const void* const* attn_norm; /* per layer [hidden] */
  const void* const* wqkv;      /* per layer [(heads+2*kv_heads)*head_dim, hidden]  rows q;k;v */
  const void* const* wo;        /* per layer [hidden, heads*head_dim] */
  const void* const* mlp_norm;  /* per layer [hidden] */
  const void* const* wgu;       /* per layer [2*inter, hidden]  rows gate;up */
  const void* const* wdown;     /* per layer [hidden, inter] */
  /* Optional fp8 (OCP e4m3fn) copies + per-row fp32 scales of the five streamed matrices.  When wqkv8 != NULL the DECODE
   * step streams these through srgpt_gemv_w8 (half the bytes per token) and prefill multiplies them with srgpt_gemm_w8; the
   * dtype matrices above (wqkv / wo / wgu / wdown / lm_head) are then never read and may be NULL (per-layer arrays of NULLs):
   * the fp8 bytes are the only copy of those weights in HBM. */
  const void* lm_head8;
  const float* lm_head_scale;
  const void* const* wqkv8;
  const float* const* wqkv_scale;
  const void* const* wo8;
  const float* const* wo_scale;
  const void* const* wgu8;
  const float* const* wgu_scale;
  const void* const* wdown8;
  const float* const* wdown_scale;
  /* fp8 weights only.  0: W8A16 everywhere (exact weights x bf16 activations -- the default, what the parity tests pin).
   * 1: prefill quantises each GEMM's input per token (srgpt_quant_rows_e4m3) and multiplies on the fp8 matrix pipe
   * (srgpt_gemm_w8a8); decode and the all-position lm_head stay W8A16.  Needs hidden, inter and heads * head_dim % 128 == 0. */
  int fp8_act;
  /* ABI 9: optional PACKED copies (srgpt_pack_decode_weights) of the fp8 layer matrices for the batched decode step (2+ rows per GPU);
   * NULL (array or entry): that product streams the row-major copy.  pk_rows_*: rows per granule of the matrix' packed copies. */
  const void* const* wqkv8p;
  const void* const* wo8p;
  const void* const* wgu8p;
  const void* const* wdown8p;
  int pk_rows_qkv, pk_rows_o, pk_rows_gu, pk_rows_down;
} srgpt_llm_weights;

typedef struct {
  int batch, max_pos;
  void* kcache;   /* [layers, batch, kv_heads, max_pos, head_dim] */
  void* vcache;
  int* pos;       /* device int[batch]: tokens already in the cache */
  int64_t* tok;   /* device int64[batch]: next input token id (decode) */
  int64_t* out_ids; /* device int64[batch, max_new]: generated ids, column = *step */
  int* step;      /* device int[1]: decode step counter */
  int max_new;
  int ws_tokens;  /* max prompt tokens per sequence the workspace was sized for */
  void* ws;       /* workspace, srgpt_llm_ws_bytes(w, batch, ws_tokens).  It holds the decode attention's arrival tickets, which
                   * must be zero before a decode step: every srgpt_llm_prefill* zeroes them (so a hipMalloc'ed, never-zeroed
                   * workspace is fine as long as a prefill precedes the first step, which it must anyway) */
  float* logits;  /* [batch, vocab] fp32 (last position) */
  srgpt_sampling* sampling; /* ABI 6: DEVICE pointer or NULL.  NULL: greedy (argmax).  Set: srgpt_llm_sample_first / srgpt_llm_decode_step
                             * and the graphs captured from this state DRAW the next token (see srgpt_sampling) */
} srgpt_llm_state;

int64_t srgpt_llm_ws_bytes(const srgpt_llm_weights* w, int batch, int max_tokens);
/* Prefill: inputs_embeds [B, T, hidden] (already spliced), all rows valid (equal-length prompts).
 * Fills the cache, sets pos[b] = T, writes last-position logits to st->logits; if all_logits != NULL
 * also writes fp32 logits for every position [B, T, vocab]; if hidden_out != NULL writes the
 * (layers+1) pre-norm hidden states [layers+1, B, T, hidden] (parity hook). */
int srgpt_llm_prefill(const srgpt_llm_weights* w, srgpt_llm_state* st, const void* inputs_embeds, int T,
                      float* all_logits, void* hidden_out, srgpt_stream_t stream);
/* Ragged batch (prompts of different lengths in one call -- the reference right-pads and hands flash-attn the unpadded
 * rows, llava_arch.py:549-611 + modeling_llama.py:540-608): inputs_embeds [batch, T, hidden] RIGHT-padded, lens[b] =
 * valid rows of sequence b (device int32).  Causal attention never looks right, so valid rows are exact whatever the
 * padding holds; logits come from row lens[b]-1 and decoding continues at position lens[b] per sequence. */
int srgpt_llm_prefill_ragged(const srgpt_llm_weights* w, srgpt_llm_state* st, const void* inputs_embeds, int T,
                             const int* lens, float* all_logits, void* hidden_out, srgpt_stream_t stream);
/* One greedy decode step, entirely device-side: embeds st->tok, runs the layers against the cache,
 * argmax -> st->tok, st->out_ids[:, *step], ++pos, ++*step.  No host sync. */
int srgpt_llm_decode_step(const srgpt_llm_weights* w, srgpt_llm_state* st, srgpt_stream_t stream);
/* Health check of the decode steps since the last prefill (synchronises `stream` once): 0 when every arrival ticket of the decode
 * attention in st->ws is re-armed (zero) and no captured step met a caller-written token id outside the embedding table;
 * SRGPT_ERR_STATE otherwise (srgpt_last_error says which). */
int srgpt_llm_decode_sync_state(const srgpt_llm_weights* w, const srgpt_llm_state* st, srgpt_stream_t stream);
/* first token after prefill: argmax(st->logits) -> tok / out_ids[:,0] / step=1 */
int srgpt_llm_sample_first(const srgpt_llm_weights* w, srgpt_llm_state* st, srgpt_stream_t stream);

/* hipGraph capture of one decode step (replayed per token; removes per-launch host cost).  A replay continues from st->tok: normally
 * the token the step before it (srgpt_llm_decode_step, srgpt_llm_sample_first or the previous replay) picked, whose embedding row
 * that step already left in the residual-stream workspace; the captured step's first node compares st->tok with the token that row
 * belongs to and re-embeds when the caller wrote a token of its own into st->tok between replays; an id outside the table cannot be
 * embedded -- the step then runs on the previous row and raises a sticky error that srgpt_llm_decode_sync_state reports. */
typedef struct srgpt_graph srgpt_graph;
int srgpt_llm_decode_graph_create(const srgpt_llm_weights* w, srgpt_llm_state* st, srgpt_stream_t stream,
                                  srgpt_graph** out);
int srgpt_graph_launch(srgpt_graph* g, int times, srgpt_stream_t stream);
int srgpt_graph_destroy(srgpt_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* SRGPT_H_ */
