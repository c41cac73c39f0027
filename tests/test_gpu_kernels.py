"""GPU parity tests, one per kernel family, through the C ABI (spatialrgpt_amd.ops -> libsrgpt_hip.so).
Checker = the oracle's functions / plain fp32 torch on the CPU.  Tolerances: fp32 path 1e-4 (accumulation
order), bf16 path 2e-2 relative to the tensor scale (one bf16 ulp = 0.4 %, chained roundings)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, load_kat

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from spatialrgpt_amd import _lib, ops
    return ops, _lib


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _tol(ref, dtype, k=1.0):
    s = float(ref.float().abs().max()) + 1e-6
    return (2e-2 if dtype == torch.bfloat16 else 2e-5) * s * k


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(130, 200, 136), (259, 512, 1024), (64, 64, 64), (729, 1152, 592), (100, 264, 4304),
                                   (8, 4096, 1152), (300, 1000, 72),
                                   # 96-row tiles (1 x 4 waves): single-buffered un-split grid (>= 2 blocks per CU), double-buffered
                                   # split-K, ragged last row tile (259 = 2 x 96 + 67), ragged N, K tail through registers
                                   (259, 28672, 512), (259, 4096, 4096), (1458, 4304, 1152), (97, 130, 200), (288, 640, 64),
                                   # 257 .. 383 rows (prompts a little longer than the benchmark's): split-K 5 / 8 / none, ragged N
                                   (259, 6144, 4096), (300, 1000, 1024), (383, 4096, 512), (257, 128, 2048), (352, 264, 4096),
                                   (320, 4096, 14336)])
def test_gemm_plain(dtype, M, N, K):
    _gemm_case(dtype, M, N, K)


# The whole-M prefill kernel (gemm288.hip: 225 .. 272 rows, N * K >= 40 Mi elements): split-K 3 with 22 / 21 / 21 K tiles, an odd
# tile count (65: the three-stage ring wraps mid-split), split-K 8 over K = 14336, no tail rows (M = 225), one tail row + ragged N.
@pytest.mark.parametrize("M,N,K", [(259, 12288, 4096), (272, 10240, 4160), (259, 4096, 14336), (225, 28672, 1536), (257, 45000, 1024),
                                   (259, 28672, 4096)])
def test_gemm_whole_m_tile(M, N, K):
    _gemm_case(torch.bfloat16, M, N, K)


def _gemm_case(dtype, M, N, K):
    ops, L = _ops()
    if dtype == torch.float32 and K % 4:
        pytest.skip("K % 4")
    a, w = _rand((M, K), dtype, 1), _rand((N, K), dtype, 2, 0.05)
    b, r = _rand((N,), dtype, 3), _rand((M, N), dtype, 4)
    ref = a.float() @ w.float().T
    out = ops.gemm(a.to(DEV), w.to(DEV))
    assert_close(out, ref, _tol(ref, dtype), 0, "gemm")
    # asymmetric epilogue: bias, tanh-gelu, residual
    ref2 = F.gelu((ref + b.float()).to(dtype).float(), approximate="tanh").to(dtype).float() + r.float()
    out2 = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV), act=L.ACT_GELU_TANH)
    assert_close(out2, ref2, _tol(ref2, dtype), 0, "gemm+bias+gelu+res")
    out3 = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), act=L.ACT_GELU_ERF, out_f32=True)
    assert out3.dtype == torch.float32
    assert_close(out3, F.gelu((ref + b.float()).to(dtype).float()), _tol(ref, dtype), 0, "gemm f32 out")


# o_proj / down_proj (+ bias) + residual + the next norm in one call: the norm inside the split-K reduction (small tiles split 2, the
# whole-M kernel split 8, a 256-row-tile shape, the ViT's out_proj / fc2 with LayerNorm), the wide-row variant (N = 8192), and the
# fallbacks (no split; N > 8192; fp32)
@pytest.mark.parametrize("dtype,M,N,K,layer", [(torch.bfloat16, 259, 4096, 4096, False), (torch.bfloat16, 259, 4096, 14336, False),
                                               (torch.bfloat16, 707, 4096, 4096, False), (torch.bfloat16, 259, 8192, 2048, True),
                                               (torch.bfloat16, 300, 2560, 6912, False), (torch.bfloat16, 259, 28672, 512, False),
                                               (torch.bfloat16, 64, 16384, 256, True), (torch.float32, 130, 512, 256, False),
                                               (torch.bfloat16, 1458, 1152, 1152, True), (torch.bfloat16, 1458, 1152, 4304, True),
                                               (torch.bfloat16, 1154, 1024, 4096, True), (torch.float32, 100, 256, 128, True)])
def test_gemm_norm_equals_the_two_launches(dtype, M, N, K, layer):
    ops, L = _ops()
    a, w = _rand((M, K), dtype, 11), _rand((N, K), dtype, 12, 0.05)
    r, g = _rand((M, N), dtype, 13), (1 + 0.1 * _rand((N,), torch.float32, 14)).to(dtype)
    bias = _rand((N,), dtype, 15) if layer else None
    nb = _rand((N,), dtype, 16) if layer else None
    a, w, r, g = a.to(DEV), w.to(DEV), r.to(DEV), g.to(DEV)
    bias, nb = (bias.to(DEV), nb.to(DEV)) if layer else (None, None)
    c_ref = ops.gemm(a, w, bias, r)
    y_ref = ops.layernorm(c_ref, g, nb, 1e-6) if layer else ops.rmsnorm(c_ref, g, 1e-5)
    eps = 1e-6 if layer else 1e-5
    c, y = ops.gemm_norm(a, w, r, g, eps, bias=bias, norm_b=nb)
    assert torch.equal(c, c_ref) and torch.equal(y, y_ref)
    x = r.clone()  # in place on the residual stream and the normalised rows over the product's input, as the layer loops call it
    if N == K:
        a2 = a.clone()
        c2, y2 = ops.gemm_norm(a2, w, x, g, eps, bias=bias, norm_b=nb, out=x, y=a2)
        assert y2.data_ptr() == a2.data_ptr()
    else:
        c2, y2 = ops.gemm_norm(a, w, x, g, eps, bias=bias, norm_b=nb, out=x)
    assert c2.data_ptr() == x.data_ptr() and torch.equal(x, c_ref) and torch.equal(y2, y_ref)
    ref = (a.float() @ w.float().T + (bias.float() if layer else 0)).to(dtype).float() + r.float()
    assert_close(c, ref, _tol(ref, dtype), 0, "gemm_norm C")


# the prefill's q/k/v projection with RoPE + cache append inside the split-K reduction: Llama-3 geometry (whole-M kernel, 5 splits),
# Llama-2-7B (MHA: N = 12288), a 2 x 100-row batch with a position offset (small tiles), 707 rows, and the fallbacks (no split; fp32)
@pytest.mark.parametrize("dtype,B,T,Hq,Hkv,D,K,off", [(torch.bfloat16, 1, 259, 32, 8, 128, 4096, False),
                                                      (torch.bfloat16, 1, 259, 32, 32, 128, 4096, False),
                                                      (torch.bfloat16, 2, 100, 32, 8, 128, 4096, True),
                                                      (torch.bfloat16, 1, 707, 32, 8, 128, 4096, False),
                                                      (torch.bfloat16, 1, 300, 20, 20, 128, 2560, False),
                                                      (torch.bfloat16, 3, 40, 4, 2, 64, 128, True),
                                                      (torch.float32, 2, 30, 4, 2, 32, 64, True)])
def test_gemm_rope_kv_append_equals_the_two_launches(dtype, B, T, Hq, Hkv, D, K, off):
    ops, L = _ops()
    from spatialrgpt_amd.config import SrgptConfig as PC
    from spatialrgpt_amd.weights import rope_tables
    max_pos = 1024
    N = (Hq + 2 * Hkv) * D
    cos_t, sin_t = rope_tables(PC(hidden=Hq * D, heads=Hq, kv_heads=Hkv, rope_theta=500000.0), max_pos, dtype, DEV)
    a, w = _rand((B * T, K), dtype, 21).to(DEV), _rand((N, K), dtype, 22, 0.05).to(DEV)
    pos0 = torch.tensor([7 * (b + 1) for b in range(B)], dtype=torch.int32, device=DEV) if off else None
    kc_ref = torch.zeros((B, Hkv, max_pos, D), device=DEV, dtype=dtype)
    vc_ref = torch.zeros_like(kc_ref)
    qkv_ref = ops.gemm(a, w)
    ops.rope_kv_append(qkv_ref, kc_ref, vc_ref, cos_t, sin_t, B, T, Hq, Hkv, D, pos0=pos0)
    kc, vc = torch.zeros_like(kc_ref), torch.zeros_like(kc_ref)
    qkv = ops.gemm_rope_kv_append(a, w, kc, vc, cos_t, sin_t, B, T, Hq, Hkv, D, pos0=pos0)
    assert torch.equal(qkv[:, :Hq * D], qkv_ref[:, :Hq * D]), "rotated q"
    assert torch.equal(kc, kc_ref) and torch.equal(vc, vc_ref), "caches"
    assert torch.equal(qkv, qkv_ref), "the k / v columns of the q/k/v buffer"


# LlamaMLP's gate / up products + SiLU * up in one call: the whole-M kernel with the activation as its epilogue (Llama-3, Llama-2-7B
# and a no-tail-row shape), and the shapes that fall back to the two launches (too few columns; other row counts; fp32)
@pytest.mark.parametrize("dtype,M,I,K", [(torch.bfloat16, 259, 14336, 4096), (torch.bfloat16, 272, 11008, 4096),
                                         (torch.bfloat16, 240, 8192, 1024), (torch.bfloat16, 259, 6912, 2560),
                                         (torch.bfloat16, 1036, 2048, 512), (torch.bfloat16, 100, 512, 256),
                                         (torch.float32, 64, 256, 128)])
def test_gemm_swiglu_equals_the_two_launches(dtype, M, I, K):
    ops, L = _ops()
    a, w = _rand((M, K), dtype, 31).to(DEV), _rand((2 * I, K), dtype, 32, 0.05).to(DEV)
    ref = ops.silu_mul(ops.gemm(a, w))
    out = ops.gemm_swiglu(a, w)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_strided_a_and_row_modulo_residual(dtype):
    ops, L = _ops()
    big = _rand((96, 3 * 64), dtype, 5)
    a = big[:, 64:128]  # row stride 192, K = 64
    w, pos = _rand((80, 64), dtype, 6, 0.1), _rand((32, 80), dtype, 7)
    ref = a.float() @ w.float().T
    ref = ref.to(dtype).float() + pos.float().repeat(3, 1)
    out = ops.gemm(big.to(DEV)[:, 64:128], w.to(DEV), residual=pos.to(DEV), res_mod=32)
    assert_close(out, ref, _tol(ref, dtype), 0, "strided A + residual row modulo")
    # the same at 300 rows (96-row tiles, split K)
    big = _rand((300, 3 * 1024), dtype, 8)
    w2, pos2 = _rand((200, 1024), dtype, 9, 0.05), _rand((100, 200), dtype, 10)
    ref = (big[:, 1024:2048].float() @ w2.float().T).to(dtype).float() + pos2.float().repeat(3, 1)
    out = ops.gemm(big.to(DEV)[:, 1024:2048], w2.to(DEV), residual=pos2.to(DEV), res_mod=100)
    assert_close(out, ref, _tol(ref, dtype), 0, "strided A + residual row modulo, 300 rows")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n_img,g,C", [(1, 27, 64), (2, 6, 32), (1, 24, 48)])
def test_gemm_deconv2x(dtype, n_img, g, C):
    """ConvTranspose2d(C,C,2,2) as GEMM + pixel shuffle, channels-last."""
    ops, L = _ops()
    x = _rand((n_img, g * g, C), dtype, 8)
    wt, b = _rand((C, C, 2, 2), dtype, 9, 0.1), _rand((C,), dtype, 10)
    ref = F.conv_transpose2d(x.float().reshape(n_img, g, g, C).permute(0, 3, 1, 2), wt.float(), b.float(), stride=2)
    ref = F.gelu(ref.to(dtype).float()).flatten(2).transpose(1, 2)  # N (H W) C
    w2 = wt.permute(2, 3, 1, 0).reshape(4 * C, C).contiguous()
    out = ops.gemm(x.reshape(-1, C).to(DEV), w2.to(DEV), b.to(DEV), act=L.ACT_GELU_ERF, bias_mod=C,
                   out_mode=L.OUT_DECONV2X, gw=g, out_shape=(n_img * 4 * g * g, C))
    assert_close(out.reshape(n_img, 4 * g * g, C), ref, _tol(ref, dtype), 0, "deconv2x")


@pytest.mark.parametrize("M,N,K,what", [
    (3000, 4304, 1152, "ragged M and N, 204 tiles, no split (ViT fc1 shape at ~2 images x 2)"),
    (2048, 4096, 1024, "128 tiles -> split-K 2 (deterministic slabs)"),
    (4096, 4096, 512, "exact tiles, 8 K tiles"),
    (2072, 28672, 4096, "batched prefill gate/up (8 requests x 259 rows)"),
    (11664, 1152, 4352, "batched ViT fc2 with K padded to a multiple of 64"),
    (2305, 3456, 256, "minimum K (4 tiles): prologue/tail paths only")])
def test_gemm_256_tile_kernel(M, N, K, what):
    """gemm256.hip (256 x 256 x 64 tiles, 8 waves, counted LDS-DMA waits): every shape here takes that kernel.  Checked
    against fp32 torch on the device with an ASYMMETRIC epilogue (bias + tanh-GELU + residual) and a plain one, twice (the
    second run must be bit-identical: a DMA / barrier race would show as run-to-run differences)."""
    ops, L = _ops()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    b = torch.randn((N,), generator=g, device=DEV).to(torch.bfloat16)
    r = torch.randn((M, N), generator=g, device=DEV).to(torch.bfloat16)
    ref = a.float() @ w.float().T
    out = ops.gemm(a, w)
    assert_close(out, ref, _tol(ref, torch.bfloat16), 0, f"gemm256 plain: {what}")
    ref2 = F.gelu((ref + b.float()).to(torch.bfloat16).float(), approximate="tanh").to(torch.bfloat16).float() + r.float()
    out2 = ops.gemm(a, w, b, r, act=L.ACT_GELU_TANH)
    assert_close(out2, ref2, _tol(ref2, torch.bfloat16), 0, f"gemm256 bias+gelu+res: {what}")
    for _ in range(3):
        assert torch.equal(ops.gemm(a, w, b, r, act=L.ACT_GELU_TANH), out2), "run-to-run difference"
    # rows of A with a stride (lda > K), as the ViT workspace hands them over
    big = torch.randn((M, K + 64), generator=g, device=DEV).to(torch.bfloat16)
    out4 = ops.gemm(big[:, :K], w)
    assert_close(out4, big[:, :K].float() @ w.float().T, _tol(ref, torch.bfloat16), 0, f"gemm256 strided A: {what}")


def test_gemm_256_deconv_epilogue():
    """the second deconv of the refinement module at 4 images: [4*54*54, 1152] x [4608, 1152] with the pixel-shuffle epilogue"""
    ops, L = _ops()
    n_img, g, C = 4, 54, 1152
    x = _rand((n_img, g * g, C), torch.bfloat16, 8).to(DEV)
    wt, b = _rand((C, C, 2, 2), torch.bfloat16, 9, 0.03).to(DEV), _rand((C,), torch.bfloat16, 10).to(DEV)
    ref = F.conv_transpose2d(x.float().reshape(n_img, g, g, C).permute(0, 3, 1, 2), wt.float(), b.float(), stride=2)
    ref = F.gelu(ref.to(torch.bfloat16).float()).flatten(2).transpose(1, 2)
    w2 = wt.permute(2, 3, 1, 0).reshape(4 * C, C).contiguous()
    out = ops.gemm(x.reshape(-1, C), w2, b, act=L.ACT_GELU_ERF, bias_mod=C, out_mode=L.OUT_DECONV2X, gw=g,
                   out_shape=(n_img * 4 * g * g, C))
    assert_close(out.reshape(n_img, 4 * g * g, C), ref, _tol(ref, torch.bfloat16), 0, "gemm256 deconv2x")


@pytest.mark.parametrize("M,N,K", [(259, 6144, 4096), (2072, 4096, 4096), (300, 1000, 512), (64, 128258 // 8, 256), (1036, 28672 // 4, 4096),
                                   (210, 64, 160), (37, 130, 72)])  # the last two: K % 64 != 0 -> scalar fallback
def test_gemm_w8_equals_gemm_on_dequantised_weights(M, N, K):
    """srgpt_gemm_w8 (fp8 weight bytes staged through LDS, widened to bf16 in front of the MFMA, row scale in the epilogue) ==
    srgpt_gemm on dequant(quant(W)): BIT-identical for the plain product (power-of-two scales commute with the fp32
    accumulation) -- compared at equal K-split so the accumulation order is the same -- and within bf16 tolerance of fp32 torch
    with a bias / SiLU / residual epilogue."""
    ops, L = _ops()
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N)
    a = torch.randn((M, K), generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    b = torch.randn((N,), generator=g, device=DEV).to(torch.bfloat16)
    r = torch.randn((M, N), generator=g, device=DEV).to(torch.bfloat16)
    q8, sc, deq = ops.quantize_fp8_rows(w)
    ref = a.float() @ deq.float().T
    out = ops.gemm_w8(a, q8, sc)
    assert_close(out, ref, _tol(ref, torch.bfloat16), 0, "gemm_w8 vs fp32 torch on the dequantised weights")
    out2 = ops.gemm_w8(a, q8, sc, b, r, act=L.ACT_SILU)
    ref2 = F.silu((ref + b.float()).to(torch.bfloat16).float()).to(torch.bfloat16).float() + r.float()
    assert_close(out2, ref2, _tol(ref2, torch.bfloat16), 0, "gemm_w8 bias + silu + residual")
    for _ in range(2):
        assert torch.equal(ops.gemm_w8(a, q8, sc, b, r, act=L.ACT_SILU), out2), "run-to-run difference"
    lo = ops.gemm_w8(a, q8, sc, out_f32=True)
    assert lo.dtype == torch.float32
    assert_close(lo, ref, _tol(ref, torch.bfloat16), 0, "gemm_w8 fp32 out")


# ------------------------------------------------------------------------------------------------ GEMV
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,N,K", [(1, 1000, 4096), (1, 1001, 4096), (2, 512, 11008), (4, 300, 1024), (1, 128, 64),
                                   (3, 77, 2560),
                                   # > 4 rows: MFMA skinny kernel (bf16, 16 rows per launch) / chunked VALU kernel (fp32);
                                   # ragged N (not a multiple of 16), K tails (not a multiple of 1024 / 256), multi-pass N
                                   (5, 1000, 4096), (8, 4096, 4096), (16, 1001, 2560), (13, 512, 11008), (7, 40, 72),
                                   (20, 300, 1024), (33, 130, 512), (6, 33000, 256)])
def test_gemv_variants(dtype, B, N, K):
    ops, L = _ops()
    x, w = _rand((B, K), dtype, 11), _rand((N, K), dtype, 12, 0.03)
    g, r = (1 + 0.1 * _rand((K,), torch.float32, 13)).to(dtype), _rand((B, N), dtype, 14)
    ref = x.float() @ w.float().T
    assert_close(ops.gemv(x.to(DEV), w.to(DEV)), ref, _tol(ref, dtype), 0, "gemv")
    assert_close(ops.gemv(x.to(DEV), w.to(DEV), residual=r.to(DEV)), ref.to(dtype).float() + r.float(), _tol(ref, dtype, 2), 0, "gemv+res")
    lo = ops.gemv(x.to(DEV), w.to(DEV), out_f32=True)
    assert lo.dtype == torch.float32
    assert_close(lo, ref, _tol(ref, dtype), 0, "gemv f32")
    # fused RMSNorm prologue (LlamaRMSNorm rounding points)
    xf = x.float()
    xn = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dtype)
    xn = (g * xn)
    refn = xn.float() @ w.float().T
    assert_close(ops.gemv(x.to(DEV), w.to(DEV), norm_w=g.to(DEV), eps=1e-5), refn, _tol(refn, dtype), 0, "rmsnorm+gemv")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,N,K", [(1, 1024, 4096), (2, 333, 512), (8, 1024, 4096), (16, 333, 512), (5, 14336, 1024), (19, 100, 264)])
def test_gemv_swiglu(dtype, B, N, K):
    ops, L = _ops()
    x, w = _rand((B, K), dtype, 15), _rand((2 * N, K), dtype, 16, 0.03)
    gate = (x.float() @ w[:N].float().T).to(dtype)
    up = (x.float() @ w[N:].float().T).to(dtype)
    ref = (F.silu(gate.float()).to(dtype).float() * up.float())
    assert_close(ops.gemv(x.to(DEV), w.to(DEV), swiglu=True), ref, _tol(ref, dtype), 0, "swiglu gemv")


def test_fp8_decode_matches_torch_float8():
    """the hardware fp8 -> fp32 conversion (v_cvt_pk_f32_fp8 on gfx950) is OCP e4m3fn: all 256 codes through the W8 kernel
    equal torch.float8_e4m3fn's decode (NaN codes 0x7f / 0xff excluded)."""
    ops, L = _ops()
    codes = torch.arange(256, dtype=torch.uint8)
    codes = codes[(codes & 0x7F) != 0x7F]
    n = codes.numel()
    # W8 row r holds code r at k = 0 and zeros elsewhere; x = e_0 -> out[r] = decode(code r)
    w8 = torch.zeros((n, 64), dtype=torch.uint8)
    w8[:, 0] = codes
    x = torch.zeros((1, 64), dtype=torch.bfloat16)
    x[0, 0] = 1.0
    out = ops.gemv_w8(x.to(DEV), w8.to(DEV), torch.ones(n, dtype=torch.float32, device=DEV), out_f32=True)
    ref = codes.view(torch.float8_e4m3fn).float()
    assert torch.equal(out[0].cpu(), ref)
    # the MFMA skinny kernel (3+ rows) widens with v_cvt_scalef32_pk_bf16_fp8: same table, every code, both staging forms
    for rows in (3, 8, 16):
        xb = torch.zeros((rows, 64), dtype=torch.bfloat16)
        xb[:, 0] = 1.0
        ob = ops.gemv_w8(xb.to(DEV), w8.to(DEV), torch.ones(n, dtype=torch.float32, device=DEV), out_f32=True)
        assert torch.equal(ob.cpu(), ref[None].expand(rows, -1)), rows


@pytest.mark.parametrize("B,N,K", [(1, 1000, 4096), (2, 4096, 4096), (4, 1001, 2560), (8, 512, 11008), (16, 333, 1024),
                                   (20, 130, 264), (1, 33000, 256),
                                   # 3..8 rows, K % 16 == 0: the 512-k fp8 form of the skinny kernel (raw bytes in LDS) -- one block
                                   # per unit with 8 waves (N/16 <= CUs), balanced 4-wave grid, several passes, K tails of a
                                   # 512-slice, a single partial slice; K % 16 != 0 and 9..16 rows keep the 256-k form
                                   (8, 4096, 4096), (5, 6144, 4096), (3, 1024, 4096), (7, 33000, 528), (8, 40, 48), (6, 64, 264),
                                   (8, 14336, 1024), (4, 4096, 14336)])
def test_gemv_w8_variants(B, N, K):
    """W8A16 decode product vs fp32 torch on the dequantised weights (tolerance: bf16 output rounding)."""
    ops, L = _ops()
    dtype = torch.bfloat16
    x, w = _rand((B, K), dtype, 21), _rand((N, K), dtype, 22, 0.03)
    g, r = (1 + 0.1 * _rand((K,), torch.float32, 23)).to(dtype), _rand((B, N), dtype, 24)
    q8, sc, deq = ops.quantize_fp8_rows(w.to(DEV))
    assert q8.dtype == torch.uint8 and sc.dtype == torch.float32 and deq.dtype == dtype
    # the quantiser itself: per-row scale = max|w|/448, codes = round-to-nearest e4m3fn, error <= 2^-4 of the row max
    assert float(((deq.float().cpu() - w.float()).abs() / w.float().abs().amax(1, keepdim=True)).max()) <= 2 ** -4
    wq = (q8.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None])
    ref = x.float() @ wq.T
    xd = x.to(DEV)
    assert_close(ops.gemv_w8(xd, q8, sc), ref, _tol(ref, dtype), 0, "gemv_w8")
    assert_close(ops.gemv_w8(xd, q8, sc, residual=r.to(DEV)), ref.to(dtype).float() + r.float(), _tol(ref, dtype, 2), 0, "gemv_w8+res")
    lo = ops.gemv_w8(xd, q8, sc, out_f32=True)
    assert lo.dtype == torch.float32
    assert_close(lo, ref, _tol(ref, dtype), 0, "gemv_w8 f32")
    xf = x.float()
    xn = g * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dtype)
    refn = xn.float() @ wq.T
    assert_close(ops.gemv_w8(xd, q8, sc, norm_w=g.to(DEV), eps=1e-5), refn, _tol(refn, dtype), 0, "rmsnorm+gemv_w8")
    # consistency with the bf16 path on the dequantised weights (what prefill multiplies)
    assert_close(ops.gemv_w8(xd, q8, sc), ops.gemv(xd, deq).float().cpu(), _tol(ref, dtype, 2), 0, "w8 vs bf16(deq)")


@pytest.mark.parametrize("B,N,K", [(1, 1024, 4096), (8, 333, 512), (16, 100, 264), (8, 14336, 4096), (4, 1000, 2560), (5, 512, 1040)])
def test_gemv_w8_swiglu(B, N, K):
    ops, L = _ops()
    dtype = torch.bfloat16
    x, w = _rand((B, K), dtype, 25), _rand((2 * N, K), dtype, 26, 0.03)
    q8, sc, deq = ops.quantize_fp8_rows(w.to(DEV))
    wq = (q8.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None])
    gate = (x.float() @ wq[:N].T).to(dtype)
    up = (x.float() @ wq[N:].T).to(dtype)
    ref = (F.silu(gate.float()).to(dtype).float() * up.float())
    assert_close(ops.gemv_w8(x.to(DEV), q8, sc, swiglu=True), ref, _tol(ref, dtype), 0, "swiglu gemv_w8")


# ------------------------------------------------------------------------------------------------ row-statistics hand-off (ABI 8)
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,N,K,N2", [(2, 4096, 4096, 6144), (8, 4096, 14336, 28672 // 2), (4, 2560, 6912, 1000), (16, 4096, 4096, 333),
                                      (5, 1000, 1024, 130), (20, 520, 264, 77), (3, 4104, 512, 40),
                                      (17, 576, 1280, 1040)])  # 16 rows + a single-row tail: the tail stays on the MFMA kernel in srgpt_gemv too (found by tests/test_gpu_fuzz_shapes.py)
def test_gemv_rowss_handoff(fp8, B, N, K, N2):
    """srgpt_gemv_rowss: (1) the table a residual product publishes sums, per row, to sum(out^2) of the bf16 rows it wrote -- slots
    past its grid zero; (2) the output itself is srgpt_gemv's, bit for bit; (3) a consumer normalising from that table agrees with
    the consumer computing the statistics itself to within one bf16 rounding of a normalised element (the two reductions associate
    differently), and both sit at the usual distance from fp32 torch; (4) the table is a pure function of the inputs (repeats equal)."""
    ops, L = _ops()
    dtype = torch.bfloat16
    if not ops.gemv_rowss_supported(B, fp8):
        pytest.skip("no hand-off kernel for this batch")
    x, w = _rand((B, K), dtype, 31).to(DEV), _rand((N, K), dtype, 32, 0.03).to(DEV)
    r = _rand((B, N), dtype, 33).to(DEV)
    g = (1 + 0.1 * _rand((N,), torch.float32, 34)).to(dtype).to(DEV)
    w2 = _rand((N2, N), dtype, 35, 0.03).to(DEV)
    if fp8:
        q8, sc, _ = ops.quantize_fp8_rows(w)
        q2, sc2, _ = ops.quantize_fp8_rows(w2)
        kw1, kw2 = dict(w8=q8, wscale=sc), dict(w8=q2, wscale=sc2)
        plain1 = ops.gemv_w8(x, q8, sc, residual=r)
    else:
        kw1, kw2 = dict(w=w), dict(w=w2)
        plain1 = ops.gemv(x, w, residual=r)
    h, table = ops.gemv_rowss(x, residual=r, publish=True, **kw1)
    assert torch.equal(h, plain1)
    assert table.shape == (B, L.ROWSS_STRIDE) and bool(torch.isfinite(table).all())
    want = h.double().pow(2).sum(-1)
    got = table.double().sum(-1)
    assert float(((got - want).abs() / want).max()) < 2e-6, (got, want)
    h2, table2 = ops.gemv_rowss(x, residual=r, publish=True, **kw1)
    assert torch.equal(table, table2) and torch.equal(h, h2)
    # consumer: published statistics vs its own
    own = ops.gemv_w8(h, q2, sc2, norm_w=g, eps=1e-5) if fp8 else ops.gemv(h, w2, norm_w=g, eps=1e-5)
    pub = ops.gemv_rowss(h, norm_w=g, eps=1e-5, rowss_in=table, **kw2)
    hf = h.float().cpu()
    xn = g.cpu().float() * (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dtype).float()
    wq2 = (q2.cpu().view(torch.float8_e4m3fn).float() * sc2.cpu()[:, None]) if fp8 else w2.float().cpu()
    ref = xn.to(dtype).float() @ wq2.T
    assert_close(pub, ref, _tol(ref, dtype), 0, "rowss consumer vs torch")
    assert_close(pub, own.float().cpu(), _tol(ref, dtype, 0.5), 0, "rowss consumer vs own statistics")
    # fp32 logits and SwiGLU consumers take the table too
    lo = ops.gemv_rowss(h, norm_w=g, eps=1e-5, rowss_in=table, out_f32=True, **kw2)
    assert lo.dtype == torch.float32
    assert_close(lo, ref, _tol(ref, dtype), 0, "rowss consumer f32")
    if N2 % 2 == 0:
        half = N2 // 2
        gate, up = (xn.to(dtype).float() @ wq2[:half].T).to(dtype), (xn.to(dtype).float() @ wq2[half:].T).to(dtype)
        refs = F.silu(gate.float()).to(dtype).float() * up.float()
        assert_close(ops.gemv_rowss(h, norm_w=g, eps=1e-5, rowss_in=table, swiglu=True, **kw2), refs, _tol(refs, dtype), 0, "rowss swiglu")


def test_gemv_rowss_rejects_what_it_cannot_do():
    ops, L = _ops()
    x = _rand((1, 256), torch.bfloat16, 1).to(DEV)
    w = _rand((64, 256), torch.bfloat16, 2).to(DEV)
    assert not ops.gemv_rowss_supported(1) and not ops.gemv_rowss_supported(1, True)
    with pytest.raises(NotImplementedError):
        ops.gemv_rowss(x, w=w)
    x4 = _rand((4, 256), torch.bfloat16, 1).to(DEV)
    t = torch.zeros((4, L.ROWSS_STRIDE), device=DEV)
    with pytest.raises(ValueError):  # a table without the RMSNorm it feeds
        ops.gemv_rowss(x4, w=w, rowss_in=t)
    w2 = _rand((128, 256), torch.bfloat16, 2).to(DEV)
    with pytest.raises(ValueError):  # SwiGLU outputs are not residual-stream rows: nothing to publish
        ops.gemv_rowss(x4, w=w2, swiglu=True, publish=True)


# ------------------------------------------------------------------------------------------------ packed decode layout (ABI 9)
def packed_layout_reference(w: torch.Tensor, rows: int) -> torch.Tensor:
    """include/srgpt.h (ABI 9), restated with index arithmetic on the host: out[((n / rows) (K / kblock) + k / kblock) rows 64
    + ((k % 32) / 8 rows + n % rows) 16 + byte], zeros for the rows that pad N to whole granules."""
    eb = w.element_size()
    N, K = w.shape
    kblock = 64 if eb == 1 else 32
    Np = (N + rows - 1) // rows * rows
    src = torch.zeros((Np, K * eb), dtype=torch.uint8)
    src[:N] = w.contiguous().view(torch.uint8).reshape(N, K * eb)
    n = torch.arange(Np)[:, None]
    k = torch.arange(K)[None, :]
    byte = (k % 8) * 2 if eb == 2 else ((k % 64) // 32) * 8 + k % 8
    off = ((n // rows) * (K // kblock) + k // kblock) * rows * 64 + ((k % 32) // 8 * rows + n % rows) * 16 + byte
    out = torch.zeros((Np * K * eb,), dtype=torch.uint8)
    srcv = src.reshape(Np, K, eb)
    for j in range(eb):
        out[(off + j).reshape(-1)] = srcv[:, :, j].reshape(-1)
    return out


@pytest.mark.parametrize("rows", [4, 8, 16])
@pytest.mark.parametrize("fp8", [False, True])
def test_pack_decode_weights_is_the_documented_permutation(rows, fp8):
    ops, L = _ops()
    for N, K in ((48, 128), (50, 192), (7, 64), (160, 320)):
        w = _rand((N, K), torch.bfloat16, 3 + N)
        if fp8:
            w = torch.randint(0, 256, (N, K), dtype=torch.uint8, generator=torch.Generator().manual_seed(N))
        got = ops.pack_decode_weights(w.to(DEV), rows).cpu()
        ref = packed_layout_reference(w, rows)
        assert got.numel() == ref.numel() == (N + rows - 1) // rows * rows * K * w.element_size()
        assert torch.equal(got, ref), (N, K, rows, fp8)
    with pytest.raises(ValueError):  # K must be whole k blocks
        ops.pack_decode_weights(torch.zeros((16, 48), dtype=torch.uint8, device=DEV), rows)


@pytest.mark.parametrize("fp8", [True, False])
@pytest.mark.parametrize("B,N,K,N2,rows", [(8, 4096, 4096, 6144, 16), (8, 4096, 14336, 28672 // 2, 4), (2, 4096, 4096, 6144, 4),
                                           (4, 2560, 6912, 1000, 8), (16, 4096, 4096, 336, 16), (5, 1024, 1024, 128, 4),
                                           (20, 576, 320, 80, 16), (3, 4160, 512, 48, 8)])
def test_gemv_rowss_packed_equals_row_major_bit_for_bit(fp8, B, N, K, N2, rows):
    """The packed decode layout changes where the kernel finds its operands, not what it multiplies: every product of
    srgpt_gemv_rowss -- residual + published statistics, RMSNorm from a table, fp32 logits, SwiGLU -- returns the row-major call's
    bits for granules of 4 / 8 / 16 rows, fp8 and bf16 weights, 2 ... 20 rows, K with a partial last slice."""
    ops, L = _ops()
    dtype = torch.bfloat16
    if not ops.gemv_rowss_supported(B, fp8):
        pytest.skip("no hand-off kernel for this batch")
    x, w = _rand((B, K), dtype, 41).to(DEV), _rand((N, K), dtype, 42, 0.03).to(DEV)
    r = _rand((B, N), dtype, 43).to(DEV)
    g = (1 + 0.1 * _rand((N,), torch.float32, 44)).to(dtype).to(DEV)
    w2 = _rand((N2, N), dtype, 45, 0.03).to(DEV)

    def variants(m):
        if fp8:
            q8, sc, _ = ops.quantize_fp8_rows(m)
            return dict(w8=q8, wscale=sc), dict(w8=ops.pack_decode_weights(q8, rows), wscale=sc, packed_rows=rows, n_rows=m.shape[0])
        return dict(w=m), dict(w=ops.pack_decode_weights(m, rows), packed_rows=rows, n_rows=m.shape[0])

    rm1, pk1 = variants(w)
    rm2, pk2 = variants(w2)
    h, table = ops.gemv_rowss(x, residual=r, publish=True, **rm1)
    hp, tablep = ops.gemv_rowss(x, residual=r, publish=True, **pk1)
    assert torch.equal(h, hp) and torch.equal(table, tablep)
    for kw in (dict(), dict(out_f32=True)) + ((dict(swiglu=True),) if N2 % (2 * rows) == 0 else ()):
        a = ops.gemv_rowss(h, norm_w=g, eps=1e-5, rowss_in=table, **kw, **rm2)
        b = ops.gemv_rowss(h, norm_w=g, eps=1e-5, rowss_in=table, **kw, **pk2)
        assert torch.equal(a, b), kw
    a = ops.gemv_rowss(h, norm_w=g, eps=1e-5, **rm2)  # statistics computed by the consumer
    b = ops.gemv_rowss(h, norm_w=g, eps=1e-5, **pk2)
    assert torch.equal(a, b)


def test_gemv_rowss_packed_rejects_bad_geometry():
    ops, L = _ops()
    x = _rand((4, 96), torch.bfloat16, 1).to(DEV)
    w = torch.zeros((64 * 96,), dtype=torch.uint8, device=DEV)
    sc = torch.ones((64,), device=DEV)
    with pytest.raises(ValueError):  # fp8: K must be a multiple of 64
        ops.gemv_rowss(x, w8=w, wscale=sc, packed_rows=4, n_rows=64)
    with pytest.raises(ValueError):  # 5 rows per granule is not a layout
        ops.gemv_rowss(_rand((4, 128), torch.bfloat16, 1).to(DEV), w8=torch.zeros((64 * 128,), dtype=torch.uint8, device=DEV), wscale=sc,
                       packed_rows=5, n_rows=64)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(7, 1152), (3, 4608), (5, 64), (2, 72)])
def test_layernorm_rmsnorm(dtype, rows, cols):
    ops, L = _ops()
    if dtype == torch.bfloat16 and cols % 8:
        pytest.skip("cols % 8")
    x = _rand((rows, cols), dtype, 17) + 0.3
    w, b = (1 + 0.1 * _rand((cols,), torch.float32, 18)).to(dtype), _rand((cols,), dtype, 19, 0.1)
    ref = F.layer_norm(x.float(), (cols,), w.float(), b.float(), 1e-6)
    assert_close(ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6), ref, _tol(ref, dtype), 0, "layernorm")
    assert_close(ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, act=L.ACT_GELU_ERF), F.gelu(ref.to(dtype).float()),
                 _tol(ref, dtype), 0, "layernorm+gelu")
    xf = x.float()
    refr = w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dtype).float()
    assert_close(ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5), refr, _tol(refr, dtype), 0, "rmsnorm")


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, causal, kv_len=None):
    B, Tq, Hq, D = q.shape
    Tk, Hkv = k.shape[1], k.shape[2]
    qf, kf, vf = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    rep = Hq // Hkv
    kf, vf = kf.repeat_interleave(rep, 1), vf.repeat_interleave(rep, 1)
    s = qf @ kf.transpose(-1, -2) / math.sqrt(D)
    mask = torch.ones((B, 1, Tq, Tk), dtype=torch.bool)
    if causal:
        mask &= (torch.arange(Tk)[None, :] <= torch.arange(Tq)[:, None] + (Tk - Tq))[None, None]
    if kv_len is not None:
        mask &= (torch.arange(Tk)[None, :] < kv_len[:, None])[:, None, None, :]
    s = s.masked_fill(~mask, float("-inf"))
    return (s.softmax(-1) @ vf).transpose(1, 2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,Tq,Tk,Hq,Hkv,D,causal", [
    (2, 729, 729, 3, 3, 72, False),    # SigLIP geometry (head_dim 72 -> padded to 96)
    (1, 259, 259, 8, 2, 128, True),    # Llama-3 GQA prefill
    (1, 130, 130, 4, 4, 16, True),     # tiny config
    (2, 65, 200, 4, 2, 64, True),      # Tq != Tk causal offset
    (1, 64, 64, 2, 1, 32, False),
    (1, 2048, 2048, 8, 2, 128, True),  # long prompts (model_max_length 4096, scripts/srgpt/llama3_8b/3_sft.sh:58): 32 / 64 key tiles per
    (1, 4096, 4096, 4, 1, 128, True),  # query block, the online rescale runs the whole way
    (1, 700, 4096, 4, 2, 128, True),   # a late chunk of queries against a long cache (causal offset Tk - Tq = 3396)
])
def test_attention(dtype, B, Tq, Tk, Hq, Hkv, D, causal):
    ops, L = _ops()
    q, k, v = _rand((B, Tq, Hq, D), dtype, 20), _rand((B, Tk, Hkv, D), dtype, 21), _rand((B, Tk, Hkv, D), dtype, 22)
    ref = _attn_ref(q, k, v, causal)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal)
    assert_close(out, ref, _tol(ref, dtype), 0, "attention")


def test_attention_packed_qkv_and_kvlen():
    """strided views (packed ViT qkv), per-row kv_len, and a softmax spike that forces the online rescale."""
    ops, L = _ops()
    dtype = torch.bfloat16
    B, T, H, D = 2, 150, 4, 72
    qkv = _rand((B, T, 3, H, D), dtype, 23)
    qkv[0, 140, 1, 2] *= 12.0  # one key far above the rest, late in the sequence (rule: force the rare branch)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    kv_len = torch.tensor([150, 97], dtype=torch.int32)
    ref = _attn_ref(q, k, v, False, kv_len)
    g = qkv.to(DEV)
    out = ops.attention(g[:, :, 0], g[:, :, 1], g[:, :, 2], causal=False, kv_len=kv_len.to(DEV))
    assert_close(out, ref, _tol(ref, dtype), 0, "attention packed + kv_len")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,Hq,Hkv,D,P,max_pos", [
    (1, 32, 8, 128, 300, 1024), (2, 4, 2, 16, 0, 1024), (1, 8, 8, 64, 17, 1024), (1, 20, 20, 128, 700, 1024), (3, 8, 1, 32, 129, 1024),
    # head_dim 128 in bf16 = the MFMA decode kernel (16-key fragments, 64 keys per block): empty cache, a single cached key, the
    # 16- / 64-key group edges, every GQA ratio it is instantiated for, a batch, the benchmark's cache (512: 8 splits), and a cache
    # beyond 64 x 64 positions where every wave walks more than one key group (online rescale of the running P.V)
    (1, 4, 4, 128, 0, 512), (1, 4, 4, 128, 1, 512), (2, 8, 4, 128, 15, 512), (1, 8, 4, 128, 16, 512), (1, 32, 8, 128, 63, 512),
    (1, 32, 8, 128, 64, 512), (3, 32, 8, 128, 259, 512), (1, 16, 2, 128, 386, 512), (1, 32, 8, 128, 511, 512),
    (1, 8, 2, 128, 5000, 8192), (2, 4, 4, 128, 8000, 8192),
    # long contexts in a 4096-position cache (64 fixed 64-key splits): 32 and 63 live splits -- the merge walks them in batches of
    # 16 -- and a short sequence in the same cache (5 live splits, 59 blocks per kv head leave at once)
    (1, 32, 8, 128, 2047, 4096), (1, 32, 8, 128, 4000, 4096), (2, 32, 8, 128, 300, 4096), (1, 32, 8, 128, 1030, 4096)])
def test_rope_append_and_decode_attention(dtype, B, Hq, Hkv, D, P, max_pos):
    """prefill RoPE+append of P tokens, then one decode step; both against the oracle's rotary + softmax."""
    from oracle import srgpt_oracle as so
    ops, L = _ops()
    QW = (Hq + 2 * Hkv) * D
    cfg = so.SrgptConfig(hidden=Hq * D, heads=Hq, kv_heads=Hkv, rope_theta=10000.0)
    from spatialrgpt_amd.weights import rope_tables
    from spatialrgpt_amd.config import SrgptConfig as PC
    cos_t, sin_t = rope_tables(PC(hidden=Hq * D, heads=Hq, kv_heads=Hkv, rope_theta=10000.0), max_pos, dtype, DEV)
    kc = torch.zeros((B, Hkv, max_pos, D), device=DEV, dtype=dtype)
    vc = torch.zeros_like(kc)
    allq = _rand((B, P + 1, QW), dtype, 24)

    def split(x):
        T = x.shape[1]
        return (x[..., :Hq * D].reshape(B, T, Hq, D), x[..., Hq * D:(Hq + Hkv) * D].reshape(B, T, Hkv, D),
                x[..., (Hq + Hkv) * D:].reshape(B, T, Hkv, D))

    q_all, k_all, v_all = split(allq)
    pos = torch.arange(P + 1)[None].expand(B, -1)
    c, s = so.rope_cos_sin(cfg, pos, dtype)
    q_rot, k_rot = so.apply_rope(q_all.transpose(1, 2), k_all.transpose(1, 2), c, s)  # [B,H,T,D]
    if P > 0:
        pre = allq[:, :P].contiguous().to(DEV)
        ops.rope_kv_append(pre, kc, vc, cos_t, sin_t, B, P, Hq, Hkv, D)
        assert_close(pre[..., :Hq * D].reshape(B, P, Hq, D), q_rot[:, :, :P].transpose(1, 2), _tol(q_rot, dtype, 0.5), 0, "rope q")
        assert_close(kc[:, :, :P], k_rot[:, :, :P], _tol(k_rot, dtype, 0.5), 0, "rope k -> cache")
        assert_close(vc[:, :, :P], v_all[:, :P].transpose(1, 2), 0, 0, "v -> cache")
    # decode step for the token at position P
    posd = torch.full((B,), P, dtype=torch.int32, device=DEV)
    out = ops.decode_attention(allq[:, P].contiguous().to(DEV), kc, vc, posd, cos_t, sin_t, Hq, Hkv, D)
    ref = _attn_ref(q_rot[:, :, P:P + 1].transpose(1, 2), k_rot.transpose(1, 2), v_all, True)
    assert_close(out.reshape(B, 1, Hq, D), ref, _tol(ref, dtype), 0, "decode attention")
    assert_close(kc[:, :, P], k_rot[:, :, P], _tol(k_rot, dtype, 0.5), 0, "decode appended k")
    assert_close(vc[:, :, P], v_all[:, P], 0, 0, "decode appended v")


# ------------------------------------------------------------------------------------------------ region extractor
def test_region_pool_golden_kats():
    """MaskPooling known-answer vectors minted from the reference module itself."""
    ops, L = _ops()
    z = load_kat()
    for tag in ["rgb108", "depth27", "soft336_to_108", "soft336_to_96", "up56_to_108"]:
        feat = torch.from_numpy(z[f"pool.{tag}.feat_q64"].astype(np.float32) / 64)
        masks = torch.from_numpy(z[f"pool.{tag}.masks_q16"].astype(np.float32) / 16)
        out = ops.region_pool(feat.to(DEV), masks.to(DEV))
        assert_close(out, torch.from_numpy(z[f"pool.{tag}.out"]), 2e-6, 1e-5, tag)
    feat = torch.from_numpy(z["pool.rgb108_bf16.feat_q64"].astype(np.float32) / 64).bfloat16()
    masks = torch.from_numpy(z["pool.rgb108_bf16.masks_q16"].astype(np.float32) / 16).bfloat16()
    ref = torch.from_numpy(z["pool.rgb108_bf16.out"].view(np.int16).copy()).view(torch.bfloat16)
    out = ops.region_pool(feat.to(DEV), masks.to(DEV))
    assert_close(out, ref, _tol(ref, torch.bfloat16, 0.5), 0, "bf16 KAT")
    assert float(out[4].float().abs().max()) == 0.0  # empty mask -> exact zeros, no NaN
    out_f32mask = ops.region_pool(feat.to(DEV), masks.float().to(DEV))  # mask.float() contract
    assert torch.equal(out_f32mask, out)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,S,fw,C", [(8, 384, 108, 1152), (16, 336, 108, 256), (3, 384, 27, 1152), (17, 96, 24, 64),
                                      (1, 96, 27, 72), (5, 256, 64, 200), (2, 192, 48, 1096)])  # channel counts off the 64-wide slab, one mask
def test_region_pool_true_shape_vs_oracle(dtype, M, S, fw, C):
    from oracle import srgpt_oracle as so
    ops, L = _ops()
    g = torch.Generator().manual_seed(30)
    feat = _rand((fw * fw, C), dtype, 31)
    masks = torch.zeros((M, S, S))
    for m in range(M):
        y0, x0 = int(torch.randint(0, S // 2, (1,), generator=g)), int(torch.randint(0, S // 2, (1,), generator=g))
        h, w = int(torch.randint(S // 8, S // 2, (1,), generator=g)), int(torch.randint(S // 8, S // 2, (1,), generator=g))
        masks[m, y0:y0 + h, x0:x0 + w] = 1
    masks = masks.to(dtype)
    ref = so.mask_pooling(feat[None], [masks])[0]
    out = ops.region_pool(feat.to(DEV), masks.to(DEV))
    assert_close(out, ref, _tol(ref, dtype), 0, "region_pool")
    # linearity in the features (size-independent property): pool(2f) == 2 pool(f) exactly in binary fp
    out2 = ops.region_pool((feat * 2).to(DEV), masks.to(DEV))
    assert torch.equal(out2.float(), out.float() * 2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_avgpool_s2d_im2col(dtype):
    ops, L = _ops()
    for in_w, C in [(108, 64), (96, 32)]:
        x = _rand((2, in_w * in_w, C), dtype, 32)
        ref = F.adaptive_avg_pool2d(x.float().reshape(2, in_w, in_w, C).permute(0, 3, 1, 2), 27).flatten(2).transpose(1, 2)
        assert_close(ops.avgpool(x.to(DEV), 2, in_w, 27), ref, _tol(ref, dtype), 0, f"avgpool {in_w}")
    z = load_kat()
    x = torch.from_numpy(z["s2d.in"]).to(dtype)
    assert torch.equal(ops.s2d(x.to(DEV)).cpu(), torch.from_numpy(z["s2d.out"]).to(dtype))
    x = _rand((1, 24 * 24, 16), dtype, 33)  # even grid: no padding
    from oracle import srgpt_oracle as so
    assert torch.equal(ops.s2d(x.to(DEV)).cpu(), so.flat_square(x.reshape(1, 24, 24, 16)).reshape(1, -1, 64))
    img = _rand((2, 3, 56, 56), dtype, 34)
    kp = 592
    col = ops.im2col(img.to(DEV), 14, kp).cpu()
    ref = F.unfold(img.float(), kernel_size=14, stride=14).transpose(1, 2).reshape(-1, 588).to(dtype)
    assert torch.equal(col[:, :588], ref) and float(col[:, 588:].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ token stream
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_embed_scatter_silu_argmax(dtype):
    ops, L = _ops()
    table = _rand((50, 64), dtype, 35)
    ids = torch.tensor([3, 49, 0, 3, 7])
    assert torch.equal(ops.embed_rows(table.to(DEV), ids.to(DEV)).cpu(), table[ids])
    dst = torch.zeros((10, 64), dtype=dtype, device=DEV)
    src = _rand((4, 64), dtype, 36)
    ops.scatter_rows(src.to(DEV), torch.tensor([9, -1, 0, 4], dtype=torch.int32, device=DEV), dst,
                     src_idx=torch.tensor([3, 2, 1, 0], dtype=torch.int32, device=DEV))
    exp = torch.zeros((10, 64), dtype=dtype)
    exp[9], exp[0], exp[4] = src[3], src[1], src[0]
    assert torch.equal(dst.cpu(), exp)
    gu = _rand((5, 2 * 96), dtype, 37)
    ref = F.silu(gu[:, :96].float()).to(dtype).float() * gu[:, 96:].float()
    assert_close(ops.silu_mul(gu.to(DEV)), ref, _tol(ref, dtype), 0, "silu_mul")
    logits = _rand((3, 128258), torch.float32, 38)
    logits[1, 77] = logits[1, 90000] = 50.0  # tie -> first index wins, like torch.argmax
    logits[2, 128257] = 60.0
    assert torch.equal(ops.argmax(logits.to(DEV)).cpu(), torch.tensor([int(logits[0].argmax()), 77, 128257]))


@pytest.mark.parametrize("rows,V", [(7, 128), (33, 1000), (5, 128258), (1, 17)])
def test_cross_entropy_vs_torch(rows, V):
    """srgpt_cross_entropy == F.cross_entropy(mean over labels != -100) in fp32, including rows that are ignored."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(31)
    logits = torch.randn((rows, V), generator=g) * 3
    labels = torch.randint(0, V, (rows,), generator=g)
    if rows > 2:
        labels[1] = -100
        labels[rows - 1] = -100
    ref = F.cross_entropy(logits, labels, ignore_index=-100)
    loss, n = ops.cross_entropy(logits.to(DEV), labels.to(DEV))
    assert int(n) == int((labels != -100).sum())
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    loss0, n0 = ops.cross_entropy(logits.to(DEV), torch.full((rows,), -100, dtype=torch.int64, device=DEV))
    assert int(n0) == 0 and bool(torch.isnan(loss0))  # torch: mean over zero targets is nan


@pytest.mark.parametrize("geom", ["gqa_4096", "mha_4096", "mha_2560"])
def test_decode_step_equals_per_op_composition_bit_for_bit(geom):
    """srgpt_llm_decode_step (what the hipGraph replays: fused prologues / epilogues, L2 prefetch blocks riding on the attention
    launch, embedding rows dropped by the advance kernel) must be BIT-identical to the same layers composed from the public
    per-op entries (srgpt_gemv + srgpt_decode_attention + srgpt_gemv ...), step after step, across a cache-granule boundary, and
    leave every arrival ticket re-armed (srgpt_llm_decode_sync_state).  (Round 3 used this test to prove the fused attention +
    o_proj launch bit-exact before measuring it slower and removing it.)"""
    import ctypes as C

    from spatialrgpt_amd import _lib as L
    from spatialrgpt_amd import ops
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine
    from spatialrgpt_amd.weights import synth_state_dict

    kw = dict(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=42, patch_size=14, layers=3, vocab=4098,
              mask_token_id=4096, depth_token_id=4097)
    if geom == "gqa_4096":
        kw.update(hidden=4096, inter=14336, heads=32, kv_heads=8)
    elif geom == "mha_4096":
        kw.update(hidden=4096, inter=11008, heads=32, kv_heads=32, rope_theta=10000.0)
    else:
        kw.update(hidden=2560, inter=6912, heads=20, kv_heads=20, rope_theta=10000.0)
    cfg = SrgptConfig(**kw)
    dt = torch.bfloat16
    eng = SrgptEngine(cfg, synth_state_dict(cfg, seed=5, dtype=dt, device=DEV), device=DEV, dtype=dt, rope_positions=512)
    w = eng.w
    T0, G = 120, 14  # contexts 120 .. 133: crosses the 128-row cache granule
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn((1, T0, cfg.hidden), device=DEV, generator=g) * 0.5).to(dt)
    st, _, _ = eng.prefill(x, max_new=G + 2)
    Hq, Hkv, D, Hd = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
    toks = torch.randint(3, 4096, (G,), device=DEV, generator=g)
    # the per-op composition keeps its own copy of the cache (the engine's step appends to st's)
    kc, vc = st.kcache.clone(), st.vcache.clone()
    for t in range(G):
        tok = toks[t:t + 1].reshape(1, 1)
        got = eng.step(st, tok)
        h = ops.embed_rows(w.embed, tok.reshape(-1))
        pos = torch.tensor([T0 + t], device=DEV, dtype=torch.int32)
        for i in range(cfg.layers):
            qkv = ops.gemv(h, w.llm_t["wqkv"][i], norm_w=w.llm_t["attn_norm"][i], eps=cfg.rms_eps)
            a = ops.decode_attention(qkv, kc[i], vc[i], pos, w.rope_cos, w.rope_sin, Hq, Hkv, D)
            h = ops.gemv(a, w.llm_t["wo"][i], residual=h)
            act = ops.gemv(h, w.llm_t["wgu"][i], norm_w=w.llm_t["mlp_norm"][i], eps=cfg.rms_eps, swiglu=True)
            h = ops.gemv(act, w.llm_t["wdown"][i], residual=h)
        ref = ops.gemv(h, w.lm_head, norm_w=w.final_norm, eps=cfg.rms_eps, out_f32=True)
        assert torch.equal(got, ref), f"{geom}: step {t} (context {T0 + t}): max diff {float((got - ref).abs().max())}"
    assert torch.equal(st.kcache[:, :, :, :T0 + G], kc[:, :, :, :T0 + G])
    # the arrival tickets of the decode attention are re-armed (zero) between steps
    assert L.load().srgpt_llm_decode_sync_state(C.byref(w.llm), C.byref(st.c), ops._stream()) == 0, L.last_error()


@pytest.mark.parametrize("B,fp8", [(2, False), (8, False), (8, True), (3, True), (20, False), (20, True)])
def test_batched_decode_step_equals_rowss_composition_bit_for_bit(B, fp8):
    """The batched decode step (2+ rows: o_proj / down_proj publish the rows' sums of squares, the next RMSNorm reads them) must be
    BIT-identical to the same layers composed from srgpt_gemv_rowss + srgpt_decode_attention, step after step; layer 0's q/k/v
    normalises the embedding rows itself, lm_head takes the last down_proj's table."""
    import ctypes as C

    from spatialrgpt_amd import _lib as L
    from spatialrgpt_amd import ops
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine
    from spatialrgpt_amd.weights import synth_state_dict

    cfg = SrgptConfig(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=42, patch_size=14, layers=3, vocab=4098,
                      mask_token_id=4096, depth_token_id=4097, hidden=4096, inter=14336, heads=32, kv_heads=8)
    dt = torch.bfloat16
    eng = SrgptEngine(cfg, synth_state_dict(cfg, seed=5, dtype=dt, device=DEV), device=DEV, dtype=dt, rope_positions=512,
                      **({"llm_weight_format": "fp8"} if fp8 else {}))
    w = eng.w
    assert ops.gemv_rowss_supported(B, fp8)
    T0, G = 60, 5
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn((B, T0, cfg.hidden), device=DEV, generator=g) * 0.5).to(dt)
    st, _, _ = eng.prefill(x, max_new=G + 2)
    Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
    toks = torch.randint(3, 4096, (G, B), device=DEV, generator=g)
    kc, vc = st.kcache.clone(), st.vcache.clone()

    def mat(name, i):
        if fp8 and name in w.llm_pk:  # the packed copy the step streams (the row-major call returns the same bits: test above)
            return dict(w8=w.llm_pk[name][i], wscale=w.llm_q[name][1][i], packed_rows=w.pk_rows[name], n_rows=w.llm_q[name][0][i].shape[0])
        if fp8:
            return dict(w8=w.llm_q[name][0][i], wscale=w.llm_q[name][1][i])
        return dict(w=w.llm_t[name][i])

    if fp8:
        assert set(w.llm_pk) == {"wqkv", "wo", "wgu", "wdown"} and w.pk_rows["wo"] == 16 and w.pk_rows["wgu"] == 4

    for t in range(G):
        tok = toks[t].reshape(B, 1)
        got = eng.step(st, tok)
        h = ops.embed_rows(w.embed, tok.reshape(-1))
        pos = torch.full((B,), T0 + t, device=DEV, dtype=torch.int32)
        ss = None
        for i in range(cfg.layers):
            qkv = ops.gemv_rowss(h, norm_w=w.llm_t["attn_norm"][i], eps=cfg.rms_eps, rowss_in=ss, **mat("wqkv", i))
            a = ops.decode_attention(qkv, kc[i], vc[i], pos, w.rope_cos, w.rope_sin, Hq, Hkv, D)
            h, ss = ops.gemv_rowss(a, residual=h, publish=True, **mat("wo", i))
            act = ops.gemv_rowss(h, norm_w=w.llm_t["mlp_norm"][i], eps=cfg.rms_eps, swiglu=True, rowss_in=ss, **mat("wgu", i))
            h, ss = ops.gemv_rowss(act, residual=h, publish=True, **mat("wdown", i))
        head = dict(w8=w.lm_head8, wscale=w.lm_head_scale) if fp8 else dict(w=w.lm_head)
        ref = ops.gemv_rowss(h, norm_w=w.final_norm, eps=cfg.rms_eps, out_f32=True, rowss_in=ss, **head)
        assert torch.equal(got, ref), f"B={B} fp8={fp8}: step {t}: max diff {float((got - ref).abs().max())}"
    assert torch.equal(st.kcache[:, :, :, :T0 + G], kc[:, :, :, :T0 + G])
    assert L.load().srgpt_llm_decode_sync_state(C.byref(w.llm), C.byref(st.c), ops._stream()) == 0, L.last_error()


@pytest.mark.parametrize("B,Hq,Hkv,P", [(1, 32, 8, 300), (2, 8, 8, 77), (1, 20, 20, 700)])
def test_decode_attention_bits_do_not_depend_on_cache_capacity(B, Hq, Hkv, P):
    """A pooled decode state serves requests of different sizes, so the same sequence may meet caches of different capacities: the
    attention output must be bit-identical whatever `max_pos` is (the key partition depends on the sequence length only).
    scripts/soak.py caught the first MFMA decode kernel splitting keys by capacity."""
    from spatialrgpt_amd.config import SrgptConfig as PC
    from spatialrgpt_amd.weights import rope_tables
    ops, L = _ops()
    D, dtype = 128, torch.bfloat16
    QW = (Hq + 2 * Hkv) * D
    outs = []
    for max_pos in (768, 1024, 2048, 4096):
        cos_t, sin_t = rope_tables(PC(hidden=Hq * D, heads=Hq, kv_heads=Hkv, rope_theta=10000.0), max_pos, dtype, DEV)
        kc = torch.zeros((B, Hkv, max_pos, D), device=DEV, dtype=dtype)
        vc = torch.zeros_like(kc)
        allq = _rand((B, P + 1, QW), dtype, 24).to(DEV)
        ops.rope_kv_append(allq[:, :P].contiguous(), kc, vc, cos_t, sin_t, B, P, Hq, Hkv, D)
        kc[:, :, P + 1:] = 7.0  # whatever lies beyond the sequence must not matter
        vc[:, :, P + 1:] = -3.0
        posd = torch.full((B,), P, dtype=torch.int32, device=DEV)
        outs.append(ops.decode_attention(allq[:, P].contiguous(), kc, vc, posd, cos_t, sin_t, Hq, Hkv, D))
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


# ------------------------------------------------------------------------------------------------ beam cache permutation (ABI 9)
@pytest.mark.parametrize("nb,B,dtype", [(2, 1, torch.bfloat16), (3, 2, torch.bfloat16), (5, 3, torch.bfloat16), (8, 1, torch.float32), (4, 2, torch.float32)])
def test_kv_beam_reorder_equals_index_select(nb, B, dtype):
    """srgpt_kv_beam_reorder against HF's `_reorder_cache` arithmetic (index_select over the row axis), in place, live positions only:
    positions past `live` keep their old bytes; indices stay inside their batch item; identity rows are untouched."""
    ops, L = _ops()
    Ly, Hkv, P, D, live = 3, 4, 70, 128 if dtype == torch.bfloat16 else 64, 37
    g = torch.Generator().manual_seed(nb * 10 + B)
    kc = torch.randn((Ly, B * nb, Hkv, P, D), generator=g).to(dtype).to(DEV)
    vc = torch.randn((Ly, B * nb, Hkv, P, D), generator=g).to(dtype).to(DEV)
    idx = torch.cat([b * nb + torch.randint(0, nb, (nb,), generator=g) for b in range(B)]).to(DEV)
    idx[0] = 0  # (one identity row)
    want_k, want_v = kc.clone(), vc.clone()
    want_k[:, :, :, :live] = kc[:, :, :, :live].index_select(1, idx)
    want_v[:, :, :, :live] = vc[:, :, :, :live].index_select(1, idx)
    assert ops.kv_beam_reorder(kc, vc, idx, nb, live)
    assert torch.equal(kc, want_k) and torch.equal(vc, want_v)
    assert not ops.kv_beam_reorder(kc, vc, torch.arange(B * nb, device=DEV), 9 if (B * nb) % 9 == 0 else 11, live)  # served by torch
