"""Tensor-level wrappers over the C ABI (include/srgpt.h).  torch is used only for device memory and
streams: every arithmetic op below is a HIP kernel from libsrgpt_hip.so, launched on the caller's
current torch stream.  Inputs must be CUDA(HIP) tensors; there is no CPU path."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}


def dt_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise ValueError(f"srgpt kernels support float32 and bfloat16 tensors, got {t.dtype}") from None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("srgpt ops need tensors on the GPU (no CPU fallback)")


def _same_dtype(op: str, ref: torch.Tensor, **others):
    """The kernels take ONE dtype code per call (the activation's): a weight / bias / residual buffer in another dtype would
    be read with the wrong element size (bf16 weights read as fp32 run 2x past their end).  torch raises a dtype-mismatch
    RuntimeError for the same call; so do we."""
    for name, t in others.items():
        if t is not None and t.dtype != ref.dtype:
            raise RuntimeError(f"{op}: expected {name} to have the activation dtype {ref.dtype}, got {t.dtype}")


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


_gemm_ws = {}


def _splitk_ws(device, M, N):
    """fp32 split-K workspace (grown on demand, one per device; same-stream reuse is ordered by the stream)."""
    need = 8 * M * N * 4
    t = _gemm_ws.get(device)
    if t is None or t.numel() < need:
        t = torch.empty((need,), device=device, dtype=torch.uint8)
        _gemm_ws[device] = t
    return t


def gemm(a, w, bias=None, residual=None, act=L.ACT_NONE, out=None, out_f32=False, bias_mod=0, res_mod=0,
         out_mode=L.OUT_PLAIN, gw=0, out_shape=None):
    """act(a @ w.T + bias) + residual.  a [M,K] (row stride may exceed K), w [N,K]."""
    _dev(a, w, bias, residual)
    _same_dtype("gemm", a, weight=w, bias=bias, residual=residual)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.is_contiguous()
    if out is None:
        shape = out_shape if out_shape is not None else (M, N)
        out = torch.empty(shape, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    ldc = N if out_mode != L.OUT_PLAIN else out.stride(0) if out.dim() == 2 else N
    ws = _splitk_ws(a.device, M, N) if (a.dtype == torch.bfloat16 and M * N <= (1 << 24)) else None
    L.check(L.load().srgpt_gemm(_p(a), _p(w), _p(bias), _p(residual), _p(out), M, N, K, a.stride(0), ldc, act,
                                bias_mod, res_mod, int(out_f32), out_mode, gw, _p(ws), 0 if ws is None else ws.numel(),
                                dt_code(a), _stream()))
    return out


def gemm_norm(a, w, residual, norm_w, eps, bias=None, norm_b=None, out=None, y=None):
    """(a @ w.T + bias + residual, norm(that)): the tail of an attention / MLP block and the next block's input norm in one call --
    RMSNorm (norm_b None: Llama) or LayerNorm (ViT).  `out` may be `residual` (in place), `y` may be `a`.  Bit-identical to gemm(...)
    followed by rmsnorm(...) / layernorm(...)."""
    _dev(a, w, residual, norm_w, bias, norm_b)
    _same_dtype("gemm_norm", a, weight=w, residual=residual, norm_weight=norm_w, bias=bias, norm_bias=norm_b)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.is_contiguous() and w.is_contiguous() and (residual is None or residual.is_contiguous())
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    if y is None:
        y = torch.empty((M, N), device=a.device, dtype=a.dtype)
    ws = _splitk_ws(a.device, M, N) if (a.dtype == torch.bfloat16 and M * N <= (1 << 24)) else None
    kind = L.NORM_RMS if norm_b is None else L.NORM_LAYER
    L.check(L.load().srgpt_gemm_norm(_p(a), _p(w), _p(bias), _p(residual), _p(out), M, N, K, _p(ws), 0 if ws is None else ws.numel(),
                                     kind, _p(norm_w), _p(norm_b), _p(y), float(eps), dt_code(a), _stream()))
    return out, y


def gemv(x, w, norm_w=None, eps=0.0, residual=None, swiglu=False, out=None, out_f32=False):
    """decode GEMV: x [B,K], w [N(,2N if swiglu),K] -> [B,N]."""
    _dev(x, w, norm_w, residual)
    _same_dtype("gemv", x, weight=w, norm_weight=norm_w, residual=residual)
    B, K = x.shape
    N = w.shape[0] // 2 if swiglu else w.shape[0]
    if out is None:
        out = torch.empty((B, N), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    L.check(L.load().srgpt_gemv(_p(_c(x)), _p(w), _p(norm_w), float(eps), _p(residual), _p(out), B, N, K, int(swiglu),
                                int(out_f32), dt_code(x), _stream()))
    return out


def quantize_fp8_rows(w: torch.Tensor):
    """Weight-only fp8 quantisation used by the decode path.  Per output row: scale = the smallest power of two with
    max|row| / scale <= 448 (exact on every device: frexp / ldexp, no division), codes = OCP e4m3fn bytes of row / scale
    (round to nearest even).  A power-of-two scale costs a floating-point format nothing in relative precision.
    Returns (codes uint8 [N, K], scale fp32 [N], dequantised weights in w.dtype)."""
    wf = w.float()
    m, e = torch.frexp(wf.abs().amax(dim=1).clamp_min(2.0 ** -100))  # amax = m * 2^e, m in [0.5, 1)
    k = e - 9 + (m > 0.875).to(e.dtype)                              # 448 = 0.875 * 2^9
    scale = torch.ldexp(torch.ones_like(m), k)
    q = (wf * torch.ldexp(torch.ones_like(m), -k)[:, None]).to(torch.float8_e4m3fn)
    deq = (q.float() * scale[:, None]).to(w.dtype)
    return q.view(torch.uint8).contiguous(), scale.contiguous(), deq.contiguous()


def gemv_w8(x, w8, wscale, norm_w=None, eps=0.0, residual=None, swiglu=False, out=None, out_f32=False):
    """decode product with fp8 weights: x [B,K] bf16, w8 uint8 [N(,2N if swiglu),K], wscale fp32 [rows] -> [B,N]."""
    _dev(x, w8, wscale, norm_w, residual)
    _same_dtype("gemv_w8", x, norm_weight=norm_w, residual=residual)
    if x.dtype != torch.bfloat16 or w8.dtype != torch.uint8 or wscale.dtype != torch.float32:
        raise ValueError("gemv_w8: x must be bf16, w8 uint8, wscale fp32")
    B, K = x.shape
    N = w8.shape[0] // 2 if swiglu else w8.shape[0]
    if out is None:
        out = torch.empty((B, N), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    L.check(L.load().srgpt_gemv_w8(_p(_c(x)), _p(w8), _p(wscale), _p(norm_w), float(eps), _p(residual), _p(out), B, N, K,
                                   int(swiglu), int(out_f32), _stream()))
    return out


def kv_beam_reorder(kcache, vcache, beam_idx, num_beams: int, live: int) -> bool:
    """srgpt_kv_beam_reorder: in-place permutation of the live cache rows by the beam indices (kcache / vcache [layers, batch *
    num_beams, kv_heads, max_pos, head_dim], beam_idx int64 [batch * num_beams] on the device).  False: this beam count is not
    served by the kernel (the caller permutes with torch ops)."""
    if not 2 <= num_beams <= 8:
        return False
    _dev(kcache, vcache, beam_idx)
    Ly, R, Hkv, P, D = kcache.shape
    L.check(L.load().srgpt_kv_beam_reorder(_p(kcache), _p(vcache), _p(beam_idx.to(torch.int64).contiguous()), Ly, R // num_beams,
                                           int(num_beams), Hkv, P, D, int(live), dt_code(kcache), _stream()))
    return True


def gemv_rowss_supported(batch: int, fp8: bool = False) -> bool:
    """does a (batch, bf16 activations, bf16 / fp8 weights) decode product take the kernel with the row-statistics hand-off?"""
    return bool(L.load().srgpt_gemv_rowss_supported(int(batch), L.BF16, int(fp8)))


def pack_decode_weights(w, rows: int):
    """srgpt_pack_decode_weights: the packed decode layout (granules of `rows` = 4 / 8 / 16 weight rows in MFMA-operand order,
    include/srgpt.h ABI 9) of a row-major matrix [N, K] (uint8 = fp8 bytes, or bf16).  Returns a flat uint8 tensor; a SwiGLU matrix
    [gate; up] is packed as the one matrix of 2 N rows it is."""
    _dev(w)
    if w.dim() != 2 or not w.is_contiguous() or w.dtype not in (torch.uint8, torch.bfloat16):
        raise ValueError("pack_decode_weights: a contiguous uint8 (fp8) or bf16 matrix")
    N, K = w.shape
    eb = w.element_size()
    lib = L.load()
    out = torch.empty((int(lib.srgpt_packed_bytes(N, K, eb, int(rows))),), device=w.device, dtype=torch.uint8)
    L.check(lib.srgpt_pack_decode_weights(_p(w), _p(out), N, K, eb, int(rows), _stream()))
    return out


def gemv_rowss(x, w=None, w8=None, wscale=None, norm_w=None, eps=0.0, residual=None, swiglu=False, out=None, out_f32=False,
               rowss_in=None, publish=False, packed_rows=0, n_rows=None):
    """srgpt_gemv_rowss: the decode product of 2+ bf16 rows as the batched decode step runs it.  `rowss_in`: the statistics
    table of x's rows ([B, 512] fp32, from the call that produced x) replaces the RMSNorm's own reduction; `publish`: also
    return the table of the output rows.  `packed_rows` > 0: w / w8 is the flat packed array of pack_decode_weights (n_rows = the
    matrix' row count, 2 N for SwiGLU).  Returns out, or (out, table)."""
    wt = w8 if w8 is not None else w
    _dev(x, wt, wscale, norm_w, residual, rowss_in)
    _same_dtype("gemv_rowss", x, norm_weight=norm_w, residual=residual)
    if x.dtype != torch.bfloat16:
        raise ValueError("gemv_rowss: bf16 activations only")
    B, K = x.shape
    if packed_rows:
        if n_rows is None:
            raise ValueError("gemv_rowss: a packed matrix needs n_rows")
        rows_total = int(n_rows)
    else:
        rows_total = wt.shape[0]
    N = rows_total // 2 if swiglu else rows_total
    if out is None:
        out = torch.empty((B, N), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    table = torch.empty((B, L.ROWSS_STRIDE), device=x.device, dtype=torch.float32) if publish else None
    if rowss_in is not None and (rowss_in.dtype != torch.float32 or tuple(rowss_in.shape) != (B, L.ROWSS_STRIDE)
                                 or not rowss_in.is_contiguous()):
        raise ValueError("gemv_rowss: rowss_in must be a contiguous fp32 [batch, 512] table")
    L.check(L.load().srgpt_gemv_rowss(_p(_c(x)), _p(w) if w8 is None else None, _p(w8), _p(wscale), _p(norm_w), float(eps),
                                      _p(residual), _p(out), B, N, K, int(swiglu), int(out_f32), _p(rowss_in), _p(table),
                                      int(packed_rows), _stream()))
    return (out, table) if publish else out


def gemm_w8(a, w8, wscale, bias=None, residual=None, act=L.ACT_NONE, out=None, out_f32=False):
    """act((a @ fp8(w8).T) * wscale + bias) + residual: a [M,K] bf16 (row stride may exceed K), w8 uint8 [N,K] (OCP e4m3fn),
    wscale fp32 [N]."""
    _dev(a, w8, wscale, bias, residual)
    _same_dtype("gemm_w8", a, bias=bias, residual=residual)
    if a.dtype != torch.bfloat16 or w8.dtype != torch.uint8 or wscale.dtype != torch.float32:
        raise ValueError("gemm_w8: a must be bf16, w8 uint8, wscale fp32")
    M, K = a.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and a.stride(1) == 1 and w8.is_contiguous() and wscale.numel() == N
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    ws = _splitk_ws(a.device, M, N) if M * N <= (1 << 24) else None
    L.check(L.load().srgpt_gemm_w8(_p(a), _p(w8), _p(wscale), _p(bias), _p(residual), _p(out), M, N, K, a.stride(0), out.stride(0),
                                   act, int(out_f32), _p(ws), 0 if ws is None else ws.numel(), _stream()))
    return out


def quant_rows_e4m3(x):
    """per-row (per-token) fp8 quantisation of activations: x [M,K] bf16 (row stride may exceed K) -> (codes uint8 [M,K],
    scale fp32 [M]); the power-of-two rule of quantize_fp8_rows, on the device (srgpt_quant_rows_e4m3)."""
    _dev(x)
    if x.dtype != torch.bfloat16 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("quant_rows_e4m3: x must be a bf16 matrix with contiguous rows")
    M, K = x.shape
    q = torch.empty((M, K), device=x.device, dtype=torch.uint8)
    sc = torch.empty((M,), device=x.device, dtype=torch.float32)
    L.check(L.load().srgpt_quant_rows_e4m3(_p(x), _p(q), _p(sc), M, K, x.stride(0), _stream()))
    return q, sc


def quant_rows_e4m3_rmsnorm(x, norm_w, eps):
    """quant_rows_e4m3(rmsnorm(x, norm_w, eps)) in one launch (bit-identical to the two; the W8A8 prefill's form)."""
    _dev(x, norm_w)
    if x.dtype != torch.bfloat16 or norm_w.dtype != torch.bfloat16 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("quant_rows_e4m3_rmsnorm: x must be a bf16 matrix with contiguous rows, norm_w bf16")
    M, K = x.shape
    q = torch.empty((M, K), device=x.device, dtype=torch.uint8)
    sc = torch.empty((M,), device=x.device, dtype=torch.float32)
    L.check(L.load().srgpt_quant_rows_e4m3_rmsnorm(_p(x), _p(norm_w), float(eps), _p(q), _p(sc), M, K, x.stride(0), _stream()))
    return q, sc


def quant_rows_e4m3_swiglu(gate_up):
    """quant_rows_e4m3(silu_mul(gate_up)) in one launch: gate_up [M, 2 * inter] = [gate | up], contiguous."""
    _dev(gate_up)
    if gate_up.dtype != torch.bfloat16 or gate_up.dim() != 2 or not gate_up.is_contiguous() or gate_up.shape[1] % 2:
        raise ValueError("quant_rows_e4m3_swiglu: gate_up must be a contiguous bf16 [M, 2 * inter] matrix")
    M, inter = gate_up.shape[0], gate_up.shape[1] // 2
    q = torch.empty((M, inter), device=gate_up.device, dtype=torch.uint8)
    sc = torch.empty((M,), device=gate_up.device, dtype=torch.float32)
    L.check(L.load().srgpt_quant_rows_e4m3_swiglu(_p(gate_up), _p(q), _p(sc), M, inter, _stream()))
    return q, sc


def gemm_w8a8(a8, ascale, w8, wscale, bias=None, residual=None, out=None, out_f32=False):
    """((fp8(a8) @ fp8(w8).T) * ascale[:,None] * wscale[None,:] + bias) + residual on the fp8 matrix pipe: a8 uint8 [M,K],
    w8 uint8 [N,K] (OCP e4m3fn), ascale fp32 [M], wscale fp32 [N]; bias / residual / out bf16 (out fp32 if out_f32)."""
    _dev(a8, ascale, w8, wscale, bias, residual)
    if a8.dtype != torch.uint8 or w8.dtype != torch.uint8 or ascale.dtype != torch.float32 or wscale.dtype != torch.float32:
        raise ValueError("gemm_w8a8: a8 / w8 must be uint8, ascale / wscale fp32")
    for name, t in (("bias", bias), ("residual", residual)):
        if t is not None and t.dtype != torch.bfloat16:
            raise ValueError(f"gemm_w8a8: {name} must be bf16")
    M, K = a8.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and a8.stride(1) == 1 and w8.is_contiguous() and ascale.numel() == M and wscale.numel() == N
    if out is None:
        out = torch.empty((M, N), device=a8.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    ws = _splitk_ws(a8.device, M, N) if M * N <= (1 << 24) else None
    L.check(L.load().srgpt_gemm_w8a8(_p(a8), _p(ascale), _p(w8), _p(wscale), _p(bias), _p(residual), _p(out), M, N, K,
                                     a8.stride(0), out.stride(0), int(out_f32), _p(ws), 0 if ws is None else ws.numel(),
                                     _stream()))
    return out


def layernorm(x, w, b, eps, act=L.ACT_NONE, out=None):
    _dev(x, w, b)
    _same_dtype("layernorm", x, weight=w, bias=b, out=out)
    x = _c(x)
    cols = x.shape[-1]
    rows = x.numel() // cols
    if out is None:
        out = torch.empty_like(x)
    L.check(L.load().srgpt_layernorm(_p(x), _p(w), _p(b), _p(out), rows, cols, float(eps), act, dt_code(x), _stream()))
    return out


def rmsnorm(x, w, eps, out=None):
    _dev(x, w)
    _same_dtype("rmsnorm", x, weight=w, out=out)
    x = _c(x)
    cols = x.shape[-1]
    rows = x.numel() // cols
    if out is None:
        out = torch.empty_like(x)
    L.check(L.load().srgpt_rmsnorm(_p(x), _p(w), _p(out), rows, cols, float(eps), dt_code(x), _stream()))
    return out


def attention(q, k, v, causal=False, scale=None, kv_len=None):
    """q [B,Tq,Hq,D], k/v [B,Tk,Hkv,D] (arbitrary strides with unit last stride) -> [B,Tq,Hq,D]."""
    _dev(q, k, v)
    _same_dtype("attention", q, k=k, v=v)
    B, Tq, Hq, D = q.shape
    Tk, Hkv = k.shape[1], k.shape[2]
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    o = torch.empty((B, Tq, Hq, D), device=q.device, dtype=q.dtype)
    L.check(L.load().srgpt_attention(_p(q), _p(k), _p(v), _p(o), B, Tq, Tk, Hq, Hkv, D, q.stride(0), q.stride(1),
                                     q.stride(2), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1),
                                     v.stride(2), float(scale), int(causal), _p(kv_len), dt_code(q), _stream()))
    return o


def gemm_swiglu(a, wgu, out=None):
    """silu(a @ wg.T) * (a @ wu.T) for the stacked weight wgu = [wg; wu] ([2 I, K]): gemm + silu_mul in one call."""
    _dev(a, wgu)
    _same_dtype("gemm_swiglu", a, weight=wgu)
    M, K = a.shape
    I = wgu.shape[0] // 2
    assert wgu.shape == (2 * I, K) and a.is_contiguous() and wgu.is_contiguous()
    if out is None:
        out = torch.empty((M, I), device=a.device, dtype=a.dtype)
    scratch = torch.empty((M, 2 * I), device=a.device, dtype=a.dtype)
    ws = _splitk_ws(a.device, M, 2 * I) if (a.dtype == torch.bfloat16 and M * 2 * I <= (1 << 24)) else None
    L.check(L.load().srgpt_gemm_swiglu(_p(a), _p(wgu), _p(out), M, I, K, _p(scratch), _p(ws), 0 if ws is None else ws.numel(),
                                       dt_code(a), _stream()))
    return out


def gemm_rope_kv_append(a, w, kcache, vcache, cos_tab, sin_tab, B, T, Hq, Hkv, D, pos0=None, qkv=None):
    """qkv = a @ w.T for the B * T token rows, RoPE on q (in qkv) and k, k / v appended to the caches: gemm + rope_kv_append in one call."""
    _dev(a, w, kcache, vcache, cos_tab, sin_tab)
    M, K = a.shape
    N = (Hq + 2 * Hkv) * D
    assert M == B * T and w.shape == (N, K) and a.is_contiguous() and w.is_contiguous()
    if qkv is None:
        qkv = torch.empty((M, N), device=a.device, dtype=a.dtype)
    ws = _splitk_ws(a.device, M, N) if (a.dtype == torch.bfloat16 and M * N <= (1 << 24)) else None
    max_pos = kcache.shape[-2]
    L.check(L.load().srgpt_gemm_rope_kv_append(_p(a), _p(w), _p(qkv), K, _p(ws), 0 if ws is None else ws.numel(), _p(kcache),
                                               _p(vcache), _p(pos0), _p(cos_tab), _p(sin_tab), B, T, Hq, Hkv, D, max_pos,
                                               dt_code(a), _stream()))
    return qkv


def rope_kv_append(qkv, kcache, vcache, cos_tab, sin_tab, B, T, Hq, Hkv, D, pos0=None):
    _dev(qkv, kcache, vcache, cos_tab, sin_tab)
    max_pos = kcache.shape[-2]
    L.check(L.load().srgpt_rope_kv_append(_p(qkv), _p(kcache), _p(vcache), _p(pos0), _p(cos_tab), _p(sin_tab), B, T, Hq,
                                          Hkv, D, max_pos, dt_code(qkv), _stream()))


def decode_attention(qkv, kcache, vcache, pos, cos_tab, sin_tab, Hq, Hkv, D):
    _dev(qkv, kcache, vcache, pos)
    B = qkv.shape[0]
    max_pos = kcache.shape[-2]
    out = torch.empty((B, Hq * D), device=qkv.device, dtype=qkv.dtype)
    ws = torch.zeros((L.load().srgpt_decode_attn_ws_floats(B, Hq, D),), device=qkv.device, dtype=torch.float32)  # tickets start at 0
    L.check(L.load().srgpt_decode_attention(_p(qkv), _p(kcache), _p(vcache), _p(pos), _p(cos_tab), _p(sin_tab), _p(out),
                                            _p(ws), B, Hq, Hkv, D, max_pos, dt_code(qkv), _stream()))
    return out


def region_pool(feat, masks, fw: Optional[int] = None):
    """MaskPooling for one image: feat [L,C], masks [M,mh,mw] (float32 or bfloat16) -> [M,C]."""
    _dev(feat, masks)
    Lf, Cc = feat.shape
    M, mh, mw = masks.shape
    sf = (Lf / (mh * mw)) ** 0.5  # base_extractor.py:53-54 (python float = double)
    oh, ow = int(math.floor(mh * sf)), int(math.floor(mw * sf))  # F.interpolate(scale_factor=...) output size
    if oh * ow != Lf or oh != ow:
        raise RuntimeError(f"mask of size {mh}x{mw} resamples to {oh}x{ow}, which does not match {Lf} feature tokens")
    if masks.dtype not in (torch.float32, torch.bfloat16):
        masks = masks.float()  # reference: mask.float()
    masks = _c(masks)
    rs = 1.0 / sf  # ATen passes static_cast<float>(1.0 / scale_factor)
    outs = []
    lib = L.load()
    for m0 in range(0, M, 16):
        mm = min(16, M - m0)
        out = torch.empty((mm, Cc), device=feat.device, dtype=feat.dtype)
        ws = torch.empty((lib.srgpt_region_pool_ws_floats(mm, oh, Cc),), device=feat.device, dtype=torch.float32)
        L.check(lib.srgpt_region_pool(_p(_c(feat)), _p(masks[m0:m0 + mm]), _p(out), _p(ws), mm, mh, mw, oh, Cc, rs, rs,
                                      _DT[masks.dtype], dt_code(feat), _stream()))
        outs.append(out)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)


def region_pool_u8(feat, masks_u8, size: int):
    """MaskPooling straight from raw uint8 masks [M, H, W]: the cv2-nearest resize to the processor size
    (`size` x `size`, image_aspect_ratio == "resize"), float(uint8) and the bilinear resample run inside the pooling kernel.
    Bit-identical to region_pool(feat, process_regions_device(masks, ...))."""
    from .mm_utils import cv2_nearest_index

    _dev(feat, masks_u8)
    if masks_u8.dtype != torch.uint8 or masks_u8.dim() != 3:
        raise ValueError("region_pool_u8: masks must be a uint8 tensor [M, H, W]")
    Lf, Cc = feat.shape
    M, rh, rw = masks_u8.shape
    mh = mw = int(size)
    sf = (Lf / (mh * mw)) ** 0.5
    oh, ow = int(math.floor(mh * sf)), int(math.floor(mw * sf))
    if oh * ow != Lf or oh != ow:
        raise RuntimeError(f"mask of size {mh}x{mw} resamples to {oh}x{ow}, which does not match {Lf} feature tokens")
    ys = torch.from_numpy(cv2_nearest_index(rh, mh)).to(feat.device)
    xs = torch.from_numpy(cv2_nearest_index(rw, mw)).to(feat.device)
    masks_u8 = _c(masks_u8)
    rs = 1.0 / sf
    lib = L.load()
    outs = []
    for m0 in range(0, M, 16):
        mm = min(16, M - m0)
        out = torch.empty((mm, Cc), device=feat.device, dtype=feat.dtype)
        ws = torch.empty((lib.srgpt_region_pool_ws_floats(mm, oh, Cc),), device=feat.device, dtype=torch.float32)
        L.check(lib.srgpt_region_pool_u8(_p(_c(feat)), _p(masks_u8[m0:m0 + mm]), _p(ys), _p(xs), _p(out), _p(ws), mm, rh, rw, mh, mw,
                                         oh, Cc, rs, rs, dt_code(feat), _stream()))
        outs.append(out)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)


def avgpool(x, n_img, in_w, out_w):
    _dev(x)
    Cc = x.shape[-1]
    y = torch.empty((n_img, out_w * out_w, Cc), device=x.device, dtype=x.dtype)
    L.check(L.load().srgpt_avgpool(_p(_c(x)), _p(y), n_img, in_w, out_w, Cc, dt_code(x), _stream()))
    return y


def s2d(x):
    """DownSampleBlock: [n, g*g, C] -> [n, ceil(g/2)^2, 4C]."""
    _dev(x)
    n, Lx, Cc = x.shape
    g = int(Lx ** 0.5)
    hb = (g + 1) // 2
    y = torch.empty((n, hb * hb, 4 * Cc), device=x.device, dtype=x.dtype)
    L.check(L.load().srgpt_s2d(_p(_c(x)), _p(y), n, g, Cc, dt_code(x), _stream()))
    return y


def im2col(images, patch, kp):
    _dev(images)
    n, ch, S, _ = images.shape
    assert ch == 3
    g = S // patch
    out = torch.empty((n * g * g, kp), device=images.device, dtype=images.dtype)
    L.check(L.load().srgpt_im2col(_p(_c(images)), _p(out), n, S, patch, kp, dt_code(images), _stream()))
    return out


def embed_rows(table, ids):
    """rows of `table` at `ids`; ids must already be validated against the table height (engine._check_ids)."""
    _dev(table, ids)
    ids = _c(ids.to(torch.int64)).reshape(-1)
    out = torch.empty((ids.numel(), table.shape[1]), device=table.device, dtype=table.dtype)
    L.check(L.load().srgpt_embed_rows(_p(table), _p(ids), _p(out), ids.numel(), table.shape[1], dt_code(table), _stream()))
    return out


def scatter_rows(src, idx, dst, src_idx=None):
    """dst[idx[i]] = src[src_idx[i] if src_idx is given else i]"""
    _dev(src, idx, dst, src_idx)
    _same_dtype("scatter_rows", src, dst=dst)
    assert idx.dtype == torch.int32 and src.is_contiguous() and dst.is_contiguous()
    assert src_idx is None or (src_idx.dtype == torch.int32 and src_idx.numel() == idx.numel())
    n = idx.numel()
    if n == 0:
        return dst
    L.check(L.load().srgpt_scatter_rows(_p(src), _p(src_idx), _p(idx), _p(dst), n, src.shape[-1], dt_code(src), _stream()))
    return dst


def silu_mul(gu):
    _dev(gu)
    rows, two_i = gu.shape
    out = torch.empty((rows, two_i // 2), device=gu.device, dtype=gu.dtype)
    L.check(L.load().srgpt_silu_mul(_p(_c(gu)), _p(out), rows, two_i // 2, dt_code(gu), _stream()))
    return out


def argmax(logits):
    _dev(logits)
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    B, V = logits.shape
    out = torch.empty((B,), device=logits.device, dtype=torch.int64)
    L.check(L.load().srgpt_argmax(_p(logits), _p(out), B, V, _stream()))
    return out


def cross_entropy(shift_logits: torch.Tensor, shift_labels: torch.Tensor, ignore_index: int = -100):
    """mean cross entropy over the rows whose label is not `ignore_index`: shift_logits fp32 [rows, V] (already shifted),
    shift_labels int64 [rows].  Returns (loss 0-d fp32 tensor, number of targets 0-d tensor)."""
    _dev(shift_logits, shift_labels)
    if shift_logits.dtype != torch.float32 or shift_labels.dtype != torch.int64:
        raise ValueError("cross_entropy: logits must be fp32 and labels int64")
    rows, V = shift_logits.shape
    row_loss = torch.empty((rows,), device=shift_logits.device, dtype=torch.float32)
    out = torch.empty((2,), device=shift_logits.device, dtype=torch.float32)
    L.check(L.load().srgpt_cross_entropy(_p(_c(shift_logits)), _p(_c(shift_labels)), _p(row_loss), _p(out), rows, V,
                                         int(ignore_index), _stream()))
    return out[0], out[1]


class SamplingParams:
    """The device-resident parameter block of the device-side draw (srgpt_sampling, include/srgpt.h): temperature -> top-k ->
    top-p -> categorical, HF GenerationMixin.sample's warper chain.  `supported()` says whether a setting is served on the device
    (top_k 1..64 with any top_p; or no top-k and no top-p: pure temperature sampling by Gumbel-max)."""

    def __init__(self, device, batch: int, keep_kept_sets: bool = False):
        self.device = torch.device(device)
        self.buf = torch.zeros((C.sizeof(L.Sampling),), dtype=torch.uint8, device=self.device)
        self.kept = torch.zeros((batch, L.SAMPLING_KEPT_MAX + 1), dtype=torch.int32, device=self.device) if keep_kept_sets else None
        self.host = L.Sampling()

    @staticmethod
    def supported(temperature, top_k, top_p, vocab: int) -> bool:
        if temperature is None or not temperature > 0 or vocab > L.SAMPLING_VOCAB_MAX:
            return False
        k = 0 if top_k is None else int(top_k)
        p_on = top_p is not None and top_p < 1.0
        if k == 0:
            return not p_on
        return 1 <= k <= L.SAMPLING_TOP_K_MAX

    def set(self, temperature: float, top_k, top_p, seed: int, counter: int = 0):
        """one small H2D copy on the current stream"""
        h = self.host
        h.temperature = float(temperature)
        h.top_k = 0 if top_k is None else int(top_k)
        tp = 1.0 if top_p is None else float(top_p)
        h.top_p = tp
        h.top_p_rm = float(torch.tensor(1.0 - tp, dtype=torch.float64).to(torch.float32))  # HF compares against (1 - top_p) in fp32
        h.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        h.counter = int(counter)
        h.kept_out = None if self.kept is None else self.kept.data_ptr()
        src = torch.frombuffer(bytearray(bytes(h)), dtype=torch.uint8)
        self.buf.copy_(src, non_blocking=False)
        return self

    def ptr(self):
        return self.buf.data_ptr()


def sample(logits: torch.Tensor, params: SamplingParams, check: bool = False) -> torch.Tensor:
    """one draw per row of fp32 logits [B, V] with the device-side sampler; advances params' counter.  -> int64 [B]"""
    _dev(logits)
    if logits.dtype != torch.float32 or logits.ndim != 2:
        raise ValueError("sample: logits must be fp32 [B, V]")
    B, V = logits.shape
    lib = L.load()
    ws = torch.empty((lib.srgpt_sample_ws_bytes(B),), dtype=torch.uint8, device=logits.device)
    out = torch.empty((B,), dtype=torch.int64, device=logits.device)
    L.check(lib.srgpt_sample(_p(_c(logits)), params.ptr(), _p(out), _p(ws), B, V, _stream()))
    if check:  # settings the device sampler does not serve raise instead of drawing from a clamped distribution
        L.check(lib.srgpt_sample_status(_p(ws), B, _stream()))
    return out
