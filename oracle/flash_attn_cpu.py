"""TEST INFRASTRUCTURE ONLY -- a CPU stand-in for the `flash_attn` package, installed into sys.modules by
`oracle/ref_harness.install_vendored_llama()` so that the reference's VENDORED Llama
(llava/train/transformers_replace/models/llama/modeling_llama.py, whose decoder layer hard-wires `LlamaFlashAttention2`,
:611-619) can execute on the build container's CPU.  flash-attn (Dao-AILab/flash-attention, pinned to the v2.5.8 wheel by the
reference's environment_setup.sh:20-21) is a CUDA extension absent from /root/reference and from this image, so its PUBLISHED semantics are
restated here in plain torch; everything around it -- projections, the linear-scaling rotary classes, the cache concat, the
unpadding logic, RMSNorm, the MLP, LlamaModel / LlamaForCausalLM.forward -- is the reference's own code, loaded from where it lies.

Semantics restated (flash_attn/flash_attn_interface.py docstrings of v2.5.8):
  flash_attn_func(q, k, v, dropout_p, softmax_scale=None, causal=False): q [B, Sq, Hq, D], k / v [B, Sk, Hkv, D] (MQA / GQA: query
    head h uses kv head h // (Hq / Hkv)); out [B, Sq, Hq, D] = softmax(q k^T * scale [+ causal]) v; scale defaults to D ** -0.5;
    since v2.1 the causal mask is aligned to the BOTTOM-RIGHT corner: query i sees key j iff j <= i + (Sk - Sq).
    Numerics of the kernel: scores and softmax in fp32, P rounded to the input dtype for the P.V product, fp32 accumulation, one
    final rounding to the input dtype.
  flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, ...): the same over packed
    sequences; sequence s owns rows cu_seqlens[s] : cu_seqlens[s + 1].
  bert_padding.index_first_axis / unpad_input / pad_input: row gather / mask-driven unpadding / its inverse.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _attend(q, k, v, softmax_scale, causal):
    """q [Sq, Hq, D], k / v [Sk, Hkv, D] -> [Sq, Hq, D]"""
    Sq, Hq, D = q.shape
    Sk, Hkv, _ = k.shape
    rep = Hq // Hkv
    scale = D ** -0.5 if softmax_scale is None else softmax_scale
    kk = k[:, :, None, :].expand(Sk, Hkv, rep, D).reshape(Sk, Hq, D)
    vv = v[:, :, None, :].expand(Sk, Hkv, rep, D).reshape(Sk, Hq, D)
    s = torch.einsum("qhd,khd->hqk", q.float(), kk.float()) * scale
    if causal:
        qi = torch.arange(Sq)[:, None]
        kj = torch.arange(Sk)[None, :]
        s = s.masked_fill(~(kj <= qi + (Sk - Sq))[None], float("-inf"))
    p = torch.softmax(s, dim=-1).to(q.dtype)
    return torch.einsum("hqk,khd->qhd", p.float(), vv.float()).to(q.dtype)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **unused):
    assert dropout_p == 0.0, "inference only"
    return torch.stack([_attend(q[b], k[b], v[b], softmax_scale, causal) for b in range(q.shape[0])], dim=0)


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0, softmax_scale=None,
                           causal=False, **unused):
    assert dropout_p == 0.0, "inference only"
    out = torch.empty_like(q)
    for s in range(cu_seqlens_q.numel() - 1):
        q0, q1 = int(cu_seqlens_q[s]), int(cu_seqlens_q[s + 1])
        k0, k1 = int(cu_seqlens_k[s]), int(cu_seqlens_k[s + 1])
        out[q0:q1] = _attend(q[q0:q1], k[k0:k1], v[k0:k1], softmax_scale, causal)
    return out


def index_first_axis(x, indices):
    return x[indices]


def unpad_input(hidden_states, attention_mask):
    seqlens = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    cu = F.pad(torch.cumsum(seqlens, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape(-1, *hidden_states.shape[2:])
    return flat[indices], indices, cu, int(seqlens.max())


def pad_input(hidden_states, indices, batch, seqlen):
    out = torch.zeros((batch * seqlen, *hidden_states.shape[1:]), dtype=hidden_states.dtype, device=hidden_states.device)
    out[indices] = hidden_states
    return out.reshape(batch, seqlen, *hidden_states.shape[1:])
