"""Weight layout in HBM.

Input: a state dict with the reference's checkpoint key names (llava/model/llava_arch.py:181-250:
sub-dirs llm/ vision_tower/ mm_projector/ region_extractor/; HF 4.37.2 key names inside).
Output: device tensors arranged for the kernels --
  * q/k/v rows concatenated ([q;k;v], one GEMM / one weight stream per layer), gate/up rows concatenated
  * patch-embed conv flattened to [C, 3*p*p] and zero-padded to a 16-byte multiple
  * ConvTranspose2d(k=2,s=2) weights [Cin,Cout,2,2] re-laid as [(a,b,co), ci] so the deconv is a GEMM whose
    output columns are contiguous channels of one output pixel
  * RoPE cos/sin tables (fp32 angle, cast to the model dtype -- modeling_llama.py:81-130)
and the ctypes structs (include/srgpt.h) that point at them.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import _lib as L
from .config import SrgptConfig

VT = "vision_tower.vision_tower.vision_model."
RE = "region_extractor."
MP = "mm_projector.layers."
LM = "llm."


def _ptr_array(tensors: List[torch.Tensor]):
    arr = (L.vp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def rope_tables(cfg: SrgptConfig, n_pos: int, dtype, device):
    d = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    pos = torch.arange(n_pos, dtype=torch.int64).float()
    if cfg.rope_factor != 1.0:
        # linear scaling (context_length_extension, language_model/builder.py:31-38): the reference's vendored
        # LlamaLinearScalingRotaryEmbedding divides the fp32 POSITIONS (modeling_llama.py:133-140); dividing inv_freq instead
        # rounds differently for factors that are not powers of two (pinned by tests/golden/vendored_llama_kat.npz)
        pos = pos / cfg.rope_factor
    freqs = pos[:, None] * inv_freq[None, :]  # fp32, as LlamaRotaryEmbedding (autocast disabled)
    return freqs.cos().to(dtype).to(device).contiguous(), freqs.sin().to(dtype).to(device).contiguous()


class PreparedWeights:
    """Owns the device tensors and the C structs pointing at them."""

    ALL_PARTS = ("vit", "region", "projector", "llm")

    def __init__(self, cfg: SrgptConfig, sd: Dict[str, torch.Tensor], device, dtype, rope_positions: int = 0,
                 consume: bool = False, llm_weight_format: str = "native", parts=None, decode_layout: str = "packed"):
        """llm_weight_format: "native" (the engine dtype) or "fp8" -- weight-only OCP e4m3fn quantisation of the five
        streamed LLM matrices with one fp32 (power-of-two) scale per output row (BASELINE config 5; bf16 engines only).  The
        fp8 bytes are the ONLY copy of those matrices in HBM: decode streams them (srgpt_gemv_w8), prefill multiplies them
        (srgpt_gemm_w8); `dequantised(name, layer)` rebuilds the bf16 values for checks."""
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        # parts: which components to materialise -- all of them for the model, ONE for the module-level factories
        # (spatialrgpt_amd/factories.py: the reference's build_vision_tower / build_mm_projector / build_region_extractor /
        # build_llm_and_tokenizer seams); a missing part's weights are neither read nor required
        self.parts = tuple(self.ALL_PARTS if parts is None else parts)
        for p_ in self.parts:
            if p_ not in self.ALL_PARTS:
                raise ValueError(f"unknown part {p_!r}")
        self.vit = self.llm = None
        self.vocab = 0
        self.rope_len = 0
        if llm_weight_format not in ("native", "fp8", "fp8_w8a8"):
            raise ValueError(f"unknown llm_weight_format {llm_weight_format!r}")
        # "fp8_w8a8": the same fp8 weights; prefill additionally quantises every GEMM input per token to e4m3 and multiplies on
        # the fp8 matrix pipe (srgpt_gemm_w8a8) -- opt-in, it changes the numbers; decode stays W8A16
        self.fp8_act = llm_weight_format == "fp8_w8a8"
        if self.fp8_act:
            llm_weight_format = "fp8"
            if cfg.hidden % 128 or cfg.inter % 128 or (cfg.heads * cfg.head_dim) % 128:
                raise ValueError("fp8_w8a8 needs hidden, inter and heads * head_dim to be multiples of 128")
        if llm_weight_format == "fp8" and dtype != torch.bfloat16:
            raise ValueError("fp8 LLM weights need a bf16 engine")
        self.llm_weight_format = llm_weight_format
        if decode_layout not in ("packed", "rowmajor"):
            raise ValueError(f"unknown decode_layout {decode_layout!r}")
        self.decode_layout = decode_layout  # fp8 weights: also keep the MFMA-operand-order copies the batched decode step streams
        self._keep: List[object] = []
        code = {torch.float32: L.F32, torch.bfloat16: L.BF16}[dtype]

        def get(name):
            t = sd.pop(name) if consume else sd[name]
            return t.to(device=self.device, dtype=dtype).contiguous()

        if "vit" in self.parts:
            self._prepare_vit(cfg, get, code)
        if "region" in self.parts and cfg.enable_region:
            self._prepare_region(cfg, get)
        if "projector" in self.parts:
            # ---------------- projector (mlp_downsample) ----------------
            self.mp_ln_w, self.mp_ln_b = get(MP + "1.weight"), get(MP + "1.bias")
            self.mp_w1, self.mp_b1 = get(MP + "2.weight"), get(MP + "2.bias")
            self.mp_w2, self.mp_b2 = get(MP + "4.weight"), get(MP + "4.bias")
        if "llm" in self.parts:
            self._prepare_llm(cfg, get, code, llm_weight_format, rope_positions)

    def _prepare_vit(self, cfg, get, code):
        dtype = self.dtype
        # ---------------- vision tower ----------------
        C_, p = cfg.vit_hidden, cfg.patch_size
        kk = 3 * p * p
        self.kp = (kk + 7) // 8 * 8
        pw = get(VT + "embeddings.patch_embedding.weight").reshape(C_, kk)
        patch_w = torch.zeros((C_, self.kp), device=self.device, dtype=dtype)
        patch_w[:, :kk] = pw
        self.patch_w = patch_w
        clip = cfg.tower == "clip"
        self.patch_b = None if clip else get(VT + "embeddings.patch_embedding.bias")
        self.pos_emb = get(VT + "embeddings.position_embedding.weight")
        if self.pos_emb.shape[0] != cfg.tower_tokens:
            raise ValueError(f"position embedding has {self.pos_emb.shape[0]} rows, tower expects {cfg.tower_tokens}")
        self.cls_emb = get(VT + "embeddings.class_embedding").reshape(-1) if clip else None
        self.pre_ln_w = get(VT + "pre_layrnorm.weight") if clip else None
        self.pre_ln_b = get(VT + "pre_layrnorm.bias") if clip else None
        n_run = cfg.vit_layers_run
        self.inter_pad = (cfg.vit_inter + 63) // 64 * 64
        v = {k: [] for k in ("ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2")}
        for i in range(n_run):
            q = f"{VT}encoder.layers.{i}."
            v["ln1_w"].append(get(q + "layer_norm1.weight"))
            v["ln1_b"].append(get(q + "layer_norm1.bias"))
            v["wqkv"].append(torch.cat([get(q + f"self_attn.{n}_proj.weight") for n in "qkv"], 0).contiguous())
            v["bqkv"].append(torch.cat([get(q + f"self_attn.{n}_proj.bias") for n in "qkv"], 0).contiguous())
            v["wo"].append(get(q + "self_attn.out_proj.weight"))
            v["bo"].append(get(q + "self_attn.out_proj.bias"))
            v["ln2_w"].append(get(q + "layer_norm2.weight"))
            v["ln2_b"].append(get(q + "layer_norm2.bias"))
            v["w1"].append(get(q + "mlp.fc1.weight"))
            v["b1"].append(get(q + "mlp.fc1.bias"))
            w2 = get(q + "mlp.fc2.weight")  # [hidden, inter] -> [hidden, inter_pad], zero columns (fc2's K tiles exactly by 64)
            if self.inter_pad != cfg.vit_inter:
                w2p = torch.zeros((C_, self.inter_pad), device=self.device, dtype=dtype)
                w2p[:, :cfg.vit_inter] = w2
                w2 = w2p
            v["w2"].append(w2)
            v["b2"].append(get(q + "mlp.fc2.bias"))
        self.vit_t = v
        vw = L.VitWeights()
        vw.dtype, vw.hidden, vw.inter, vw.heads = code, C_, cfg.vit_inter, cfg.vit_heads
        vw.n_layers_run, vw.image_size, vw.patch, vw.kp, vw.eps = n_run, cfg.image_size, p, self.kp, cfg.vit_eps
        vw.inter_pad = self.inter_pad
        vw.act = L.ACT_QUICK_GELU if clip else L.ACT_GELU_TANH
        vw.patch_w, vw.pos_emb = self.patch_w.data_ptr(), self.pos_emb.data_ptr()
        vw.patch_b = None if self.patch_b is None else self.patch_b.data_ptr()
        vw.cls_emb = None if self.cls_emb is None else self.cls_emb.data_ptr()
        vw.pre_ln_w = None if self.pre_ln_w is None else self.pre_ln_w.data_ptr()
        vw.pre_ln_b = None if self.pre_ln_b is None else self.pre_ln_b.data_ptr()
        for k, ts in v.items():
            arr = _ptr_array(ts)
            self._keep.append(arr)
            setattr(vw, k, arr)
        self.vit = vw

    def _prepare_region(self, cfg, get):
        # ---------------- region extractor ----------------
        fr = RE + "feature_refinement_module."

        def deconv_w(name):
            w = get(name)  # [Cin, Cout, 2, 2]
            cin, cout = w.shape[0], w.shape[1]
            return w.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous()

        self.dc1_w, self.dc1_b = deconv_w(fr + "0.weight"), get(fr + "0.bias")
        self.ln2d_w, self.ln2d_b = get(fr + "1.weight"), get(fr + "1.bias")
        self.dc2_w, self.dc2_b = deconv_w(fr + "3.weight"), get(fr + "3.bias")
        self.rgb_w, self.rgb_b = get(RE + "rgb_projector.weight"), get(RE + "rgb_projector.bias")
        self.depth_w, self.depth_b = get(RE + "depth_projector.weight"), get(RE + "depth_projector.bias")

    def _prepare_llm(self, cfg, get, code, llm_weight_format, rope_positions):
        dtype = self.dtype
        # ---------------- language model ----------------
        self.embed = get(LM + "model.embed_tokens.weight")
        self.final_norm = get(LM + "model.norm.weight")
        self.lm_head = get(LM + "lm_head.weight")
        lt = {k: [] for k in ("attn_norm", "wqkv", "wo", "mlp_norm", "wgu", "wdown")}
        for i in range(cfg.layers):
            q = f"{LM}model.layers.{i}."
            lt["attn_norm"].append(get(q + "input_layernorm.weight"))
            lt["wqkv"].append(torch.cat([get(q + f"self_attn.{n}_proj.weight") for n in "qkv"], 0).contiguous())
            lt["wo"].append(get(q + "self_attn.o_proj.weight"))
            lt["mlp_norm"].append(get(q + "post_attention_layernorm.weight"))
            lt["wgu"].append(torch.cat([get(q + "mlp.gate_proj.weight"), get(q + "mlp.up_proj.weight")], 0).contiguous())
            lt["wdown"].append(get(q + "mlp.down_proj.weight"))
        self.llm_t = lt
        self.llm_q = None
        if llm_weight_format == "fp8":
            from . import ops  # local import: ops needs the loaded library

            qd = {k: ([], []) for k in ("wqkv", "wo", "wgu", "wdown")}
            for k in qd:
                for i in range(cfg.layers):
                    q8, sc, _ = ops.quantize_fp8_rows(lt[k][i])
                    lt[k][i] = None  # the bf16 matrix is dropped: nothing reads it
                    qd[k][0].append(q8)
                    qd[k][1].append(sc)
            self.lm_head8, self.lm_head_scale, _ = ops.quantize_fp8_rows(self.lm_head)
            self.lm_head = None
            self.llm_q = qd
            # Packed copies for the batched decode step (2+ sequences per GPU; srgpt_pack_decode_weights, include/srgpt.h ABI 9): the
            # same fp8 bytes in MFMA-operand order.  The row-major copies stay: prefill (srgpt_gemm_w8 / _w8a8) and the one-row
            # decode kernel read those -- +6.5 GB of the 288 for an 8B model.  Granule rows per matrix from the measured table
            # (profiles/r06_skinny_packed.txt): 16 (1 KiB per instruction) where a CU owns a single 16-column tile, else 4.
            self.llm_pk, self.pk_rows = {}, {}
            if self.decode_layout == "packed":
                n_cu = int(L.load().srgpt_device_cus())
                for k in qd:
                    rows_total, K = qd[k][0][0].shape
                    n_out = rows_total // 2 if k == "wgu" else rows_total
                    rows = 16 if (n_out + n_cu - 1) // n_cu <= 16 else 4
                    if K % 64 or n_out % rows:
                        continue  # (tiny test geometries: this product streams the row-major copy)
                    self.llm_pk[k] = [ops.pack_decode_weights(q8, rows) for q8 in qd[k][0]]
                    self.pk_rows[k] = rows
        n_pos = rope_positions or cfg.max_position_embeddings
        self.rope_len = n_pos
        self.rope_cos, self.rope_sin = rope_tables(cfg, n_pos, dtype, self.device)
        lw = L.LlmWeights()
        lw.dtype, lw.hidden, lw.inter, lw.layers = code, cfg.hidden, cfg.inter, cfg.layers
        lw.heads, lw.kv_heads, lw.head_dim, lw.vocab, lw.rms_eps = cfg.heads, cfg.kv_heads, cfg.head_dim, self.embed.shape[0], cfg.rms_eps
        lw.rope_cos, lw.rope_sin = self.rope_cos.data_ptr(), self.rope_sin.data_ptr()
        lw.embed, lw.final_norm = self.embed.data_ptr(), self.final_norm.data_ptr()
        lw.lm_head = None if self.lm_head is None else self.lm_head.data_ptr()
        for k, ts in lt.items():
            arr = _ptr_array(ts)
            self._keep.append(arr)
            setattr(lw, k, arr)
        if self.llm_q is not None:
            lw.lm_head8, lw.lm_head_scale = self.lm_head8.data_ptr(), self.lm_head_scale.data_ptr()
            for k, (qs, scs) in self.llm_q.items():
                a8, asc = _ptr_array(qs), _ptr_array(scs)
                self._keep += [a8, asc]
                setattr(lw, k + "8", a8)
                setattr(lw, k + "_scale", asc)
            for k, field in (("wqkv", "pk_rows_qkv"), ("wo", "pk_rows_o"), ("wgu", "pk_rows_gu"), ("wdown", "pk_rows_down")):
                if k in self.llm_pk:
                    ap = _ptr_array(self.llm_pk[k])
                    self._keep.append(ap)
                    setattr(lw, k + "8p", ap)
                    setattr(lw, field, self.pk_rows[k])
        lw.fp8_act = 1 if self.fp8_act else 0
        self.llm = lw
        self.vocab = self.embed.shape[0]

    def resize_vocab(self, n: int) -> None:
        """HF `resize_token_embeddings` on the live weights (the reference's loader calls it after adding `<mask>` / `<depth>`,
        builder.py:186-199 -> PreTrainedModel._get_resized_embeddings / _get_resized_lm_head): embed_tokens and lm_head grow or shrink to
        `n` rows, the common rows are kept, new rows = the mean of the old rows -- what spatialrgpt_amd.builder gives the rows it adds
        at load (transformers' mean-resizing; 4.37.2 drew them N(0, initializer_range): untrained rows either way).
        fp8 lm_head: the new rows are quantised with the loader's rule and appended."""
        old = self.embed.shape[0]
        if n == old:
            return
        if n <= 0:
            raise ValueError(f"resize_token_embeddings({n})")
        keep = min(old, n)

        def resized(t):
            out = torch.empty((n, t.shape[1]), device=t.device, dtype=t.dtype)
            out[:keep] = t[:keep]
            if n > keep:
                out[keep:] = t.float().mean(0, keepdim=True).to(t.dtype)
            return out

        self.embed = resized(self.embed)
        lw = self.llm
        if self.llm_q is None:
            self.lm_head = resized(self.lm_head)
            lw.lm_head = self.lm_head.data_ptr()
        else:
            from . import ops
            q8, sc = self.lm_head8[:keep], self.lm_head_scale[:keep]
            if n > keep:
                fresh = self.dequantised("lm_head").float().mean(0, keepdim=True).to(self.dtype).expand(n - keep, -1).contiguous()
                nq, nsc, _ = ops.quantize_fp8_rows(fresh)
                q8, sc = torch.cat([q8, nq], 0), torch.cat([sc, nsc], 0)
            self.lm_head8, self.lm_head_scale = q8.contiguous(), sc.contiguous()
            lw.lm_head8, lw.lm_head_scale = self.lm_head8.data_ptr(), self.lm_head_scale.data_ptr()
        lw.embed, lw.vocab = self.embed.data_ptr(), n
        self.vocab = n

    def dequantised(self, name: str, layer: int = 0) -> torch.Tensor:
        """bf16 values of an fp8-held matrix ("wqkv" | "wo" | "wgu" | "wdown" | "lm_head"): code * scale, exact in bf16."""
        if self.llm_q is None:
            raise ValueError("weights are not fp8")
        q8, sc = (self.lm_head8, self.lm_head_scale) if name == "lm_head" else (self.llm_q[name][0][layer], self.llm_q[name][1][layer])
        return (q8.view(torch.float8_e4m3fn).float() * sc[:, None]).to(self.dtype)

    def llm_weight_bytes(self) -> int:
        """bytes streamed from HBM per decoded token at batch 1 (everything but embed_tokens)."""
        es = self.embed.element_size()
        if self.llm_q is None:
            n = self.final_norm.numel() + self.lm_head.numel()
            for ts in self.llm_t.values():
                n += sum(t.numel() for t in ts)
            return n * es
        n = (self.final_norm.numel() + sum(t.numel() for k in ("attn_norm", "mlp_norm") for t in self.llm_t[k])) * es
        n += self.lm_head8.numel() + 4 * self.lm_head_scale.numel()
        for qs, scs in self.llm_q.values():
            n += sum(t.numel() for t in qs) + 4 * sum(t.numel() for t in scs)
        return n


def weight_shapes(cfg: SrgptConfig) -> Dict[str, tuple]:
    C_, I, H, F_, V, d = cfg.vit_hidden, cfg.vit_inter, cfg.hidden, cfg.inter, cfg.vocab, cfg.head_dim
    s: Dict[str, tuple] = {}
    s[VT + "embeddings.patch_embedding.weight"] = (C_, 3, cfg.patch_size, cfg.patch_size)
    if cfg.tower == "clip":
        s[VT + "embeddings.class_embedding"] = (C_,)
        s[VT + "pre_layrnorm.weight"] = (C_,)
        s[VT + "pre_layrnorm.bias"] = (C_,)
    else:
        s[VT + "embeddings.patch_embedding.bias"] = (C_,)
    s[VT + "embeddings.position_embedding.weight"] = (cfg.tower_tokens, C_)
    for i in range(cfg.vit_layers_run):  # layers past select_layer never influence the output
        p = f"{VT}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (C_,)
            s[p + n + ".bias"] = (C_,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (C_, C_)
            s[p + f"self_attn.{n}.bias"] = (C_,)
        s[p + "mlp.fc1.weight"] = (I, C_)
        s[p + "mlp.fc1.bias"] = (I,)
        s[p + "mlp.fc2.weight"] = (C_, I)
        s[p + "mlp.fc2.bias"] = (C_,)
    p = RE + "feature_refinement_module."
    s[p + "0.weight"] = (C_, C_, 2, 2)
    s[p + "0.bias"] = (C_,)
    s[p + "1.weight"] = (C_,)
    s[p + "1.bias"] = (C_,)
    s[p + "3.weight"] = (C_, C_, 2, 2)
    s[p + "3.bias"] = (C_,)
    for n in ("rgb_projector", "depth_projector"):
        s[RE + n + ".weight"] = (H, C_)
        s[RE + n + ".bias"] = (H,)
    s[MP + "1.weight"] = (4 * C_,)
    s[MP + "1.bias"] = (4 * C_,)
    s[MP + "2.weight"] = (H, 4 * C_)
    s[MP + "2.bias"] = (H,)
    s[MP + "4.weight"] = (H, H)
    s[MP + "4.bias"] = (H,)
    s[LM + "model.embed_tokens.weight"] = (V, H)
    for i in range(cfg.layers):
        p = f"{LM}model.layers.{i}."
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
        s[p + "self_attn.q_proj.weight"] = (cfg.heads * d, H)
        s[p + "self_attn.k_proj.weight"] = (cfg.kv_heads * d, H)
        s[p + "self_attn.v_proj.weight"] = (cfg.kv_heads * d, H)
        s[p + "self_attn.o_proj.weight"] = (H, cfg.heads * d)
        s[p + "mlp.gate_proj.weight"] = (F_, H)
        s[p + "mlp.up_proj.weight"] = (F_, H)
        s[p + "mlp.down_proj.weight"] = (H, F_)
    s[LM + "model.norm.weight"] = (H,)
    s[LM + "lm_head.weight"] = (V, H)
    return s


def synth_state_dict(cfg: SrgptConfig, seed: int, dtype, device, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded random weights of the named architecture, generated on `device` (no checkpoints exist offline).
    Matrices/biases ~ N(0, std), norm gains 1 + N(0, std), embeddings ~ N(0, 0.5)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in weight_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        is_gain = name.endswith(("norm.weight", "layernorm.weight", "layrnorm.weight", "layer_norm1.weight", "layer_norm2.weight")) or \
            name in (RE + "feature_refinement_module.1.weight", MP + "1.weight")
        if "position_embedding" in name or "embed_tokens" in name or "class_embedding" in name:
            t = t * 0.5
        elif is_gain:
            t = 1.0 + t * std
        else:
            t = t * std
        sd[name] = t.to(dtype)
    return sd
