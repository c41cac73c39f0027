"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (spatialrgpt_amd/).

Self-contained CPU restatement (plain torch ops, no transformers, no reference import) of the
region-grounded multimodal forward/generate path of SpatialRGPT.  Only tests/,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file, and only as
the checker / baseline.

Pinning status: the reference ships no tests or golden vectors, so this
restatement is pinned against *outputs of the reference itself run in the build container*:
`oracle/make_golden.py` imports the real reference (via `oracle/ref_harness.py`), runs it on seeded
inputs, asserts this file reproduces every stage, and commits the vectors under tests/golden/.
The ViT arithmetic and the generation loops live in third-party `transformers` (pinned ==4.37.2 by the
reference's pyproject.toml:17; 5.15.0 is what is installed here) -- for those two pieces the pin is
"the reference's call sites executed against the installed transformers".
Round 4: the LLM (`llama_forward`, `rope_cos_sin`) is additionally pinned against the reference's VENDORED
llava/train/transformers_replace/models/llama/modeling_llama.py executed on CPU with oracle/flash_attn_cpu.py
standing in for the flash-attn extension (`make_golden.py vendored` -> tests/golden/vendored_llama_kat.npz:
max|d| = 0 in fp32 incl. linear RoPE scaling, cache steps beyond max_position_embeddings and the ragged
varlen branch); what stays restated-from-publication on that side is the flash-attn kernel's arithmetic.

Weight naming = the reference checkpoint layout (llava/model/llava_arch.py:181-250): prefixes
`llm.` (HF LlamaForCausalLM keys), `vision_tower.vision_tower.vision_model.` (HF SiglipVisionModel
4.37.2 keys), `mm_projector.layers.{1,2,4}.`, `region_extractor.`.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
import warnings
from dataclasses import asdict, dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200  # llava/constants.py:27
IGNORE_INDEX = -100  # llava/constants.py:26


@dataclass
class SrgptConfig:
    # vision tower (HF SiglipVisionConfig fields)
    vit_hidden: int = 1152
    vit_inter: int = 4304
    vit_layers: int = 27
    vit_heads: int = 16
    image_size: int = 384
    patch_size: int = 14
    vit_eps: float = 1e-6
    select_layer: int = -2  # scripts/srgpt/llama3_8b/3_sft.sh:30
    select_feature: str = "cls_patch"  # "patch" drops token 0 (vision_encoder.py:26-34)
    tower: str = "siglip"  # or "clip" (multimodal_encoder/clip_encoder.py: HF CLIPVisionModel)
    # language model (HF LlamaConfig fields)
    hidden: int = 4096
    inter: int = 14336
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    vocab: int = 128258
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 1.0  # linear scaling, language_model/builder.py:31-38
    # token stream
    mask_token_id: int = 128256
    depth_token_id: int = 128257
    enable_region: bool = True
    enable_depth: bool = True
    tokenizer_model_max_length: Optional[int] = None
    padding_side: str = "right"
    eos_token_id: Optional[int] = None

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def vit_head_dim(self) -> int:
        return self.vit_hidden // self.vit_heads

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    def to_dict(self):
        return asdict(self)


VT = "vision_tower.vision_tower.vision_model."
RE = "region_extractor."
MP = "mm_projector.layers."
LM = "llm."


# ------------------------------------------------------------------------------------------------
# A1  vision tower: VisionTower.forward + feature_select  (multimodal_encoder/vision_encoder.py:115-132,
#     26-34) over HF SiglipVisionModel (third-party; patch conv + learned pos-emb + pre-LN encoder).
# ------------------------------------------------------------------------------------------------

def vit_forward(w: Dict[str, torch.Tensor], cfg: SrgptConfig, images: torch.Tensor, collect_hidden: Optional[list] = None) -> torch.Tensor:
    """images [N,3,S,S] -> hidden_states[select_layer] = [N, grid^2, C] cast back to images.dtype.
    collect_hidden: a list that receives the input of every layer that runs and the last one's output (per-layer parity tests)."""
    wd = w[VT + "embeddings.patch_embedding.weight"].dtype
    clip = cfg.tower == "clip"
    x = F.conv2d(images.to(wd), w[VT + "embeddings.patch_embedding.weight"],
                 None if clip else w[VT + "embeddings.patch_embedding.bias"], stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)  # [N, L, C]
    if clip:  # HF CLIPVisionEmbeddings + pre_layrnorm (sic)
        cls = w[VT + "embeddings.class_embedding"].expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1)
    x = x + w[VT + "embeddings.position_embedding.weight"][None]
    if clip:
        x = F.layer_norm(x, (cfg.vit_hidden,), w[VT + "pre_layrnorm.weight"], w[VT + "pre_layrnorm.bias"], cfg.vit_eps)
    # hidden_states = (embeddings, out_0, ..., out_{L-1}); [-2] is the output of layer L-2
    n_run = cfg.vit_layers + 1 + cfg.select_layer if cfg.select_layer < 0 else cfg.select_layer
    H, hd = cfg.vit_heads, cfg.vit_head_dim
    for i in range(n_run):
        p = f"{VT}encoder.layers.{i}."
        if collect_hidden is not None:
            collect_hidden.append(x)
        r = x
        h = F.layer_norm(x, (cfg.vit_hidden,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], cfg.vit_eps)
        N, L, C = h.shape
        q = F.linear(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]).view(N, L, H, hd).transpose(1, 2)
        k = F.linear(h, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"]).view(N, L, H, hd).transpose(1, 2)
        v = F.linear(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"]).view(N, L, H, hd).transpose(1, 2)
        a = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
        a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(a, v).transpose(1, 2).reshape(N, L, C).contiguous()
        o = F.linear(o, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        x = r + o
        r = x
        h = F.layer_norm(x, (cfg.vit_hidden,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], cfg.vit_eps)
        h = F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])
        if clip:
            h = h * torch.sigmoid(1.702 * h)  # CLIP hidden_act = quick_gelu
        else:
            h = F.gelu(h, approximate="tanh")  # SigLIP hidden_act = gelu_pytorch_tanh
        h = F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        x = r + h
    if collect_hidden is not None:
        collect_hidden.append(x)
    if cfg.select_feature == "patch":
        x = x[:, 1:]
    elif cfg.select_feature != "cls_patch":
        raise ValueError(f"Unexpected select feature: {cfg.select_feature}")
    return x.to(images.dtype)  # vision_encoder.py:130


# ------------------------------------------------------------------------------------------------
# A2  RegionExtractor.feature_refinement (region_extractor/base_extractor.py:137-147, 87-101, 12-24)
# ------------------------------------------------------------------------------------------------

def layernorm2d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * x + bias[:, None, None]


def feature_refinement(w: Dict[str, torch.Tensor], tower_features: torch.Tensor):
    N, HW, C = tower_features.shape
    Hs = int(HW ** 0.5)
    x = tower_features.reshape(N, Hs, Hs, C).permute(0, 3, 1, 2)  # "N (H W) C -> N C H W"
    p = RE + "feature_refinement_module."
    x = F.conv_transpose2d(x, w[p + "0.weight"], w[p + "0.bias"], stride=2)
    x = layernorm2d(x, w[p + "1.weight"], w[p + "1.bias"])
    x = F.gelu(x)
    x = F.conv_transpose2d(x, w[p + "3.weight"], w[p + "3.bias"], stride=2)
    x = F.gelu(x)
    hres = x.flatten(2).transpose(1, 2)  # N (H W) C
    lres = F.adaptive_avg_pool2d(x, 27).flatten(2).transpose(1, 2)  # hard-coded 27, base_extractor.py:123
    return hres, lres


# ------------------------------------------------------------------------------------------------
# A3  MaskPooling.forward (base_extractor.py:32-84)   A4  RegionExtractor.forward (:149-173)
# ------------------------------------------------------------------------------------------------

def mask_pooling(x: torch.Tensor, mask_list: Optional[Sequence[Optional[torch.Tensor]]]) -> List[Optional[torch.Tensor]]:
    B = x.size(0)
    if mask_list is None:
        mask_list = [None] * B
    out: List[Optional[torch.Tensor]] = []
    for i in range(B):
        mask = mask_list[i]
        if mask is None:
            out.append(None)
            continue
        x_len = x.size(1)
        mask_hw = mask.size(-1) * mask.size(-2)
        scale_factor = (x_len / mask_hw) ** 0.5
        m = F.interpolate(mask.detach().float()[None], scale_factor=scale_factor, mode="bilinear")
        m = m.to(x.dtype)[0]
        denorm = (m.sum(dim=(-1, -2)) + 1e-8).unsqueeze(-1)
        m = m.flatten(start_dim=1)
        out.append(torch.einsum("lc,ml->mc", x[i], m / denorm))
    return out


def region_extractor(w, hres, depth_features, masks):
    def connect(embeds, name):
        return [None if e is None else F.linear(e, w[RE + name + ".weight"], w[RE + name + ".bias"]) for e in embeds]

    mask_embeds = connect(mask_pooling(hres, masks), "rgb_projector")
    depth_embeds = None
    if depth_features is not None:
        depth_embeds = connect(mask_pooling(depth_features, masks), "depth_projector")
    return mask_embeds, depth_embeds


# ------------------------------------------------------------------------------------------------
# A5  MultimodalProjector `mlp_downsample` (multimodal_projector/base_projector.py:32-94)
# ------------------------------------------------------------------------------------------------

def flat_square(x: torch.Tensor) -> torch.Tensor:
    n, w_, h_, c = x.size()
    if w_ % 2 == 1:
        x = torch.cat([x, torch.zeros((n, 1, h_, c), dtype=x.dtype)], dim=1).contiguous()
        n, w_, h_, c = x.size()
    if h_ % 2 == 1:
        x = torch.cat([x, torch.zeros((n, w_, 1, c), dtype=x.dtype)], dim=2).contiguous()
        n, w_, h_, c = x.size()
    x = x.view(n, w_, h_ // 2, c * 2)
    x = x.permute(0, 2, 1, 3).contiguous()
    x = x.view(n, h_ // 2, w_ // 2, c * 4)
    return x


def mm_projector(w: Dict[str, torch.Tensor], lres: torch.Tensor) -> torch.Tensor:
    n, L, c = lres.shape
    s = int(L ** 0.5)
    x = flat_square(lres.reshape(n, s, s, -1))
    x = x.reshape(n, -1, x.shape[-1])
    x = F.layer_norm(x, (x.shape[-1],), w[MP + "1.weight"], w[MP + "1.bias"], 1e-5)
    x = F.linear(x, w[MP + "2.weight"], w[MP + "2.bias"])
    x = F.gelu(x)
    x = F.linear(x, w[MP + "4.weight"], w[MP + "4.bias"])
    return x


# ------------------------------------------------------------------------------------------------
# A6  prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:333-650), inference subset
# ------------------------------------------------------------------------------------------------

def encode_visual(w, cfg: SrgptConfig, images, depths, masks):
    """llava_arch.py:387-411.  Returns (image_features, mask_embeds, depth_embeds, stages)."""
    if isinstance(images, list):
        images = torch.cat(images, dim=0)
    elif images.ndim == 5:
        images = images.flatten(0, 1)
    if depths is not None:
        if isinstance(depths, list):
            depths = torch.cat(depths, dim=0)
        elif depths.ndim == 5:
            depths = depths.flatten(0, 1)
    st = {}
    tower = vit_forward(w, cfg, images)
    st["tower_features"] = tower
    mask_embeds = depth_embeds = None
    if cfg.enable_region:
        hres, lres = feature_refinement(w, tower)
        st["hres"], st["lres"] = hres, lres
        depth_features = None
        if cfg.enable_depth and depths is not None:
            depth_features = vit_forward(w, cfg, depths)
            st["depth_features"] = depth_features
        mask_embeds, depth_embeds = region_extractor(w, hres, depth_features, masks)
    else:
        lres = tower
    image_features = mm_projector(w, lres)
    st["image_features"] = image_features
    return image_features, mask_embeds, depth_embeds, st


def splice(w, cfg: SrgptConfig, input_ids, attention_mask, image_features, mask_embeds, depth_embeds, have_depths,
           labels=None):
    """llava_arch.py:420-611.  Returns (inputs_embeds [B,T,H], attention_mask or None, position_ids or None); with `labels`
    ([B,P] int64) a 4th value: the spliced labels [B,T] (IGNORE_INDEX over image rows and padding, llava_arch.py:430-431,
    :446, :513-533, :558-611)."""
    _attention_mask = attention_mask
    _labels = labels
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    else:
        attention_mask = attention_mask.bool()
    ids0 = input_ids.clone()
    ids0[ids0 == IMAGE_TOKEN_INDEX] = 0
    input_embeds = F.embedding(ids0, w[LM + "model.embed_tokens.weight"])
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)
    ids_list = [i[m] for i, m in zip(input_ids, attention_mask)]
    emb_list = [e[m] for e, m in zip(input_embeds, attention_mask)]
    lab_list = [l[m] for l, m in zip(labels, attention_mask)]
    new_embeds = []
    new_labels = []
    cur_image_idx = 0
    for b, cur_ids in enumerate(ids_list):
        num_images = int((cur_ids == IMAGE_TOKEN_INDEX).sum())
        if num_images == 0:
            new_embeds.append(torch.cat([emb_list[b], image_features[0][0:0]], dim=0))
            new_labels.append(lab_list[b])
            continue
        cur = emb_list[b]
        img_idx = [-1] + torch.where(cur_ids == IMAGE_TOKEN_INDEX)[0].tolist() + [cur_ids.shape[0]]
        if cfg.enable_region:
            pos = cur_ids == cfg.mask_token_id
            num = int(pos.sum())
            me = mask_embeds[cur_image_idx]
            if me is None and num > 0:
                print("Error: mask embed is None, but the num of <mask> is not 0!!!")
            if me is not None:
                z = torch.zeros_like(cur)
                z[pos] = me[:num].to(dtype=z.dtype)
                cur = cur * (~pos).to(cur.dtype).unsqueeze(-1) + z
        if cfg.enable_depth and have_depths:
            pos = cur_ids == cfg.depth_token_id
            num = int(pos.sum())
            de = depth_embeds[cur_image_idx]
            if de is None and num > 0:
                print("Error: depth embed is None, but the num of <depth> is not 0!!!")
            if de is not None:
                z = torch.zeros_like(cur)
                z[pos] = de[:num].to(dtype=z.dtype)
                cur = cur * (~pos).to(cur.dtype).unsqueeze(-1) + z
        pieces, lab_pieces = [], []
        for i in range(num_images + 1):
            pieces.append(cur[img_idx[i] + 1: img_idx[i + 1]])
            lab_pieces.append(lab_list[b][img_idx[i] + 1: img_idx[i + 1]])
            if i < num_images:
                pieces.append(image_features[cur_image_idx])
                lab_pieces.append(torch.full((image_features[cur_image_idx].shape[0],), IGNORE_INDEX, dtype=labels.dtype))
                cur_image_idx += 1
        new_embeds.append(torch.cat(pieces))
        new_labels.append(torch.cat(lab_pieces))
    mx = cfg.tokenizer_model_max_length
    if mx is not None:
        if any(len(x) > mx for x in new_embeds):
            warnings.warn("Inputs truncated!")
        new_embeds = [x[:mx] for x in new_embeds]
        new_labels = [x[:mx] for x in new_labels]
    max_len = max(x.shape[0] for x in new_embeds)
    B = len(new_embeds)
    am = torch.zeros((B, max_len), dtype=torch.bool)
    pid = torch.zeros((B, max_len), dtype=torch.long)
    lab = torch.full((B, max_len), IGNORE_INDEX, dtype=labels.dtype)
    padded = []
    for i, e in enumerate(new_embeds):
        n = e.shape[0]
        z = torch.zeros((max_len - n, e.shape[1]), dtype=e.dtype)
        if cfg.padding_side == "left":
            padded.append(torch.cat((z, e), dim=0))
            if n > 0:
                am[i, -n:] = True
                pid[i, -n:] = torch.arange(n)
                lab[i, -n:] = new_labels[i]
        else:
            padded.append(torch.cat((e, z), dim=0))
            if n > 0:
                am[i, :n] = True
                pid[i, :n] = torch.arange(n)
                lab[i, :n] = new_labels[i]
    out = torch.stack(padded, dim=0)
    am_out = None if _attention_mask is None else am.to(_attention_mask.dtype)
    if _labels is not None:
        return out, am_out, pid, lab
    return out, am_out, pid


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """LlamaForCausalLM loss (modeling_llama.py:1047-1058): logits.float(), shift by one, mean cross entropy over the
    positions whose label is not IGNORE_INDEX."""
    shift_logits = logits.float()[..., :-1, :].contiguous()
    shift_labels = labels[..., 1:].contiguous()
    return F.cross_entropy(shift_logits.view(-1, shift_logits.shape[-1]), shift_labels.view(-1), ignore_index=IGNORE_INDEX)


# ------------------------------------------------------------------------------------------------
# A7-A12  Llama decoder (llava/train/transformers_replace/models/llama/modeling_llama.py; same math as
#         upstream HF LlamaForCausalLM with eager attention, which is what the reference oracle runs)
# ------------------------------------------------------------------------------------------------

def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:  # modeling_llama.py:61-75
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)


def rope_cos_sin(cfg: SrgptConfig, position_ids: torch.Tensor, dtype):  # modeling_llama.py:81-140
    d = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    pos = position_ids.float()
    if cfg.rope_factor != 1.0:
        # LlamaLinearScalingRotaryEmbedding.forward (modeling_llama.py:133-140): the POSITIONS are divided, in fp32, before the
        # product with inv_freq.  (transformers >= 4.45 divides inv_freq instead: the same angles only for power-of-two factors;
        # context_length_extension, language_model/builder.py:31-38, produces any integer factor.)
        pos = pos / cfg.rope_factor
    freqs = (inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
             @ pos[:, None, :]).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):  # modeling_llama.py:160-191
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


class KVCache:
    def __init__(self, n_layers):
        self.k = [None] * n_layers
        self.v = [None] * n_layers

    def update(self, i, k, v):
        if self.k[i] is None:
            self.k[i], self.v[i] = k, v
        else:
            self.k[i] = torch.cat([self.k[i], k], dim=2)  # modeling_llama.py:451-456
            self.v[i] = torch.cat([self.v[i], v], dim=2)
        return self.k[i], self.v[i]

    def seq_len(self):
        return 0 if self.k[0] is None else self.k[0].shape[2]


def llama_forward(w, cfg: SrgptConfig, inputs_embeds, position_ids, kv: KVCache, key_padding_mask=None,
                  last_only=False, collect_hidden=False, act_quant=None):
    """inputs_embeds [B,T,H]; key_padding_mask [B, past+T] bool (True = attend) or None.
    Returns logits fp32 [B,T,V] (or [B,1,V] when last_only) (+ list of hidden states).
    act_quant: None (the reference's arithmetic) or a callable applied to the input of the seven layer projections -- the
    restatement of the opt-in W8A8 prefill (fp8_rowwise_fake_quant below; no counterpart in the reference)."""
    lin = F.linear if act_quant is None else (lambda t, W: F.linear(act_quant(t), W))
    B, T, Hd = inputs_embeds.shape
    nh, nkv, d = cfg.heads, cfg.kv_heads, cfg.head_dim
    past = kv.seq_len()
    cos, sin = rope_cos_sin(cfg, position_ids, inputs_embeds.dtype)
    # causal mask (+ padding), additive, in the activation dtype (HF eager path)
    total = past + T
    neg = torch.finfo(inputs_embeds.dtype).min
    qpos = torch.arange(past, total)[:, None]
    kpos = torch.arange(total)[None, :]
    allowed = (kpos <= qpos)[None, None].expand(B, 1, T, total)
    if key_padding_mask is not None:
        allowed = allowed & key_padding_mask[:, None, None, :total].bool()
    mask = torch.zeros((B, 1, T, total), dtype=inputs_embeds.dtype).masked_fill(~allowed, neg)
    x = inputs_embeds
    hiddens = [x] if collect_hidden else None
    for i in range(cfg.layers):
        p = f"{LM}model.layers.{i}."
        r = x
        h = rmsnorm(x, w[p + "input_layernorm.weight"], cfg.rms_eps)
        q = lin(h, w[p + "self_attn.q_proj.weight"]).view(B, T, nh, d).transpose(1, 2)
        k = lin(h, w[p + "self_attn.k_proj.weight"]).view(B, T, nkv, d).transpose(1, 2)
        v = lin(h, w[p + "self_attn.v_proj.weight"]).view(B, T, nkv, d).transpose(1, 2)
        q, k = apply_rope(q, k, cos, sin)
        k, v = kv.update(i, k, v)
        rep = nh // nkv
        kk = k[:, :, None].expand(B, nkv, rep, total, d).reshape(B, nh, total, d)
        vv = v[:, :, None].expand(B, nkv, rep, total, d).reshape(B, nh, total, d)
        a = torch.matmul(q, kk.transpose(2, 3)) * (d ** -0.5) + mask
        a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(a, vv).transpose(1, 2).contiguous().reshape(B, T, nh * d)
        o = lin(o, w[p + "self_attn.o_proj.weight"])
        x = r + o
        r = x
        h = rmsnorm(x, w[p + "post_attention_layernorm.weight"], cfg.rms_eps)
        g = lin(h, w[p + "mlp.gate_proj.weight"])
        u = lin(h, w[p + "mlp.up_proj.weight"])
        x = r + lin(F.silu(g) * u, w[p + "mlp.down_proj.weight"])  # modeling_llama.py:221
        if collect_hidden:
            hiddens.append(x)
    x = rmsnorm(x, w[LM + "model.norm.weight"], cfg.rms_eps)
    if last_only:
        x = x[:, -1:, :]
    logits = F.linear(x, w[LM + "lm_head.weight"]).float()  # modeling_llama.py:1044-1045
    if collect_hidden:
        return logits, hiddens
    return logits


# ------------------------------------------------------------------------------------------------
# A0 / A13  LlavaLlamaModel.generate (language_model/llava_llama.py:194-213) + HF greedy loop
# ------------------------------------------------------------------------------------------------

@torch.no_grad()
def prepare_inputs(w, cfg, input_ids, images, depths=None, masks=None, attention_mask=None):
    image_features, mask_embeds, depth_embeds, st = encode_visual(w, cfg, images, depths, masks)
    embeds, am, pid = splice(w, cfg, input_ids, attention_mask, image_features, mask_embeds, depth_embeds,
                             have_depths=depths is not None)
    st["mask_embeds"] = mask_embeds
    st["depth_embeds"] = depth_embeds
    st["inputs_embeds"] = embeds
    return embeds, am, pid, st


@torch.no_grad()
def generate(w, cfg: SrgptConfig, input_ids, images, depths=None, masks=None, attention_mask=None,
             max_new_tokens=16, eos_token_id=None, return_stages=False, model_dtype=None, prefill_act_quant=None):
    """Greedy decode; returns only the new ids [B,G].  Finished rows are padded with eos.
    prefill_act_quant: None (the reference's arithmetic) or the activation quantiser of the opt-in W8A8 prefill
    (fp8_rowwise_fake_quant), applied to the PREFILL pass only -- decode steps stay W8A16, as in the engine."""
    embeds, am, pid, st = prepare_inputs(w, cfg, input_ids, images, depths, masks, attention_mask)
    if model_dtype is not None:
        embeds = embeds.to(model_dtype)  # llava_llama.py:210
    B, T, _ = embeds.shape
    kpm = torch.ones((B, T), dtype=torch.bool) if am is None else am.bool()
    kv = KVCache(cfg.layers)
    # HF: position_ids = cumsum(mask) - 1, pads set to 1 (modeling_llama.py:1127-1133)
    pos = kpm.long().cumsum(-1) - 1
    pos = pos.masked_fill(~kpm, 1)
    logits = llama_forward(w, cfg, embeds, pos, kv, key_padding_mask=kpm, last_only=False, act_quant=prefill_act_quant)
    st["prefill_logits"] = logits
    # last *valid* position per row is only well defined for left padding / no padding; HF takes [:, -1]
    nxt = logits[:, -1].argmax(-1)
    out = [nxt]
    step_logits = [logits[:, -1]]
    done = torch.zeros(B, dtype=torch.bool)
    if eos_token_id is not None:
        done |= nxt == eos_token_id
    for _ in range(max_new_tokens - 1):
        if bool(done.all()):
            break
        kpm = torch.cat([kpm, torch.ones((B, 1), dtype=torch.bool)], dim=1)
        pos = (kpm.long().sum(-1, keepdim=True) - 1)
        e = F.embedding(nxt[:, None], w[LM + "model.embed_tokens.weight"])
        logits = llama_forward(w, cfg, e, pos, kv, key_padding_mask=kpm, last_only=True)
        nxt = logits[:, -1].argmax(-1)
        step_logits.append(logits[:, -1])
        if eos_token_id is not None:
            nxt = torch.where(done, torch.full_like(nxt, eos_token_id), nxt)
            done |= nxt == eos_token_id
        out.append(nxt)
    ids = torch.stack(out, dim=1)
    if return_stages:
        st["step_logits"] = torch.stack(step_logits, dim=1)
        return ids, st
    return ids


# ------------------------------------------------------------------------------------------------
# deterministic synthetic weights / inputs, shared by tests and bench cpu_baseline
# ------------------------------------------------------------------------------------------------

def weight_shapes(cfg: SrgptConfig) -> Dict[str, tuple]:
    C, I, H, F_, V = cfg.vit_hidden, cfg.vit_inter, cfg.hidden, cfg.inter, cfg.vocab
    d = cfg.head_dim
    s: Dict[str, tuple] = {}
    s[VT + "embeddings.patch_embedding.weight"] = (C, 3, cfg.patch_size, cfg.patch_size)
    if cfg.tower == "clip":
        s[VT + "embeddings.class_embedding"] = (C,)
        s[VT + "pre_layrnorm.weight"] = (C,)
        s[VT + "pre_layrnorm.bias"] = (C,)
    else:
        s[VT + "embeddings.patch_embedding.bias"] = (C,)
    s[VT + "embeddings.position_embedding.weight"] = (cfg.grid ** 2 + (1 if cfg.tower == "clip" else 0), C)
    for i in range(cfg.vit_layers):
        p = f"{VT}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (C,)
            s[p + n + ".bias"] = (C,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (C, C)
            s[p + f"self_attn.{n}.bias"] = (C,)
        s[p + "mlp.fc1.weight"] = (I, C)
        s[p + "mlp.fc1.bias"] = (I,)
        s[p + "mlp.fc2.weight"] = (C, I)
        s[p + "mlp.fc2.bias"] = (C,)
    p = RE + "feature_refinement_module."
    s[p + "0.weight"] = (C, C, 2, 2)
    s[p + "0.bias"] = (C,)
    s[p + "1.weight"] = (C,)
    s[p + "1.bias"] = (C,)
    s[p + "3.weight"] = (C, C, 2, 2)
    s[p + "3.bias"] = (C,)
    for n in ("rgb_projector", "depth_projector"):
        s[RE + n + ".weight"] = (H, C)
        s[RE + n + ".bias"] = (H,)
    s[MP + "1.weight"] = (4 * C,)
    s[MP + "1.bias"] = (4 * C,)
    s[MP + "2.weight"] = (H, 4 * C)
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


def synth_weights(cfg: SrgptConfig, seed: int = 0, dtype=torch.float32, device="cpu", std: float = 0.02):
    """Seeded synthetic weights: N(0, std) for matrices/biases, norm gains 1 + N(0, std) (so a missing
    gain shows up in tests)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = {}
    for name, shape in weight_shapes(cfg).items():
        is_gain = name.endswith("norm.weight") or name.endswith("layernorm.weight") or name.endswith("layrnorm.weight") or \
            name.endswith("layer_norm1.weight") or name.endswith("layer_norm2.weight") or \
            name in (RE + "feature_refinement_module.1.weight", MP + "1.weight")
        t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        if is_gain:
            t = t + 1.0
        if "position_embedding" in name or "embed_tokens" in name or "class_embedding" in name:
            t = t * (1.0 / std) * 0.5  # O(1)-ish embeddings keep activations away from denormal land
        w[name] = t.to(dtype).to(device)
    return w


def synth_inputs(cfg: SrgptConfig, *, batch=1, regions=8, prompt_len=64, seed=1, image_hw=None, dtype=torch.float32):
    """SURVEY 8d synthetic request: images/depths ~ clipped randn, K random box masks, ids with one
    <image> sentinel and K (<mask>,<depth>) pairs."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    S = image_hw or cfg.image_size
    # quantised to multiples of 1/32 so the same pixels are exact in fp32 / bf16 and store as int8
    images = (torch.randn((batch, 3, S, S), generator=g).clamp_(-1, 1) * 32).round().div(32).to(dtype)
    depth1 = (torch.randn((batch, 1, S, S), generator=g).clamp_(-1, 1) * 32).round().div(32)
    depths = depth1.expand(batch, 3, S, S).contiguous().to(dtype)  # 3-channel grayscale copy
    masks = []
    for _ in range(batch):
        m = torch.zeros((regions, S, S))
        for r in range(regions):
            hh = int(torch.randint(S // 8, S // 2 + 1, (1,), generator=g))
            ww = int(torch.randint(S // 8, S // 2 + 1, (1,), generator=g))
            y0 = int(torch.randint(0, S - hh + 1, (1,), generator=g))
            x0 = int(torch.randint(0, S - ww + 1, (1,), generator=g))
            m[r, y0:y0 + hh, x0:x0 + ww] = 1.0
        masks.append(m.to(dtype))
    n_special = 1 + 2 * regions
    n_text = prompt_len - n_special - 1
    assert n_text >= regions + 2, "prompt_len too small for the requested number of regions"
    hi = min(cfg.mask_token_id, cfg.depth_token_id, cfg.vocab)
    ids = torch.empty((batch, prompt_len), dtype=torch.long)
    for b in range(batch):
        txt = torch.randint(3, hi, (n_text,), generator=g).tolist()
        pa = max(1, n_text // 4)
        per = max(1, (n_text - pa) // (regions + 1))
        seq = [1] + txt[:pa] + [IMAGE_TOKEN_INDEX]
        cur = pa
        for r in range(regions):
            seq += txt[cur:cur + per] + [cfg.mask_token_id, cfg.depth_token_id]
            cur += per
        seq += txt[cur:]
        assert len(seq) == prompt_len, (len(seq), prompt_len)
        ids[b] = torch.tensor(seq)
    return ids, images, depths, masks


def fp8_rowwise_fake_quant(x: torch.Tensor) -> torch.Tensor:
    """The activation side of the opt-in W8A8 prefill (BASELINE config 5 "fp8 MFMA"; no counterpart in the reference): every
    row (token) of x is replaced by dequant(quant(row)) -- scale = the smallest power of two with max|row| / scale <= 448, codes =
    round-to-nearest-even OCP e4m3fn -- returned in x.dtype (code * 2^k is exact in bf16).  Passed as llama_forward(act_quant=...)
    together with fp8_dequantised_weights it restates what srgpt_quant_rows_e4m3 + srgpt_gemm_w8a8 compute: fp32 accumulation of
    exact products, one rounding to the activation dtype per projection."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -100)
    m, e = torch.frexp(amax)                       # amax = m * 2^e, m in [0.5, 1)
    k = e - 9 + (m > 0.875).to(e.dtype)            # 448 = 0.875 * 2^9
    q = (xf * torch.ldexp(torch.ones_like(amax), -k)).to(torch.float8_e4m3fn)
    return (q.float() * torch.ldexp(torch.ones_like(amax), k)).to(x.dtype)


def fp8_dequantised_weights(w: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Weight-only fp8 option (BASELINE config 5; no counterpart in the reference, which offers bitsandbytes int8 at
    builder.py:51-52): every LLM Linear weight (q/k/v/o/gate/up/down projections and lm_head) is replaced by
    dequant(quant(W)) with one power-of-two fp32 scale per output row -- the smallest 2^k with max|row| / 2^k <= 448 --
    and codes = round-to-nearest-even OCP e4m3fn (torch.float8_e4m3fn).  The oracle then runs unchanged on these weights."""
    out = dict(w)
    for k, t in w.items():
        if not k.startswith("llm."):
            continue
        if k.endswith("_proj.weight") or k == "llm.lm_head.weight":
            tf = t.float()
            amax = tf.abs().amax(dim=1).clamp_min(2.0 ** -100)
            kk = torch.ceil(torch.log2(amax.double() / 448.0)).to(torch.int32)  # double: exact enough to place the power
            sc = torch.ldexp(torch.ones_like(amax), kk)
            # guard the log2 boundary: the scale must satisfy amax/sc <= 448 < 2*amax/sc... (exact check, exact fix)
            sc = torch.where(amax / sc > 448.0, sc * 2, sc)
            sc = torch.where(amax / (sc / 2) <= 448.0, sc / 2, sc)
            q = (tf / sc[:, None]).to(torch.float8_e4m3fn)
            out[k] = (q.float() * sc[:, None]).to(t.dtype)
    return out
