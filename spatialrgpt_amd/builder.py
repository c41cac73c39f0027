"""`load_pretrained_model` -- the loader every reference caller uses (llava/model/builder.py:36-213, plain
branch :141-159,183-213) for the reference's on-disk layout (llava_arch.py:181-250):

    <model_path>/config.json                      LlavaConfig (enable_region, enable_depth, mm_* fields)
    <model_path>/llm/               config.json + *.safetensors + tokenizer files   (HF LlamaForCausalLM)
    <model_path>/vision_tower/      config.json + *.safetensors + preprocessor_config.json (HF SiglipVisionModel)
    <model_path>/mm_projector/      *.safetensors   (layers.{1,2,4}.*)
    <model_path>/region_extractor/  *.safetensors

Differences from the reference, by design: weights are materialised directly in bf16 on the MI355X (the
reference builds fp16 and its SpatialRGPT callers then cast to bf16, eval_spatial.py:221); 8-bit / 4-bit / LoRA
/ MPT branches are out of scope and raise.
"""
from __future__ import annotations

import glob
import json
import os

import torch

from .config import SrgptConfig
from .constants import DEFAULT_DEPTH_TOKEN, DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_MASK_TOKEN
from .generation import generation_config_from_files
from .mm_utils import SrgptImageProcessor


def _read_json(path):
    with open(path) as f:
        return json.load(f)


def _load_dir(d, prefix, out):
    from safetensors.torch import load_file

    files = sorted(glob.glob(os.path.join(d, "*.safetensors")))
    if not files:
        bins = sorted(glob.glob(os.path.join(d, "*.bin")))
        if not bins:
            raise ValueError(f"no weights found under {d}")
        for b in bins:
            for k, v in torch.load(b, map_location="cpu").items():
                out[prefix + k] = v
        return
    for f in files:
        for k, v in load_file(f).items():
            out[prefix + k] = v


def _sub_config(model_path: str, sub: str, nested) -> dict:
    """A sub-model's config: <ckpt>/<sub>/config.json (what `save_pretrained` of the sub-model writes, llava_arch.py:181-250);
    the copy nested in the top-level config.json (`llm_cfg` / `vision_tower_cfg` objects, llava_arch.py:201,217) is the
    fallback when the directory carries weights only."""
    p = os.path.join(model_path, sub, "config.json")
    if os.path.exists(p):
        return _read_json(p)
    if isinstance(nested, dict):
        return nested
    raise ValueError(f"no config for {sub!r} under {model_path}")


def _rope(lc: dict):
    """(theta, linear factor): transformers 4.37.2 stores `rope_theta` + `rope_scaling{type, factor}` (what the reference's
    context_length_extension writes, language_model/builder.py:31-38); transformers 5 folds both into `rope_parameters`."""
    rp = lc.get("rope_parameters") or {}
    theta = lc.get("rope_theta", rp.get("rope_theta", 10000.0))
    rs = lc.get("rope_scaling") or rp
    kind = rs.get("type", rs.get("rope_type"))
    return float(theta), (float(rs.get("factor", 1.0)) if kind == "linear" else 1.0)


def config_from_checkpoint(model_path: str) -> SrgptConfig:
    top = _read_json(os.path.join(model_path, "config.json"))
    lc = _sub_config(model_path, "llm", top.get("llm_cfg"))
    vc = _sub_config(model_path, "vision_tower", top.get("vision_tower_cfg"))
    vc = vc.get("vision_config", vc)
    arch = (vc.get("architectures") or [vc.get("model_type", "")])[0].lower() if (vc.get("architectures") or vc.get("model_type")) else ""
    tname = (arch + " " + str(vc.get("model_type", ""))).lower()
    if "siglip" in tname:
        tower = "siglip"
    elif "clip" in tname:
        tower = "clip"
    else:
        raise ValueError(f"Unknown vision tower: {arch or vc.get('model_type')}")  # multimodal_encoder/builder.py:46
    if top.get("mm_projector_cfg", {}).get("mm_projector_type", "mlp_downsample") != "mlp_downsample" \
            if isinstance(top.get("mm_projector_cfg"), dict) else False:
        raise ValueError(f"Unknown projector type: {top['mm_projector_cfg']}")
    theta, factor = _rope(lc)
    # HF `from_pretrained` gives the LLM `<ckpt>/llm/generation_config.json` as its generation config when the file exists (else the
    # generation fields of llm/config.json); `llm.generate` (llava_llama.py:212) stops on ITS eos ids -- a list for Llama-3
    gp = os.path.join(model_path, "llm", "generation_config.json")
    gen = generation_config_from_files(lc, _read_json(gp) if os.path.exists(gp) else None)
    return SrgptConfig(
        vit_hidden=vc["hidden_size"], vit_inter=vc["intermediate_size"], vit_layers=vc["num_hidden_layers"],
        vit_heads=vc["num_attention_heads"], image_size=vc["image_size"], patch_size=vc["patch_size"],
        vit_eps=vc.get("layer_norm_eps", 1e-6 if tower == "siglip" else 1e-5), select_layer=top.get("mm_vision_select_layer", -2) or -2,
        select_feature=top.get("mm_vision_select_feature", "cls_patch") or "cls_patch", tower=tower,
        hidden=lc["hidden_size"], inter=lc["intermediate_size"], layers=lc["num_hidden_layers"],
        heads=lc["num_attention_heads"], kv_heads=lc.get("num_key_value_heads", lc["num_attention_heads"]),
        vocab=lc["vocab_size"], rms_eps=lc.get("rms_norm_eps", 1e-5), rope_theta=theta,
        rope_factor=factor, max_position_embeddings=int(lc.get("model_max_length") or lc.get("max_position_embeddings", 4096)),
        enable_region=bool(top.get("enable_region", False)), enable_depth=bool(top.get("enable_depth", False)),
        tokenizer_model_max_length=lc.get("tokenizer_model_max_length"),
        padding_side=lc.get("tokenizer_padding_side", "right"),
        eos_token_id=gen.get("eos_token_id"), pad_token_id=gen.get("pad_token_id"), generation_config=gen,
        image_aspect_ratio=top.get("image_aspect_ratio", "resize") or "resize",
        mm_use_im_start_end=bool(top.get("mm_use_im_start_end", False)),
        mm_use_im_patch_token=bool(top.get("mm_use_im_patch_token", True)),
    )


def resize_position_embeddings(pos_emb: torch.Tensor, num_new_tokens: int) -> torch.Tensor:
    """`VisionTower._maybe_resize_pos_embeds`, interpolate_mode "linear" (multimodal_encoder/vision_encoder.py:36-113): the
    learned position table [M, C] is resampled to N = (resolution // patch)^2 rows along the FLATTENED token index --
    pid = i / (N - 1) * (M - 1), new[i] = (pid - floor) * old[ceil] + (ceil - pid) * old[floor] (so rows that land exactly
    on an old index get weight 0 on both sides, exactly like the reference).  Load-time layout work, done once on the host."""
    M = pos_emb.shape[0]
    if num_new_tokens == M:
        return pos_emb
    mapped = torch.arange(num_new_tokens) / (num_new_tokens - 1) * (M - 1)
    fl = torch.clamp(mapped.floor().long(), min=0, max=M - 1)
    ce = torch.clamp(mapped.ceil().long(), min=0, max=M - 1)
    w = pos_emb.detach().cpu()
    return (mapped - fl)[:, None] * w[ce, :] + (ce - mapped)[:, None] * w[fl, :]


def read_checkpoint(model_path: str, vision_resolution: int = -1, interpolate_mode: str = "linear"):
    """-> (SrgptConfig, state dict with the reference's key names on the host)."""
    cfg = config_from_checkpoint(model_path)
    sd = {}
    _load_dir(os.path.join(model_path, "llm"), "llm.", sd)
    vt = {}
    _load_dir(os.path.join(model_path, "vision_tower"), "", vt)
    # transformers 4.37.2 (the reference's pin) writes the tower's keys under `vision_model.`; transformers 5 flattened the
    # wrapper away (`embeddings.* / encoder.*` at top level) -- the in-memory names here are the 4.37.2 ones
    for k, v in vt.items():
        sd["vision_tower.vision_tower." + (k if k.startswith("vision_model.") else "vision_model." + k)] = v
    del vt
    _load_dir(os.path.join(model_path, "mm_projector"), "mm_projector.", sd)
    if cfg.enable_region:
        _load_dir(os.path.join(model_path, "region_extractor"), "region_extractor.", sd)
    if vision_resolution not in (-1, None, cfg.image_size):
        # vision_resolution elevation (llava/train/utils.py:126-134 -> vision_encoder.py:36-113) applied at load: the only way
        # a SigLIP-384 checkpoint serves 336-px inputs (24 x 24 = 576 tokens)
        if interpolate_mode != "linear":
            raise NotImplementedError(interpolate_mode)  # vision_encoder.py:97-98
        if cfg.tower == "clip":
            raise NotImplementedError("position-embedding resize of a class-token tower is not defined by the reference formula")
        key = "vision_tower.vision_tower.vision_model.embeddings.position_embedding.weight"
        n_new = int((vision_resolution // cfg.patch_size) ** 2)
        print(f"Resizing vision model's position embeddings to support higher vision resolution: from {cfg.image_size} "
              f"to {vision_resolution} ...")
        sd[key] = resize_position_embeddings(sd[key], n_new).to(sd[key].dtype)
        cfg.image_size = int(vision_resolution)
    return cfg, sd


def load_tokenizer(model_path: str, cfg: SrgptConfig, sd=None, model_max_length=None):
    """The tokenizer as `build_llm_and_tokenizer` loads it (language_model/builder.py:84-91: `model_max_length=
    llm_cfg.model_max_length, padding_side="right", use_fast=False, legacy=False`; `llm_cfg.model_max_length` is the loader's
    `model_max_length` argument -- None on the eval / demo path, which OVERRIDES the value stored in tokenizer_config.json),
    then the tokenizer side effects of the loader (llava/model/builder.py:186-199): add <mask>/<depth> (and the optional
    <im_patch>/<im_start>/<im_end>) as special tokens, record their ids, grow the embedding / lm_head tables to
    len(tokenizer) (new rows = mean of the old ones, what `resize_token_embeddings` initialises them to)."""
    from transformers import AutoTokenizer

    try:
        tokenizer = AutoTokenizer.from_pretrained(os.path.join(model_path, "llm"), model_max_length=model_max_length,
                                                  padding_side="right", use_fast=False, legacy=False)
    except Exception as e:  # tokenizer problems must not hide the model; callers that need it fail on use
        import warnings

        warnings.warn(f"could not load tokenizer from {model_path}/llm: {e}")
        return None
    if cfg.enable_region:
        tokenizer.add_tokens([DEFAULT_MASK_TOKEN, DEFAULT_DEPTH_TOKEN], special_tokens=True)
        cfg.mask_token_id = tokenizer.convert_tokens_to_ids(DEFAULT_MASK_TOKEN)
        cfg.depth_token_id = tokenizer.convert_tokens_to_ids(DEFAULT_DEPTH_TOKEN)
    if cfg.mm_use_im_patch_token:
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if cfg.mm_use_im_start_end:
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    if sd is not None:
        n_tok, emb = len(tokenizer), sd["llm.model.embed_tokens.weight"]
        if n_tok > emb.shape[0]:
            extra = n_tok - emb.shape[0]
            sd["llm.model.embed_tokens.weight"] = torch.cat([emb, emb.float().mean(0, keepdim=True).to(emb.dtype).expand(extra, -1)], 0)
            head = sd["llm.lm_head.weight"]
            sd["llm.lm_head.weight"] = torch.cat([head, head.float().mean(0, keepdim=True).to(head.dtype).expand(extra, -1)], 0)
        cfg.vocab = sd["llm.model.embed_tokens.weight"].shape[0]
    # NOTE: the tokenizer's eos / pad ids are NOT copied into the generation defaults: HF generate() never consults the
    # tokenizer (a checkpoint whose configs name no EOS id generates to max_new_tokens in the reference too)
    return tokenizer


def load_image_processor(model_path: str, cfg: SrgptConfig, tower_dir: str = None):
    """SiglipImageProcessor for SigLIP towers (siglip_encoder.py:10), CLIPImageProcessor for CLIP towers (clip_encoder.py:11);
    the built-in processor of the same semantics when transformers cannot build its own (no torchvision in this image).
    tower_dir: the tower directory itself when it is not `<model_path>/vision_tower` (spatialrgpt_amd/factories.py)."""
    vt = tower_dir or os.path.join(model_path, "vision_tower")
    pp = os.path.join(vt, "preprocessor_config.json")
    try:
        if cfg.tower == "clip":
            from transformers import CLIPImageProcessor as Proc
        else:
            from transformers import SiglipImageProcessor as Proc
        proc = Proc.from_pretrained(vt)
    except Exception:
        j = _read_json(pp) if os.path.exists(pp) else {}
        size = j.get("size", {})
        if cfg.tower == "clip":
            crop = j.get("crop_size", {"height": cfg.image_size, "width": cfg.image_size})
            crop = crop if isinstance(crop, dict) else {"height": crop, "width": crop}
            proc = SrgptImageProcessor(size=crop["height"], image_mean=j.get("image_mean", (0.48145466, 0.4578275, 0.40821073)),
                                       image_std=j.get("image_std", (0.26862954, 0.26130258, 0.27577711)),
                                       shortest_edge=(size.get("shortest_edge") if isinstance(size, dict) else size) or crop["height"],
                                       center_crop=True)
        else:
            proc = SrgptImageProcessor(size=size.get("height", cfg.image_size) if isinstance(size, dict) else (size or cfg.image_size),
                                       image_mean=j.get("image_mean", (0.5, 0.5, 0.5)), image_std=j.get("image_std", (0.5, 0.5, 0.5)))
    # a vision_resolution-elevated tower: the processor follows (vision_encoder.py:103-109)
    def _get(d, k):
        try:
            return d[k]
        except (KeyError, TypeError, AttributeError):
            return getattr(d, k, None)

    if getattr(proc, "crop_size", None) is not None:  # CLIP
        if _get(proc.crop_size, "height") != cfg.image_size:
            proc.crop_size = {"height": cfg.image_size, "width": cfg.image_size}
            if _get(getattr(proc, "size", None), "shortest_edge") is not None:
                proc.size = {"shortest_edge": cfg.image_size}
    elif getattr(proc, "size", None) is not None and _get(proc.size, "height") != cfg.image_size:  # SigLIP
        proc.size = {"height": cfg.image_size, "width": cfg.image_size}
    return proc


def load_model(model_path, device="cuda", dtype=torch.bfloat16, llm_weight_format="native", vision_resolution=-1,
               interpolate_mode="linear"):
    """-> (tokenizer, LlavaLlamaModel, image_processor): everything `load_pretrained_model` and the HF-registry entry points
    (`AutoModel.from_pretrained`, `LlavaLlamaModel(config=...)`) share."""
    from .model import LlavaLlamaModel

    cfg, sd = read_checkpoint(model_path, vision_resolution, interpolate_mode)
    tokenizer = load_tokenizer(model_path, cfg, sd)
    image_processor = load_image_processor(model_path, cfg)
    model = LlavaLlamaModel(cfg, sd, device=device, dtype=dtype, tokenizer=tokenizer, image_processor=image_processor,
                            consume_state_dict=True, llm_weight_format=llm_weight_format)
    return tokenizer, model, image_processor


def load_pretrained_model(model_path, model_name, model_base=None, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", dtype=torch.bfloat16, **kwargs):
    if load_4bit:
        raise NotImplementedError("bitsandbytes 4-bit loading is out of scope (builder.py:40-60)")
    # load_8bit (bitsandbytes int8 in the reference, builder.py:51-52) maps to the engine's 8-bit option: weight-only OCP
    # fp8 (e4m3fn, one scale per output row) for the LLM matrices the decode step streams
    llm_weight_format = "fp8" if load_8bit else kwargs.pop("llm_weight_format", "native")
    if model_base is not None or "lora" in model_name.lower():
        raise NotImplementedError("LoRA / delta checkpoints are out of scope (builder.py:64-139)")
    tokenizer, model, image_processor = load_model(model_path, device=device, dtype=dtype, llm_weight_format=llm_weight_format,
                                                   vision_resolution=kwargs.pop("vision_resolution", -1),
                                                   interpolate_mode=kwargs.pop("interpolate_mode", "linear"))
    lc = _read_json(os.path.join(model_path, "llm", "config.json"))
    context_len = _read_json(os.path.join(model_path, "config.json")).get("max_sequence_length", 2048) \
        if "max_sequence_length" in lc else 2048  # builder.py:207-211
    return tokenizer, model, image_processor, context_len
