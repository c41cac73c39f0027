"""Module-level factories: the reference's component builders on the MI355X kernels (SURVEY 8b "factories").

The reference assembles its model from four builders and a maintainer can swap ONE component at a time:

    build_vision_tower(model_name_or_path, config)          llava/model/multimodal_encoder/builder.py:13-48
        -> module(images [N,3,S,S]) -> [N, L, C]; `.image_processor`, `.config`, `.is_loaded`, `.hidden_size`
    build_region_extractor(model_type_or_path, config)      llava/model/region_extractor/builder.py:12-24
        -> `.feature_refinement(tower_features) -> (hres, lres)`; module(hres, depth_features | None, masks) -> (list, list | None)
    build_mm_projector(model_type_or_path, config)          llava/model/multimodal_projector/builder.py:11-23
        -> module(x [N, L, C]) -> [N, 196, H]
    build_llm_and_tokenizer(model_name_or_path, config, attn_implementation=None, model_max_length=None)
                                                            llava/model/language_model/builder.py:41-98
        -> (llm, tokenizer); sets `config.hidden_size`; `llm.generate(inputs_embeds=..., attention_mask=..., **kw)`

Each factory here returns an object with that surface whose arithmetic is the HIP path (a `SrgptEngine` holding ONLY that
component's weights).  `model_*_path` is the component directory the reference's `save_pretrained` writes
(`<ckpt>/vision_tower`, `<ckpt>/region_extractor`, `<ckpt>/mm_projector`, `<ckpt>/llm`; llava_arch.py:181-250); instead of a path
a `state_dict=` with the reference module's own key names may be passed (e.g. `model.get_region_extractor().state_dict()`), which
is how a component of a live reference model is replaced.  The type strings of a fresh training run ("mlp_downsample",
"regiongpt": random initialisation) need a `state_dict`: this is the inference path.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .builder import _load_dir, _read_json, _rope, load_image_processor
from .config import SrgptConfig


def _cfg_get(config, name, default=None):
    if config is None:
        return default
    if isinstance(config, dict):
        return config.get(name, default)
    return getattr(config, name, default)


def _component_state(path_or_type, state_dict, prefix: str) -> Dict[str, torch.Tensor]:
    if state_dict is not None:
        return {prefix + k: v for k, v in state_dict.items()}
    if not (isinstance(path_or_type, str) and os.path.isdir(path_or_type)):
        raise ValueError(f"{path_or_type!r} is not a component directory; pass state_dict= (random initialisation of a fresh "
                         "component is the training path, out of scope)")
    out: Dict[str, torch.Tensor] = {}
    _load_dir(path_or_type, prefix, out)
    return out


def _engine(cfg: SrgptConfig, sd, parts, device, dtype, **kw):
    from .engine import SrgptEngine

    return SrgptEngine(cfg, sd, device=device, dtype=dtype, consume_state_dict=True, parts=parts, **kw)


def build_vision_tower(model_name_or_path: str, config=None, state_dict=None, device="cuda", dtype=torch.bfloat16):
    """SigLIP / CLIP tower directory (HF config.json + weights + preprocessor_config.json) -> VisionTower-like module."""
    from .model import _VisionTower

    if model_name_or_path is None:
        return None  # multimodal_encoder/builder.py:16-17
    vc = _read_json(os.path.join(model_name_or_path, "config.json")) if os.path.isdir(str(model_name_or_path)) else {}
    vc = vc.get("vision_config", vc)
    name = (str(model_name_or_path) + " " + str(vc.get("model_type", ""))).lower()
    if "siglip" in name:
        tower = "siglip"
    elif "clip" in name:
        tower = "clip"
    else:
        raise ValueError(f"Unknown vision tower: {model_name_or_path}")  # multimodal_encoder/builder.py:46
    raw = _component_state(model_name_or_path, state_dict, "")
    sd = {"vision_tower.vision_tower." + (k if k.startswith("vision_model.") else "vision_model." + k): v for k, v in raw.items()}
    cfg = SrgptConfig(vit_hidden=vc["hidden_size"], vit_inter=vc["intermediate_size"], vit_layers=vc["num_hidden_layers"],
                      vit_heads=vc["num_attention_heads"], image_size=vc["image_size"], patch_size=vc["patch_size"],
                      vit_eps=vc.get("layer_norm_eps", 1e-6 if tower == "siglip" else 1e-5),
                      select_layer=_cfg_get(config, "mm_vision_select_layer", -2) or -2,
                      select_feature=_cfg_get(config, "mm_vision_select_feature", "cls_patch") or "cls_patch", tower=tower)
    eng = _engine(cfg, sd, ("vit",), device, dtype)
    proc = load_image_processor(os.path.dirname(os.path.abspath(model_name_or_path)), cfg, tower_dir=model_name_or_path) \
        if os.path.isdir(str(model_name_or_path)) else None
    vt = _VisionTower(eng, proc)
    vt.hidden_size = cfg.vit_hidden
    vt.num_patches = cfg.tower_tokens
    if config is not None and not isinstance(config, dict):
        config.mm_hidden_size = cfg.vit_hidden  # multimodal_encoder/builder.py:47
    return vt


def build_region_extractor(model_type_or_path: str, config=None, state_dict=None, device="cuda", dtype=torch.bfloat16):
    """`<ckpt>/region_extractor` (or the state dict of the reference's RegionExtractor) -> RegionExtractor-like module."""
    from .model import _RegionExtractor

    if model_type_or_path is None:
        return None  # region_extractor/builder.py:15-16
    sd = _component_state(model_type_or_path, state_dict, "region_extractor.")
    C_ = sd["region_extractor.feature_refinement_module.0.weight"].shape[0]
    H = sd["region_extractor.rgb_projector.weight"].shape[0]
    cfg = SrgptConfig(vit_hidden=C_, hidden=H, enable_region=True, enable_depth=True)
    return _RegionExtractor(_engine(cfg, sd, ("region",), device, dtype))


def build_mm_projector(model_type_or_path: str, config=None, state_dict=None, device="cuda", dtype=torch.bfloat16):
    """`<ckpt>/mm_projector` (mlp_downsample: layers.{1,2,4}.*) -> projector module(x) -> [N, ceil(g/2)^2, H]."""
    from .model import _Projector

    if model_type_or_path is None:
        return None  # multimodal_projector/builder.py:14-15
    if isinstance(model_type_or_path, str) and os.path.isdir(model_type_or_path):
        pc = os.path.join(model_type_or_path, "config.json")
        kind = _read_json(pc).get("mm_projector_type", "mlp_downsample") if os.path.exists(pc) else "mlp_downsample"
    else:
        kind = model_type_or_path
    if kind != "mlp_downsample":
        raise ValueError(f"Unknown projector type: {kind}")  # base_projector.py:90-91 (the SpatialRGPT recipes use mlp_downsample)
    sd = _component_state(model_type_or_path, state_dict, "mm_projector.")
    H, C4 = sd["mm_projector.layers.2.weight"].shape
    cfg = SrgptConfig(vit_hidden=C4 // 4, hidden=H)
    return _Projector(_engine(cfg, sd, ("projector",), device, dtype))


def build_llm_and_tokenizer(model_name_or_path: str, config=None, attn_implementation=None, model_max_length=None, state_dict=None,
                            device="cuda", dtype=torch.bfloat16, llm_weight_format: str = "native"):
    """`<ckpt>/llm` (HF LlamaForCausalLM directory + tokenizer files) -> (llm, tokenizer).  `llm.generate(inputs_embeds=...,
    attention_mask=..., **generation_kwargs)` is the call of llava_llama.py:212; `llm.get_input_embeddings()` the embedding lookup.
    `attn_implementation` is accepted and ignored (there is one attention implementation here)."""
    from .generation import generation_config_from_files
    from .model import LlavaLlamaModel

    lc = _read_json(os.path.join(model_name_or_path, "config.json"))
    if model_max_length:
        orig = lc.get("max_position_embeddings")
        lc["model_max_length"] = model_max_length
        if orig and model_max_length > orig:  # context_length_extension, language_model/builder.py:31-38
            import math

            print(f"Scaling RoPE from {orig} to {model_max_length}")
            lc["rope_scaling"] = {"type": "linear", "factor": float(math.ceil(model_max_length / orig))}
    theta, factor = _rope(lc)
    gp = os.path.join(model_name_or_path, "generation_config.json")
    gen = generation_config_from_files(lc, _read_json(gp) if os.path.exists(gp) else None)
    cfg = SrgptConfig(hidden=lc["hidden_size"], inter=lc["intermediate_size"], layers=lc["num_hidden_layers"],
                      heads=lc["num_attention_heads"], kv_heads=lc.get("num_key_value_heads", lc["num_attention_heads"]),
                      vocab=lc["vocab_size"], rms_eps=lc.get("rms_norm_eps", 1e-5), rope_theta=theta, rope_factor=factor,
                      max_position_embeddings=int(lc.get("model_max_length") or lc.get("max_position_embeddings", 4096)),
                      eos_token_id=gen.get("eos_token_id"), pad_token_id=gen.get("pad_token_id"), generation_config=gen,
                      enable_region=False, enable_depth=False)
    sd = _component_state(model_name_or_path, state_dict, "llm.")
    cfg.vocab = sd["llm.model.embed_tokens.weight"].shape[0]
    model = LlavaLlamaModel(cfg, sd, device=device, dtype=dtype, consume_state_dict=True, llm_weight_format=llm_weight_format,
                            parts=("llm",))
    tokenizer = None
    if any(os.path.exists(os.path.join(model_name_or_path, f)) for f in ("tokenizer.model", "tokenizer.json", "tokenizer_config.json")):
        from transformers import AutoTokenizer

        kw = dict(padding_side="right", use_fast=False, legacy=False)  # language_model/builder.py:84-91
        if model_max_length:
            kw["model_max_length"] = model_max_length
        try:
            tokenizer = AutoTokenizer.from_pretrained(model_name_or_path, **kw)
        except Exception as e:  # tokenizer problems must not hide the model (as in builder.load_tokenizer)
            import warnings

            warnings.warn(f"could not load tokenizer from {model_name_or_path}: {e}")
    if config is not None and not isinstance(config, dict):
        config.hidden_size = cfg.hidden  # language_model/builder.py:97
    model.tokenizer = tokenizer
    return model.llm, tokenizer
