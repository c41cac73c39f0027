"""HF-registry side of the boundary (reference: llava/model/configuration_llava.py:4-59 and the two registrations at
llava/model/language_model/llava_llama.py:216-217).

`LlavaConfig` / `LlavaLlamaConfig` carry the reference's top-level `config.json` fields so that a caller going through
`AutoConfig.from_pretrained(path)` / `AutoModel.from_pretrained(path)` -- or the loader's own
`AutoConfig.from_pretrained` + `LlavaLlamaModel(config=..., low_cpu_mem_usage=True)` sequence (llava/model/builder.py:142-158)
-- reaches this implementation.  The fields are declared as annotated class attributes: transformers 5 turns them into
dataclass fields, transformers 4 (the reference pins 4.37.2) reads them as class-level defaults behind `**kwargs`.
"""
from __future__ import annotations

from typing import Any, Optional

from transformers import PretrainedConfig


class LlavaConfig(PretrainedConfig):
    model_type = "llava"

    # sub-model configs / paths (dict, PretrainedConfig or str -- llava/model/utils.py:25-55 resolves them)
    llm_cfg: Optional[Any] = None
    vision_tower_cfg: Optional[Any] = None
    mm_projector_cfg: Optional[Any] = None
    region_extractor_cfg: Optional[Any] = None
    resume_path: Optional[str] = None
    # SpatialRGPT switches
    enable_region: Optional[bool] = None
    enable_depth: Optional[bool] = None
    # geometry / preprocessing
    hidden_size: Optional[int] = None
    mm_hidden_size: Optional[int] = None
    image_aspect_ratio: Optional[str] = None
    num_video_frames: Optional[int] = None
    fps: Optional[float] = None
    mm_vision_select_layer: Optional[int] = None
    mm_vision_select_feature: Optional[str] = None
    mm_use_im_start_end: bool = False
    mm_use_im_patch_token: bool = True
    mm_projector_lr: Optional[float] = None
    vision_resolution: Optional[int] = None
    interpolate_mode: Optional[str] = None
    s2: Optional[bool] = None
    s2_scales: Optional[str] = None
    s2_max_split_size: Optional[int] = None
    model_dtype: Optional[str] = None  # set by prepare_config_for_eval (llava/model/builder.py:228-240)

    def checkpoint_root(self) -> Optional[str]:
        """where the four sub-directories live (llava/model/utils.py:28-31: `_name_or_path`, else `resume_path`)."""
        p = getattr(self, "_name_or_path", None)
        if p and len(p) >= 2:
            return p
        return self.resume_path


class LlavaLlamaConfig(LlavaConfig):
    model_type = "llava_llama"


class MultimodalProjectorConfig(PretrainedConfig):
    """multimodal_projector/base_projector.py:56-62 (`<ckpt>/mm_projector/config.json`)"""
    model_type = "v2l_projector"
    mm_projector_type: Optional[str] = None


class RegionExtractorConfig(PretrainedConfig):
    """region_extractor/base_extractor.py:104-110 (`<ckpt>/region_extractor/config.json`)"""
    model_type = "region_extractor"
    region_extractor_type: Optional[str] = None


class MultimodalProjector:
    """`AutoModel.from_pretrained(<ckpt>/mm_projector)` / `MultimodalProjector.from_pretrained(path, config)` (what the reference's
    build_mm_projector calls, multimodal_projector/builder.py:17-21) -> the HIP-backed projector of spatialrgpt_amd.factories."""
    config_class = MultimodalProjectorConfig

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, config=None, torch_dtype=None, **kwargs):
        import torch

        from .factories import build_mm_projector

        dt = torch_dtype if torch_dtype in (torch.bfloat16, torch.float32) else torch.bfloat16
        return build_mm_projector(str(pretrained_model_name_or_path), config, dtype=dt)


class RegionExtractor:
    """the same for `<ckpt>/region_extractor` (region_extractor/builder.py:18-22)"""
    config_class = RegionExtractorConfig

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, config=None, torch_dtype=None, **kwargs):
        import torch

        from .factories import build_region_extractor

        dt = torch_dtype if torch_dtype in (torch.bfloat16, torch.float32) else torch.bfloat16
        return build_region_extractor(str(pretrained_model_name_or_path), config, dtype=dt)


def register_auto_classes(model_cls) -> None:
    """AutoConfig.register("llava_llama", LlavaLlamaConfig); AutoModel.register(LlavaLlamaConfig, LlavaLlamaModel)
    (llava_llama.py:216-217).  If another package in the process (e.g. the reference itself) already owns the
    "llava_llama" slot, it keeps it."""
    from transformers import AutoConfig, AutoModel

    try:
        AutoConfig.register("llava_llama", LlavaLlamaConfig)
    except ValueError:
        return
    try:
        AutoModel.register(LlavaLlamaConfig, model_cls)
    except ValueError:
        pass
    # base_projector.py:97-98, base_extractor.py:176-177: the component directories load through the Auto classes too
    for name, cfg_cls, mod_cls in (("v2l_projector", MultimodalProjectorConfig, MultimodalProjector),
                                   ("region_extractor", RegionExtractorConfig, RegionExtractor)):
        try:
            AutoConfig.register(name, cfg_cls)
            AutoModel.register(cfg_cls, mod_cls)
        except ValueError:
            pass
