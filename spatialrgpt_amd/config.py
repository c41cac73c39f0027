"""Model geometry of the region-grounded path (what the reference spreads over LlavaConfig,
SiglipVisionConfig and LlamaConfig: llava/model/configuration_llava.py:4-59 + the HF sub-configs)."""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import List, Optional, Union


@dataclass
class SrgptConfig:
    # vision tower (SigLIP)
    vit_hidden: int = 1152
    vit_inter: int = 4304
    vit_layers: int = 27
    vit_heads: int = 16
    image_size: int = 384
    patch_size: int = 14
    vit_eps: float = 1e-6
    select_layer: int = -2            # mm_vision_select_layer (scripts/srgpt/llama3_8b/3_sft.sh)
    select_feature: str = "cls_patch"  # mm_vision_select_feature
    tower: str = "siglip"              # "siglip" (SiglipVisionTower) or "clip" (CLIPVisionTower)
    # language model (Llama)
    hidden: int = 4096
    inter: int = 14336
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    vocab: int = 128258
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 1.0
    max_position_embeddings: int = 8192
    # token stream
    mask_token_id: int = 128256
    depth_token_id: int = 128257
    enable_region: bool = True
    enable_depth: bool = True
    tokenizer_model_max_length: Optional[int] = None
    padding_side: str = "right"
    eos_token_id: Optional[Union[int, List[int]]] = None  # an int or a LIST (Llama-3 generation_config.json: [128001, 128009])
    pad_token_id: Optional[int] = None
    generation_config: Optional[dict] = None  # <ckpt>/llm/generation_config.json (or the generation fields of llm/config.json)
    image_aspect_ratio: str = "resize"
    mm_use_im_start_end: bool = False
    mm_use_im_patch_token: bool = False
    mm_projector_type: str = "mlp_downsample"
    region_extractor_type: str = "regiongpt"

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tower_tokens(self) -> int:
        """tokens per image inside the tower (CLIP prepends a class token)"""
        return self.grid ** 2 + (1 if self.tower == "clip" else 0)

    @property
    def llm_image_tokens(self) -> int:
        """rows one image contributes to the LLM sequence: the mlp_downsample projector folds 2x2 blocks of the (27 x 27 pooled, or
        raw tower) grid, odd sides zero-padded (base_projector.py:32-52) -- 196 for every recipe of the reference."""
        g = 27 if self.enable_region else int((self.tower_tokens - (1 if self.select_feature == "patch" else 0)) ** 0.5)
        return ((g + 1) // 2) ** 2

    @property
    def vit_layers_run(self) -> int:
        """hidden_states[select_layer] = output of this many encoder layers."""
        return self.vit_layers + 1 + self.select_layer if self.select_layer < 0 else self.select_layer

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_dict(cls, d):
        names = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in names})

    # named geometries of the reference's recipes (scripts/srgpt/{llama3_8b,llama2_7b,sheared_3b})
    @classmethod
    def vila15_8b(cls):
        return cls()

    @classmethod
    def llama2_7b(cls):
        return cls(hidden=4096, inter=11008, layers=32, heads=32, kv_heads=32, vocab=32002, rope_theta=10000.0,
                   mask_token_id=32000, depth_token_id=32001, max_position_embeddings=4096)

    @classmethod
    def clip_l14_336(cls, **llm):
        """CLIP-L/14-336 tower (576 patch tokens with select_feature="patch") in front of any LLM."""
        return cls(vit_hidden=1024, vit_inter=4096, vit_layers=24, vit_heads=16, image_size=336, patch_size=14, vit_eps=1e-5,
                   tower="clip", select_feature="patch", **llm)

    @classmethod
    def sheared_3b(cls):
        return cls(hidden=2560, inter=6912, layers=32, heads=20, kv_heads=20, vocab=32002, rope_theta=10000.0,
                   mask_token_id=32000, depth_token_id=32001, max_position_embeddings=4096)
