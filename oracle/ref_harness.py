"""TEST INFRASTRUCTURE ONLY -- not part of the product path.

Imports the *real* SpatialRGPT reference (read-only tree at /root/reference) on CPU so that
`oracle/make_golden.py` can (a) mint golden vectors from it and (b) validate the self-contained
restatement in `oracle/srgpt_oracle.py` against it.  Nothing here is importable on the GPU box
(/root/reference does not exist there); only the committed fixtures under tests/golden/ travel.

The reference tree is pure Python but imports a number of packages that are absent from this image
(deepspeed, flash_attn, s2wrapper, cv2, pycocotools, timm, torchvision, peft, bitsandbytes, open_clip).
They are replaced by inert stub modules *before* `import llava`.  No reference
source is copied: every module is loaded from where it lies.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import json
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SRGPT_REFERENCE_ROOT", "/root/reference")


class _Stub(types.ModuleType):
    """Module whose every attribute is another stub / a dummy class (dunder lookups still fail)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if name[:1].isupper():
            obj = type(name, (), {"__init__": lambda self, *a, **k: None})
        else:
            obj = _Stub(full)
            obj.__spec__ = importlib.machinery.ModuleSpec(full, None)
            obj.__path__ = []
            sys.modules.setdefault(full, obj)
        setattr(self, name, obj)
        return obj

    def __call__(self, *a, **k):
        raise RuntimeError(f"stubbed module {self.__name__} was called")


def _stub(name: str) -> _Stub:
    if name in sys.modules:
        return sys.modules[name]
    m = _Stub(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(_stub(parent), child, m)
    return m


_THIRD_PARTY_STUBS = [
    "deepspeed", "deepspeed.comm", "s2wrapper", "cv2", "pycocotools", "pycocotools.mask",
    "flash_attn", "flash_attn.bert_padding", "flash_attn.flash_attn_interface",
    "timm", "timm.models", "timm.models.layers", "timm.layers", "timm.data",
    "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
    "peft", "bitsandbytes", "open_clip", "einops_exts",
]

# out-of-scope reference modules (other towers / LLMs) that use transformers APIs removed in v5;
# imported unconditionally by llava/model/__init__.py:2-3 and multimodal_encoder/builder.py:8-9.
_REFERENCE_STUBS = [
    "llava.model.multimodal_encoder.intern_encoder",
    "llava.model.multimodal_encoder.radio_encoder",
    "llava.model.language_model.llava_mistral",
    "llava.model.language_model.llava_mixtral",
]

_installed = False


def install_shims() -> None:
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT} (only available in the build container)")
    # transformers first: its availability probes (is_torchvision_available, ...) must see the real
    # environment, not the stubs installed below.
    import transformers  # noqa: F401
    import transformers.modeling_utils as mu
    from transformers import SiglipImageProcessor, SiglipVisionModel, LlamaForCausalLM  # noqa: F401  (resolve lazies)

    for name in _THIRD_PARTY_STUBS:
        try:
            importlib.import_module(name)
        except Exception:
            _stub(name)

    if not hasattr(mu, "no_init_weights"):  # moved in transformers v5 (llava_arch.py:34 needs it)
        from transformers import initialization

        mu.no_init_weights = initialization.no_init_weights
    if not hasattr(mu, "ContextManagers"):
        from transformers.utils import ContextManagers

        mu.ContextManagers = ContextManagers
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in _REFERENCE_STUBS:
        # leaf only: the parent packages (llava, llava.model, ...) must stay the real ones
        m = _Stub(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        # names that the importing modules pull out with `from x import A, B`
        for cls in ("InternVisionTower", "RADIOVisionTower", "LlavaMistralConfig", "LlavaMistralForCausalLM",
                    "LlavaMixtralConfig", "LlavaMixtralForCausalLM"):
            setattr(m, cls, type(cls, (), {}))
    _installed = True


_vendored = None


def install_vendored_llama():
    """Load the reference's VENDORED Llama (llava/train/transformers_replace/models/llama/modeling_llama.py -- the file the
    reference copies over transformers' own, environment_setup.sh:33-35) from where it lies, as a module inside the installed
    `transformers.models.llama` package so that its relative imports resolve.  Its decoder layer hard-wires
    `LlamaFlashAttention2` (:611-619) and the file imports `flash_attn` at module level: `oracle/flash_attn_cpu.py` (a plain-torch
    restatement of flash-attn's published semantics) stands in for that CUDA extension.  Returns the module."""
    global _vendored
    if _vendored is not None:
        return _vendored
    install_shims()
    import importlib.util

    import transformers.models.llama  # noqa: F401

    from oracle import flash_attn_cpu

    fa = types.ModuleType("flash_attn")
    fa.flash_attn_func = flash_attn_cpu.flash_attn_func
    fa.flash_attn_varlen_func = flash_attn_cpu.flash_attn_varlen_func
    fa.__spec__ = importlib.machinery.ModuleSpec("flash_attn", None)
    fa.__path__ = []
    bp = types.ModuleType("flash_attn.bert_padding")
    bp.index_first_axis = flash_attn_cpu.index_first_axis
    bp.pad_input = flash_attn_cpu.pad_input
    bp.unpad_input = flash_attn_cpu.unpad_input
    bp.__spec__ = importlib.machinery.ModuleSpec("flash_attn.bert_padding", None)
    fa.bert_padding = bp
    sys.modules["flash_attn"] = fa
    sys.modules["flash_attn.bert_padding"] = bp
    path = os.path.join(REFERENCE_ROOT, "llava/train/transformers_replace/models/llama/modeling_llama.py")
    spec = importlib.util.spec_from_file_location("transformers.models.llama._srgpt_vendored_modeling_llama", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    _vendored = mod
    return mod


# ------------------------------------------------------------------------------------------------
# tiny on-disk sub-models so that the reference's own builders (from_pretrained based) can run
# ------------------------------------------------------------------------------------------------

def _write_tiny_tokenizer(path: str, vocab_size: int) -> None:
    """A sentencepiece BPE tokenizer trained on a toy corpus (no tokenizer ships offline)."""
    import sentencepiece as spm

    os.makedirs(path, exist_ok=True)
    corpus = os.path.join(path, "_corpus.txt")
    words = ("the quick brown fox jumps over lazy dog region mask depth image left right behind front "
             "wide tall big small distance between and of is how far from to what which object meters").split()
    with open(corpus, "w") as f:
        for i in range(400):
            f.write(" ".join(words[(i * 7 + j * 3) % len(words)] for j in range(9)) + "\n")
    spm.SentencePieceTrainer.train(
        input=corpus, model_prefix=os.path.join(path, "tokenizer"), vocab_size=vocab_size, model_type="bpe",
        bos_id=1, eos_id=2, unk_id=0, pad_id=-1, character_coverage=1.0, hard_vocab_limit=False,
        minloglevel=2,
    )
    os.remove(corpus)
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "LlamaTokenizer", "bos_token": "<s>", "eos_token": "</s>",
                   "unk_token": "<unk>", "add_bos_token": True, "add_eos_token": False,
                   "model_max_length": 4096, "legacy": False}, f)


def build_tiny_reference_model(workdir: str, *, llm: dict, vit: dict, dtype: str = "torch.float32", seed: int = 0,
                               tower: str = "siglip", select_feature: str = "cls_patch"):
    """Build the reference `LlavaLlamaModel` (llava/model/language_model/llava_llama.py:48) from tiny,
    seeded, randomly initialised sub-models written under `workdir`.  Mirrors what
    `load_pretrained_model` (llava/model/builder.py:141-204) does after construction."""
    install_shims()
    import torch
    from transformers import (CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM, SiglipVisionConfig,
                              SiglipVisionModel)

    torch.manual_seed(seed)
    llm_dir = os.path.join(workdir, "llm")
    vit_dir = os.path.join(workdir, f"{tower}_tower")  # the reference dispatches on "clip" / "siglip" in the path
    os.makedirs(workdir, exist_ok=True)

    lcfg = LlamaConfig(**llm)
    lcfg.architectures = ["LlamaForCausalLM"]
    lm = LlamaForCausalLM(lcfg).to(torch.float32)
    lm.save_pretrained(llm_dir)
    _write_tiny_tokenizer(llm_dir, vocab_size=llm["vocab_size"] - 8)

    if tower == "clip":
        vcfg = CLIPVisionConfig(**vit)
        vcfg.architectures = ["CLIPVisionModel"]
        vm = CLIPVisionModel(vcfg).to(torch.float32)
        pp = {"image_processor_type": "CLIPImageProcessor", "do_resize": True, "size": {"shortest_edge": vit["image_size"]},
              "do_center_crop": True, "crop_size": {"height": vit["image_size"], "width": vit["image_size"]},
              "do_rescale": True, "rescale_factor": 1 / 255.0, "do_normalize": True,
              "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711],
              "resample": 3, "do_convert_rgb": True}
    else:
        vcfg = SiglipVisionConfig(**vit)
        vcfg.architectures = ["SiglipVisionModel"]
        vm = SiglipVisionModel(vcfg).to(torch.float32)
        pp = {"image_processor_type": "SiglipImageProcessor", "do_resize": True,
              "size": {"height": vit["image_size"], "width": vit["image_size"]},
              "do_rescale": True, "rescale_factor": 1 / 255.0, "do_normalize": True,
              "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5], "resample": 3}
    vm.save_pretrained(vit_dir)
    with open(os.path.join(vit_dir, "preprocessor_config.json"), "w") as f:
        json.dump(pp, f)

    from llava.model import LlavaLlamaConfig, LlavaLlamaModel

    cfg = LlavaLlamaConfig(
        llm_cfg=llm_dir, vision_tower_cfg=vit_dir, mm_projector_cfg="mlp_downsample",
        region_extractor_cfg="regiongpt", architectures=["LlavaLlamaModel"], enable_region=True, enable_depth=True,
        resume_path=None, hidden_size=None, mm_hidden_size=None, image_aspect_ratio="resize",
        num_video_frames=None, fps=None, mm_vision_select_layer=-2, mm_vision_select_feature=select_feature,
        mm_use_im_start_end=False, mm_use_im_patch_token=False, mm_projector_lr=None, vision_resolution=None,
        interpolate_mode=None, s2=None, s2_scales=None, s2_max_split_size=None,
    )
    # transformers v5 does not run the custom __init__ defaults: set every field explicitly
    for k, v in dict(
        llm_cfg=llm_dir, vision_tower_cfg=vit_dir, mm_projector_cfg="mlp_downsample",
        region_extractor_cfg="regiongpt", enable_region=True, enable_depth=True, resume_path=None,
        hidden_size=None, mm_hidden_size=None, image_aspect_ratio="resize", num_video_frames=None, fps=None,
        mm_vision_select_layer=-2, mm_vision_select_feature=select_feature, mm_use_im_start_end=False,
        mm_use_im_patch_token=False, mm_projector_lr=None, vision_resolution=None, interpolate_mode=None,
        s2=None, s2_scales=None, s2_max_split_size=None, model_dtype=dtype,
    ).items():
        setattr(cfg, k, v)
    cfg._name_or_path = ""

    torch.manual_seed(seed + 1)
    model = LlavaLlamaModel(cfg, attn_implementation="eager")
    model.eval()
    tok = model.tokenizer
    # llava/model/builder.py:186-192
    tok.add_tokens(["<mask>", "<depth>"], special_tokens=True)
    vt_cfg = model.get_vision_tower().config
    vt_cfg.llm_mask_token_id = tok.convert_tokens_to_ids("<mask>")
    vt_cfg.llm_depth_token_id = tok.convert_tokens_to_ids("<depth>")
    # eager (deterministic, fp32-softmax) attention in the tower as well
    model.get_vision_tower().vision_tower.config._attn_implementation = "eager"
    return model, tok
