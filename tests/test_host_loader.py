"""CPU: the loader's host-side effects (llava/model/builder.py:186-199, multimodal_encoder/*_encoder.py processors, the
HF registry of llava_llama.py:216-217, vision_resolution position-embedding resize) on a synthetic checkpoint written in the
reference's on-disk layout (llava_arch.py:181-250).  No engine is built here (that needs the GPU: tests/test_gpu_loader.py)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import GOLD, load_tiny


def write_tiny_tokenizer(path, vocab_size):
    """sentencepiece BPE tokenizer trained on a toy corpus + LlamaTokenizer config (no tokenizer ships offline)."""
    import sentencepiece as spm

    os.makedirs(path, exist_ok=True)
    corpus = os.path.join(path, "_corpus.txt")
    words = ("the quick brown fox jumps over lazy dog region mask depth image left right behind front wide tall big small "
             "distance between and of is how far from to what which object meters").split()
    with open(corpus, "w") as f:
        for i in range(400):
            f.write(" ".join(words[(i * 7 + j * 3) % len(words)] for j in range(9)) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(path, "tokenizer"), vocab_size=vocab_size, model_type="bpe",
                                   bos_id=1, eos_id=2, unk_id=0, pad_id=-1, character_coverage=1.0, hard_vocab_limit=False, minloglevel=2)
    os.remove(corpus)
    json.dump({"tokenizer_class": "LlamaTokenizer", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>", "add_bos_token": True,
               "add_eos_token": False, "model_max_length": 4096, "legacy": False}, open(os.path.join(path, "tokenizer_config.json"), "w"))


def write_checkpoint(root, cfgd, w, tokenizer_vocab=None):
    from safetensors.torch import save_file

    os.makedirs(root, exist_ok=True)
    groups = {"llm": {}, "vision_tower": {}, "mm_projector": {}, "region_extractor": {}}
    for k, v in w.items():
        if k.startswith("llm."):
            groups["llm"][k[len("llm."):]] = v.contiguous()
        elif k.startswith("vision_tower.vision_tower."):
            groups["vision_tower"][k[len("vision_tower.vision_tower."):]] = v.contiguous()
        elif k.startswith("mm_projector."):
            groups["mm_projector"][k[len("mm_projector."):]] = v.contiguous()
        elif k.startswith("region_extractor."):
            groups["region_extractor"][k[len("region_extractor."):]] = v.contiguous()
    for name, sd in groups.items():
        os.makedirs(os.path.join(root, name), exist_ok=True)
        save_file(sd, os.path.join(root, name, "model.safetensors"))
    json.dump({"architectures": ["LlavaLlamaModel"], "model_type": "llava_llama", "enable_region": True, "enable_depth": True,
               "mm_vision_select_layer": -2, "mm_vision_select_feature": "cls_patch", "image_aspect_ratio": "resize",
               "mm_use_im_start_end": False, "mm_use_im_patch_token": False,
               "llm_cfg": {"architectures": ["LlamaForCausalLM"]}, "vision_tower_cfg": {"architectures": ["SiglipVisionModel"]},
               "mm_projector_cfg": {"mm_projector_type": "mlp_downsample"}, "region_extractor_cfg": {"region_extractor_type": "regiongpt"}},
              open(os.path.join(root, "config.json"), "w"))
    json.dump({"architectures": ["LlamaForCausalLM"], "model_type": "llama", "hidden_size": cfgd["hidden"], "intermediate_size": cfgd["inter"],
               "num_hidden_layers": cfgd["layers"], "num_attention_heads": cfgd["heads"], "num_key_value_heads": cfgd["kv_heads"],
               "vocab_size": cfgd["vocab"], "rms_norm_eps": cfgd["rms_eps"], "rope_theta": cfgd["rope_theta"],
               "max_position_embeddings": 2048, "eos_token_id": 2, "bos_token_id": 1},
              open(os.path.join(root, "llm", "config.json"), "w"))
    json.dump({"architectures": ["SiglipVisionModel"], "model_type": "siglip_vision_model", "hidden_size": cfgd["vit_hidden"],
               "intermediate_size": cfgd["vit_inter"], "num_hidden_layers": cfgd["vit_layers"],
               "num_attention_heads": cfgd["vit_heads"], "image_size": cfgd["image_size"], "patch_size": cfgd["patch_size"],
               "layer_norm_eps": cfgd["vit_eps"]},
              open(os.path.join(root, "vision_tower", "config.json"), "w"))
    json.dump({"image_processor_type": "SiglipImageProcessor", "size": {"height": cfgd["image_size"], "width": cfgd["image_size"]},
               "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5], "rescale_factor": 1 / 255.0, "do_normalize": True,
               "do_resize": True, "do_rescale": True, "resample": 3},
              open(os.path.join(root, "vision_tower", "preprocessor_config.json"), "w"))
    if tokenizer_vocab:
        write_tiny_tokenizer(os.path.join(root, "llm"), tokenizer_vocab)




def _ckpt(tmp_path, name="tiny_fp32.npz", base=100):
    cfgd, dtype, w, inp, ref = load_tiny(name)
    w = dict(w)
    for k in ("llm.model.embed_tokens.weight", "llm.lm_head.weight"):
        w[k] = w[k][:base].clone()
    cfgd = dict(cfgd, vocab=base)
    root = str(tmp_path / "SpatialRGPT-tiny")
    write_checkpoint(root, cfgd, w, tokenizer_vocab=base)
    return root, cfgd, w


def test_tokenizer_side_effects_of_the_loader(tmp_path):
    """builder.py:186-199 without a GPU: special tokens added in the reference's order, ids recorded, tables grown by the
    mean row, eos/pad taken from the tokenizer."""
    from spatialrgpt_amd import tokenizer_image_token
    from spatialrgpt_amd.builder import load_tokenizer, read_checkpoint

    root, cfgd, w = _ckpt(tmp_path)
    cfg, sd = read_checkpoint(root)
    assert cfg.vocab == 100 and cfg.enable_region and cfg.enable_depth and cfg.tower == "siglip"
    tok = load_tokenizer(root, cfg, sd)
    assert tok is not None and len(tok) == 102
    assert (cfg.mask_token_id, cfg.depth_token_id) == (100, 101) == (tok.convert_tokens_to_ids("<mask>"), tok.convert_tokens_to_ids("<depth>"))
    assert cfg.vocab == 102 and sd["llm.model.embed_tokens.weight"].shape[0] == 102 and sd["llm.lm_head.weight"].shape[0] == 102
    old = w["llm.model.embed_tokens.weight"].float()
    assert torch.allclose(sd["llm.model.embed_tokens.weight"][100].float(), old.mean(0), atol=1e-6)
    assert torch.equal(sd["llm.model.embed_tokens.weight"][:100], w["llm.model.embed_tokens.weight"])
    assert cfg.eos_token_id == 2
    ids = tokenizer_image_token("<image>\nhow far is region <mask> <depth> ?", tok, return_tensors="pt")
    assert ids[0] == tok.bos_token_id and int((ids == -200).sum()) == 1 and 100 in ids.tolist() and 101 in ids.tolist()
    # mm_use_im_patch_token / mm_use_im_start_end add their tokens AFTER <mask>/<depth> (builder.py:193-198)
    cfg2, sd2 = read_checkpoint(root)
    cfg2.mm_use_im_patch_token, cfg2.mm_use_im_start_end = True, True
    tok2 = load_tokenizer(root, cfg2, sd2)
    assert len(tok2) == 105 and tok2.convert_tokens_to_ids("<im_patch>") == 102 and cfg2.vocab == 105


def test_hf_registry_resolves_llava_llama(tmp_path):
    """llava_llama.py:216-217 on the host: AutoConfig maps "llava_llama" to our config class, the reference's top-level fields
    survive the round trip, and AutoModel is wired to LlavaLlamaModel (whose construction needs the GPU)."""
    from transformers import AutoConfig
    from transformers.models.auto.modeling_auto import MODEL_MAPPING

    import spatialrgpt_amd
    from spatialrgpt_amd.configuration import LlavaConfig, LlavaLlamaConfig

    root, cfgd, w = _ckpt(tmp_path)
    model_cls = spatialrgpt_amd.LlavaLlamaModel  # importing the model module performs the registration
    config = AutoConfig.from_pretrained(root)
    assert isinstance(config, LlavaLlamaConfig) and isinstance(config, LlavaConfig)
    assert config.model_type == "llava_llama" and config.enable_region is True and config.enable_depth is True
    assert config.mm_vision_select_layer == -2 and config.mm_vision_select_feature == "cls_patch"
    assert config.mm_projector_cfg == {"mm_projector_type": "mlp_downsample"} and config.resume_path is None
    assert config.checkpoint_root() == root  # _name_or_path, as llava/model/utils.py:28-31 resolves it
    assert MODEL_MAPPING[LlavaLlamaConfig] is model_cls and model_cls.config_class is LlavaLlamaConfig
    c2 = LlavaLlamaConfig(enable_region=True, resume_path="/x", vision_resolution=336)
    assert c2.checkpoint_root() == "/x" and c2.vision_resolution == 336 and c2.mm_use_im_patch_token is True
    d = c2.to_dict()
    assert d["model_type"] == "llava_llama" and d["resume_path"] == "/x"


def test_clip_checkpoint_gets_clip_image_processor(tmp_path):
    """clip_encoder.py:8-13: CLIP towers come with a CLIPImageProcessor (shortest-edge resize + centre crop, `crop_size`)."""
    from spatialrgpt_amd.builder import config_from_checkpoint, load_image_processor

    cfgd, dtype, w, inp, ref = load_tiny("tiny_clip_fp32.npz")
    root = str(tmp_path / "clip")
    write_checkpoint(root, cfgd, w)
    json.dump({"architectures": ["CLIPVisionModel"], "model_type": "clip_vision_model", "hidden_size": cfgd["vit_hidden"],
               "intermediate_size": cfgd["vit_inter"], "num_hidden_layers": cfgd["vit_layers"], "num_attention_heads": cfgd["vit_heads"],
               "image_size": cfgd["image_size"], "patch_size": cfgd["patch_size"], "layer_norm_eps": cfgd["vit_eps"]},
              open(os.path.join(root, "vision_tower", "config.json"), "w"))
    S = cfgd["image_size"]
    json.dump({"image_processor_type": "CLIPImageProcessor", "do_resize": True, "size": {"shortest_edge": S}, "do_center_crop": True,
               "crop_size": {"height": S, "width": S}, "do_rescale": True, "rescale_factor": 1 / 255.0, "do_normalize": True,
               "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711], "resample": 3,
               "do_convert_rgb": True}, open(os.path.join(root, "vision_tower", "preprocessor_config.json"), "w"))
    cfg = config_from_checkpoint(root)
    assert cfg.tower == "clip"
    proc = load_image_processor(root, cfg)
    assert "CLIP" in type(proc).__name__ or getattr(proc, "center_crop", False)
    assert proc.crop_size["height"] == S and abs(proc.image_mean[0] - 0.48145466) < 1e-6


def test_vision_resolution_position_embedding_resize_matches_reference_kat(tmp_path):
    """`VisionTower._maybe_resize_pos_embeds` (vision_encoder.py:36-113, "linear"): tests/golden/posembed_kat.npz was minted by
    running the reference's own method (oracle/make_golden.py) -- bit-identical here; and the loader applies it: a 27x27-token
    SigLIP-style checkpoint read with vision_resolution=336 has 24x24 = 576 rows and a 336-px processor."""
    from spatialrgpt_amd.builder import load_image_processor, read_checkpoint, resize_position_embeddings

    z = np.load(os.path.join(GOLD, "posembed_kat.npz"))
    for tag in ("up", "down"):
        old, new = torch.from_numpy(z[f"{tag}_old"]), torch.from_numpy(z[f"{tag}_new"])
        got = resize_position_embeddings(old, new.shape[0])
        assert got.shape == new.shape and torch.equal(got, new), f"{tag}: max diff {float((got - new).abs().max())}"
    root, cfgd, w = _ckpt(tmp_path)
    S, p = cfgd["image_size"], cfgd["patch_size"]
    res = (S // p - 3) * p
    cfg, sd = read_checkpoint(root, vision_resolution=res)
    key = "vision_tower.vision_tower.vision_model.embeddings.position_embedding.weight"
    assert cfg.image_size == res and sd[key].shape[0] == (res // p) ** 2 == cfg.tower_tokens
    assert torch.equal(sd[key].float(), resize_position_embeddings(w[key].float(), (res // p) ** 2).to(w[key].dtype).float())
    proc = load_image_processor(root, cfg)
    assert proc.size["height"] == res and proc.size["width"] == res
    with pytest.raises(NotImplementedError):
        read_checkpoint(root, vision_resolution=res, interpolate_mode="bicubic")


CKPT_TINY = os.path.join(GOLD, "ckpt_tiny")


def test_loader_reads_the_checkpoint_the_reference_writer_wrote():
    """tests/golden/ckpt_tiny/ was written by the REFERENCE's `LlavaLlamaModel.save_pretrained` (llava_arch.py:181-250) run in the
    build container (oracle/make_golden.py `ckpt`; transformers 5.15 underneath, so the tower's keys are the flattened ones and
    the LLM config carries `rope_parameters`).  The loader must recover the geometry from the nested top-level config + the
    sub-model configs, the exact weights (== tests/golden/tiny_fp32.npz, minted from the same reference model), the generation
    config the reference saved (an EOS list), and a tokenizer whose `tokenizer_image_token` ids equal the reference's."""
    from spatialrgpt_amd import tokenizer_image_token
    from spatialrgpt_amd.builder import config_from_checkpoint, load_image_processor, load_tokenizer, read_checkpoint

    top = json.load(open(os.path.join(CKPT_TINY, "config.json")))
    assert top["architectures"] == ["LlavaLlamaModel"] and isinstance(top["llm_cfg"], dict) and isinstance(top["vision_tower_cfg"], dict)
    cfg, sd = read_checkpoint(CKPT_TINY)
    cfgd, dtype, w, inp, ref = load_tiny("tiny_fp32.npz")
    for k in ("vit_hidden", "vit_inter", "vit_layers", "vit_heads", "image_size", "patch_size", "hidden", "inter", "layers", "heads",
              "kv_heads", "vocab", "rope_theta", "rms_eps", "select_layer", "select_feature", "enable_region", "enable_depth"):
        assert getattr(cfg, k) == cfgd[k], k
    assert cfg.tower == "siglip" and cfg.eos_token_id == [2, 9] and cfg.generation_config["eos_token_id"] == [2, 9]
    assert all(k in sd for k in w), [k for k in w if k not in sd][:4]
    assert all(torch.equal(sd[k], w[k]) for k in w)
    tok = load_tokenizer(CKPT_TINY, cfg, sd)
    z = np.load(os.path.join(GOLD, "ckpt_tiny_kat.npz"))
    assert (cfg.mask_token_id, cfg.depth_token_id) == (int(z["mask_token_id"]), int(z["depth_token_id"])) == (120, 121)
    assert cfg.vocab == 128 and len(tok) == 122  # the saved tokenizer already holds <mask>/<depth>; the table needs no growth
    got = tokenizer_image_token(bytes(z["prompt"]).decode(), tok, return_tensors="pt")
    assert got.tolist() == z["prompt_ids"].tolist()
    proc = load_image_processor(CKPT_TINY, cfg)
    assert proc.size["height"] == cfg.image_size
    # a checkpoint whose sub-directories hold weights only: the nested copies in the top-level config are enough
    import shutil
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        dst = os.path.join(td, "c")
        shutil.copytree(CKPT_TINY, dst)
        os.remove(os.path.join(dst, "llm", "config.json"))
        os.remove(os.path.join(dst, "vision_tower", "config.json"))
        c2 = config_from_checkpoint(dst)
        assert (c2.hidden, c2.vit_hidden, c2.rope_theta, c2.vocab) == (cfg.hidden, cfg.vit_hidden, cfg.rope_theta, 128)
    # the 4.37.2 spelling of the same fields (what a checkpoint published by the reference's authors carries)
    from spatialrgpt_amd.builder import _rope

    assert _rope({"rope_theta": 10000.0, "rope_scaling": {"type": "linear", "factor": 4.0}}) == (10000.0, 4.0)
    assert _rope({"rope_parameters": {"rope_theta": 500000.0, "rope_type": "default"}}) == (500000.0, 1.0)
    assert _rope({}) == (10000.0, 1.0)


def test_component_configs_load_through_the_auto_classes():
    """base_projector.py:97-98 / base_extractor.py:176-177: `AutoConfig.register("v2l_projector", ...)`,
    `AutoConfig.register("region_extractor", ...)` -- the component directories the reference's save_pretrained wrote resolve to this
    package's config classes (their AutoModel entries build the HIP-backed components: tests/test_gpu_loader.py)."""
    import os

    from transformers import AutoConfig

    import spatialrgpt_amd.model  # noqa: F401  (registration happens at import, like the reference)
    from spatialrgpt_amd.configuration import MultimodalProjectorConfig, RegionExtractorConfig
    from tests.util import GOLD

    pc = AutoConfig.from_pretrained(os.path.join(GOLD, "ckpt_tiny", "mm_projector"))
    rc = AutoConfig.from_pretrained(os.path.join(GOLD, "ckpt_tiny", "region_extractor"))
    assert isinstance(pc, MultimodalProjectorConfig) and pc.mm_projector_type == "mlp_downsample"
    assert isinstance(rc, RegionExtractorConfig) and rc.region_extractor_type == "regiongpt"
