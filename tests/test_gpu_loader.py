"""GPU: `load_pretrained_model` on a synthetic checkpoint written in the reference's on-disk layout
(llava_arch.py:181-250): llm/ vision_tower/ mm_projector/ region_extractor/ + top config.json."""
import json
import os

import pytest
import torch

from tests.util import load_tiny

pytestmark = pytest.mark.gpu


def _write_checkpoint(root, cfgd, w):
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


def test_load_pretrained_model_roundtrip(tmp_path):
    from spatialrgpt_amd import load_pretrained_model

    cfgd, dtype, w, inp, ref = load_tiny("tiny_bf16.npz")
    root = str(tmp_path / "SpatialRGPT-tiny")
    _write_checkpoint(root, cfgd, w)
    tokenizer, model, image_processor, context_len = load_pretrained_model(root, "SpatialRGPT-tiny")
    assert context_len == 2048 and image_processor is not None
    assert model.config.enable_region and model.config.enable_depth and model.config.vit_layers_run == cfgd["vit_layers"] - 1
    # no tokenizer files in this synthetic checkpoint: ids come from the fixture (the token ids of <mask>/<depth> too)
    model.config.mask_token_id, model.config.depth_token_id = cfgd["mask_token_id"], cfgd["depth_token_id"]
    model.to(dtype=torch.bfloat16)  # the reference's callers do this; must be a no-op
    out = model.generate(inp["input_ids"].cuda(), images=inp["images"].cuda(), depths=inp["depths"].cuda(),
                         masks=[m.cuda() for m in inp["masks"]], do_sample=False, max_new_tokens=4, eos_token_id=None)
    assert out.shape == (1, 4)
    st = {}
    model.engine.prepare_inputs(inp["input_ids"].cuda(), inp["images"].cuda(), inp["depths"].cuda(), [m.cuda() for m in inp["masks"]],
                                None, stages=st)
    err = (st["inputs_embeds"].float().cpu() - ref["inputs_embeds"].float()).abs().max().item()
    assert err <= 3e-2 * ref["inputs_embeds"].float().abs().max().item()
    with pytest.raises(NotImplementedError):
        load_pretrained_model(root, "x", load_4bit=True)
    # load_8bit -> weight-only fp8 for the streamed LLM matrices (the reference's bitsandbytes int8 slot, builder.py:51-52)
    _, m8, _, _ = load_pretrained_model(root, "SpatialRGPT-tiny", load_8bit=True)
    assert m8.engine.w.llm_weight_format == "fp8" and m8.engine.w.llm_q is not None
    m8.config.mask_token_id, m8.config.depth_token_id = cfgd["mask_token_id"], cfgd["depth_token_id"]
    out8 = m8.generate(inp["input_ids"].cuda(), images=inp["images"].cuda(), depths=inp["depths"].cuda(),
                       masks=[m.cuda() for m in inp["masks"]], do_sample=False, max_new_tokens=4, eos_token_id=None)
    assert out8.shape == (1, 4)
    with pytest.raises(NotImplementedError):
        model.to(dtype=torch.float16)
