"""GPU: `load_pretrained_model` on a synthetic checkpoint written in the reference's on-disk layout
(llava_arch.py:181-250): llm/ vision_tower/ mm_projector/ region_extractor/ + top config.json."""
import json
import os

import pytest
import torch

from tests.test_host_loader import write_checkpoint as _write_checkpoint
from tests.test_host_loader import write_tiny_tokenizer  # noqa: F401
from tests.util import load_tiny

pytestmark = pytest.mark.gpu


def _ckpt(tmp_path, name="tiny_bf16.npz"):
    """synthetic checkpoint in the reference layout WITH tokenizer files: the checkpoint's embedding table has exactly the
    tokenizer's base vocabulary, so the loader must add <mask>/<depth>, record their ids and grow the tables (builder.py:186-199)."""
    cfgd, dtype, w, inp, ref = load_tiny(name)
    base = 100  # sentencepiece pieces incl. <unk>/<s>/</s> (the trainer hits this size exactly on the toy corpus)
    w = dict(w)
    for k in ("llm.model.embed_tokens.weight", "llm.lm_head.weight"):
        w[k] = w[k][:base].clone()
    cfgd = dict(cfgd, vocab=base)
    root = str(tmp_path / "SpatialRGPT-tiny")
    _write_checkpoint(root, cfgd, w, tokenizer_vocab=base)
    return root, cfgd, dtype, w, inp, ref


def _remap_ids(ids, cfgd, model):
    """fixture ids were minted for the fixture's vocabulary: move <mask>/<depth> to the loaded tokenizer's ids and fold the
    other text ids into the (smaller) base vocabulary; the -200 sentinel stays."""
    mid, did = model.config.mask_token_id, model.config.depth_token_id
    out = ids.clone()
    text = (ids >= 0) & (ids != cfgd["mask_token_id"]) & (ids != cfgd["depth_token_id"])
    out[text] = 3 + ids[text] % (mid - 3)
    out[ids == cfgd["mask_token_id"]] = mid
    out[ids == cfgd["depth_token_id"]] = did
    return out


def test_load_pretrained_model_tokenizer_side_effects_and_roundtrip(tmp_path):
    from spatialrgpt_amd import load_pretrained_model, tokenizer_image_token
    from spatialrgpt_amd.constants import IMAGE_TOKEN_INDEX

    root, cfgd, dtype, w, inp, ref = _ckpt(tmp_path)
    tokenizer, model, image_processor, context_len = load_pretrained_model(root, "SpatialRGPT-tiny")
    assert tokenizer is not None, "the tokenizer written into the checkpoint must load"
    assert context_len == 2048 and image_processor is not None
    assert model.config.enable_region and model.config.enable_depth and model.config.vit_layers_run == cfgd["vit_layers"] - 1
    # builder.py:186-192: <mask>/<depth> added as special tokens, ids recorded on the tower config, tables resized (builder.py:199)
    mid, did = tokenizer.convert_tokens_to_ids("<mask>"), tokenizer.convert_tokens_to_ids("<depth>")
    n_base = len(tokenizer) - 2
    assert (mid, did) == (n_base, n_base + 1)
    vt = model.get_vision_tower()
    assert vt.config.llm_mask_token_id == mid and vt.config.llm_depth_token_id == did
    assert model.config.mask_token_id == mid and model.config.depth_token_id == did
    assert model.engine.w.vocab == len(tokenizer) == model.engine.w.lm_head.shape[0]
    # new embedding rows = mean of the old ones (what resize_token_embeddings initialises them to)
    old = w["llm.model.embed_tokens.weight"].float()
    assert torch.allclose(model.engine.w.embed[mid].float().cpu(), old.mean(0).to(torch.bfloat16).float(), atol=1e-2)
    # a real prompt through the reference's tokenisation path, then generate
    prompt = "<image>\nhow far is region <mask> <depth> from region <mask> <depth> ?"
    ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).cuda()
    assert int((ids == IMAGE_TOKEN_INDEX).sum()) == 1 and int((ids == mid).sum()) == 2 and int((ids == did).sum()) == 2
    model.to(dtype=torch.bfloat16)  # eval_spatial.py:221
    out = model.generate(ids, images=inp["images"].cuda(), depths=inp["depths"].cuda(), masks=[inp["masks"][0][:2].cuda()],
                         do_sample=False, max_new_tokens=4, eos_token_id=None, pad_token_id=tokenizer.pad_token_id)
    assert out.shape == (1, 4) and int(out.max()) < len(tokenizer)
    assert isinstance(tokenizer.batch_decode(out, skip_special_tokens=True)[0], str)
    with pytest.raises(IndexError):  # ids past the table: nn.Embedding semantics, not an out-of-bounds read
        model.generate(torch.tensor([[1, len(tokenizer) + 5]]).cuda(), do_sample=False, max_new_tokens=1)
    with pytest.raises(NotImplementedError):
        load_pretrained_model(root, "x", load_4bit=True)
    # load_8bit -> weight-only fp8 for the streamed LLM matrices (the reference's bitsandbytes int8 slot, builder.py:51-52)
    _, m8, _, _ = load_pretrained_model(root, "SpatialRGPT-tiny", load_8bit=True)
    assert m8.engine.w.llm_weight_format == "fp8" and m8.engine.w.llm_q is not None
    out8 = m8.generate(ids, images=inp["images"].cuda(), depths=inp["depths"].cuda(), masks=[inp["masks"][0][:2].cuda()],
                       do_sample=False, max_new_tokens=4, eos_token_id=None)
    assert out8.shape == (1, 4)


def test_hf_registry_routes_reach_the_engine(tmp_path):
    """llava_llama.py:216-217: AutoConfig / AutoModel know "llava_llama"; builder.py:142-158's own sequence
    (AutoConfig.from_pretrained -> config.resume_path -> LlavaLlamaModel(config=..., low_cpu_mem_usage=True)) builds the engine."""
    from transformers import AutoConfig, AutoModel

    import spatialrgpt_amd
    from spatialrgpt_amd.configuration import LlavaLlamaConfig

    root, cfgd, dtype, w, inp, ref = _ckpt(tmp_path)
    assert spatialrgpt_amd.LlavaLlamaModel.config_class is LlavaLlamaConfig
    config = AutoConfig.from_pretrained(root)
    assert isinstance(config, LlavaLlamaConfig) and config.enable_region and config.model_type == "llava_llama"
    config.resume_path = root
    config.model_dtype = "torch.bfloat16"  # prepare_config_for_eval (builder.py:228-240)
    m1 = spatialrgpt_amd.LlavaLlamaModel(config=config, low_cpu_mem_usage=True)
    m2 = AutoModel.from_pretrained(root, torch_dtype=torch.bfloat16)
    for m in (m1, m2):
        assert isinstance(m, spatialrgpt_amd.LlavaLlamaModel) and m.tokenizer is not None and m.dtype == torch.bfloat16
        assert m.get_vision_tower().image_processor is not None
    ids = _remap_ids(inp["input_ids"], cfgd, m1).cuda()
    kw = dict(images=inp["images"].cuda(), depths=inp["depths"].cuda(), masks=[m_.cuda() for m_ in inp["masks"]], do_sample=False,
              max_new_tokens=4, eos_token_id=None)
    assert torch.equal(m1.generate(ids, **kw), m2.generate(ids, **kw))


def test_reference_callers_dtype_flows(tmp_path):
    """The dtype flow of every reference caller, replayed verbatim against the loaded model: none of them may raise, all of them
    must produce the ids the bf16 flow produces (inputs are cast at the boundary, compute stays in the engine dtype)."""
    from spatialrgpt_amd import load_pretrained_model

    root, cfgd, dtype, w, inp, ref = _ckpt(tmp_path)
    tokenizer, model, image_processor, _ = load_pretrained_model(root, "SpatialRGPT-tiny")
    ids = _remap_ids(inp["input_ids"], cfgd, model)
    images_tensor, depths_tensor, masks = inp["images"].float(), inp["depths"].float(), inp["masks"][0].float()  # processor output: fp32
    input_ids = ids.to(device="cuda", non_blocking=True)
    gen = dict(do_sample=False, temperature=0, top_p=None, num_beams=1, use_cache=True)

    # eval_spatial.py:221-237
    model.to(dtype=torch.bfloat16)
    with torch.inference_mode():
        want = model.generate(input_ids, images=images_tensor.to(dtype=torch.bfloat16, device="cuda", non_blocking=True),
                              depths=depths_tensor.to(dtype=torch.bfloat16, device="cuda", non_blocking=True),
                              masks=[masks.to(dtype=torch.bfloat16, device="cuda", non_blocking=True)], max_new_tokens=6,
                              eos_token_id=None, **gen)
    assert want.shape == (1, 6)

    # eval_region_cls.py:313-325: fp16 images and masks, the model is never cast, no depths
    with torch.inference_mode():
        a16 = model.generate(input_ids, images=images_tensor.to(dtype=torch.float16, device="cuda", non_blocking=True),
                             masks=[masks.to(dtype=torch.float16, device="cuda", non_blocking=True)], max_new_tokens=6,
                             pad_token_id=tokenizer.pad_token_id, eos_token_id=None, **gen)
        abf = model.generate(input_ids, images=images_tensor.to(dtype=torch.bfloat16, device="cuda"),
                             masks=[masks.to(dtype=torch.bfloat16, device="cuda")], max_new_tokens=6, eos_token_id=None, **gen)
    assert torch.equal(a16, abf)  # the fixture's pixels are multiples of 1/32: exact in fp16 and bf16 alike

    # model_vqa.py:68-80: one image, .half().cuda(), no regions in the prompt
    text_ids = input_ids[:, :4]
    text_ids = torch.cat([text_ids[text_ids >= 0][None], torch.tensor([[-200]], device="cuda"), input_ids[:, 4:6].clamp(min=3)], 1)
    with torch.inference_mode():
        v = model.generate(text_ids, images=images_tensor[0].unsqueeze(0).half().cuda(), max_new_tokens=5, eos_token_id=None, **gen)
    assert v.shape == (1, 5)

    # demo/gradio_web_server_multi.py:171-213, both branches: fp16 tensors from process_*, model.to(selected_dtype), lists of tensors
    from spatialrgpt_amd import KeywordsStoppingCriteria

    for selected_dtype in (torch.bfloat16, torch.float16):
        it = images_tensor.to("cuda", dtype=torch.float16)
        dt_ = depths_tensor.to("cuda", dtype=torch.float16)
        mt = masks.to("cuda", dtype=torch.float16)
        with pytest.warns(UserWarning) if selected_dtype == torch.float16 else _nullcontext():
            model.to(dtype=selected_dtype)
        model._warned_dtype = False
        sc = KeywordsStoppingCriteria(["</s>"], tokenizer, input_ids)
        with torch.inference_mode():
            d = model.generate(input_ids, images=[it.to(dtype=selected_dtype).cuda()], depths=[dt_.to(dtype=selected_dtype).cuda()],
                               masks=[mt], do_sample=False, temperature=0, max_new_tokens=6, use_cache=True, stopping_criteria=[sc],
                               eos_token_id=None)
        assert torch.equal(d[:, :d.shape[1]], want[:, :d.shape[1]]) and d.shape[1] >= 1
    assert model.half() is model and model.dtype == torch.bfloat16  # documented policy: interface request only

    # module-level contract of the tower (vision_encoder.py:127-130): features return in the caller's image dtype
    f16 = model.get_vision_tower()(images_tensor.half().cuda())
    assert f16.dtype == torch.float16
    # fp32 images into a bf16 engine (the processor's default output): same result as casting first
    st32, stbf = {}, {}
    model.engine.prepare_inputs(input_ids, images_tensor.cuda(), depths_tensor.cuda(), [masks.cuda()], None, stages=st32)
    model.engine.prepare_inputs(input_ids, images_tensor.cuda().bfloat16(), depths_tensor.cuda().bfloat16(), [masks.cuda().bfloat16()],
                                None, stages=stbf)
    assert torch.equal(st32["inputs_embeds"], stbf["inputs_embeds"])


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_generation_config_eos_list_stops_and_pads_like_hf(tmp_path):
    """the reference generates through `self.llm.generate` (llava_llama.py:212) on an LLM whose generation config
    came from <ckpt>/llm/generation_config.json -- a LIST of EOS ids for Llama-3 -- and eval_spatial.py:224-236 passes neither
    eos / pad ids nor stopping criteria.  A checkpoint whose generation config lists two EOS ids must stop a row on EITHER,
    pad the finished row with pad_token_id (default: the FIRST eos id), cut the output at the step where the last row finished,
    and the sampling path must honour the same ids."""
    from spatialrgpt_amd import load_pretrained_model

    root, cfgd, dtype, w, inp, ref = _ckpt(tmp_path)
    tokenizer, model, _, _ = load_pretrained_model(root, "SpatialRGPT-tiny")
    assert model.config.eos_token_id == 2  # from llm/config.json (no generation_config.json yet)
    ids0 = _remap_ids(inp["input_ids"], cfgd, model)
    ids1 = ids0.clone()
    ids1[0, 1] = 3 + (int(ids1[0, 1]) + 11) % 50  # a second, different request
    input_ids = torch.cat([ids0, ids1], 0).cuda()
    images = inp["images"].cuda().bfloat16().repeat(2, 1, 1, 1)
    depths = inp["depths"].cuda().bfloat16().repeat(2, 1, 1, 1)
    masks = [inp["masks"][0].cuda().bfloat16(), inp["masks"][0].cuda().bfloat16()]
    kw = dict(images=images, depths=depths, masks=masks)
    G = 24
    free = model.generate(input_ids, do_sample=False, max_new_tokens=G, eos_token_id=None, **kw).cpu()
    assert free.shape == (2, G)

    def first_new(row, start):  # first step >= start whose id did not occur earlier in the row
        for j in range(start, G):
            if int(free[row, j]) not in free[row, :j].tolist():
                return j
        pytest.skip("degenerate free-running ids on this fixture")

    j0, j1 = first_new(0, 2), first_new(1, 9)
    eos = [int(free[1, j1]), int(free[0, j0])]  # row 1's stopper listed FIRST: the default pad id must be eos[0]

    def expect(ids, eos_ids, pad, max_new):
        stops = []
        for b in range(ids.shape[0]):
            hit = [j for j in range(max_new) if int(ids[b, j]) in eos_ids]
            stops.append(hit[0] + 1 if hit else max_new)
        n = max(stops)
        out = ids[:, :n].clone()
        for b, s_ in enumerate(stops):
            out[b, s_:] = pad
        return out

    json.dump({"bos_token_id": 1, "eos_token_id": eos, "do_sample": True, "temperature": 0.6, "top_p": 0.9},
              open(os.path.join(root, "llm", "generation_config.json"), "w"))
    del model
    tokenizer, model, _, _ = load_pretrained_model(root, "SpatialRGPT-tiny")
    assert model.config.eos_token_id == eos and model.generation_defaults()["top_p"] == 0.9
    # eval_spatial.py:224-236 verbatim keywords: nothing about eos / pad / stopping criteria
    ev = dict(do_sample=False, temperature=0, top_p=None, num_beams=1, max_new_tokens=G, use_cache=True)
    got = model.generate(input_ids, **kw, **ev).cpu()
    want = expect(free, eos, eos[0], G)
    assert want.shape[1] < G, "the fixture must stop early for this test to mean anything"
    assert torch.equal(got, want), (got, want)
    # an explicit pad id wins; a single row stops on ITS id and is not padded
    got = model.generate(input_ids, pad_token_id=0, **kw, **ev).cpu()
    assert torch.equal(got, expect(free, eos, 0, G))
    one = model.generate(input_ids[:1], images=images[:1], depths=depths[:1], masks=masks[:1], **ev).cpu()
    assert torch.equal(one, free[:1, :j0 + 1]) or int(one[0, -1]) in eos
    # eos_token_id=None (explicit) switches token stopping off again; an int overrides the list
    assert torch.equal(model.generate(input_ids, eos_token_id=None, **kw, **ev).cpu(), free)
    assert torch.equal(model.generate(input_ids, eos_token_id=eos[1], **kw, **ev).cpu(), expect(free, [eos[1]], eos[1], G))
    # sampling path (demo, gradio_web_server_multi.py:202-213): top_k=1 makes it deterministic = greedy; same stop / pad rules
    smp = model.generate(input_ids, do_sample=True, temperature=0.7, top_k=1, max_new_tokens=G, **kw).cpu()
    assert torch.equal(smp, want), (smp, want)


def test_reference_written_checkpoint_reproduces_reference_ids_bit_exactly():
    """tests/golden/ckpt_tiny/ is the output of the REFERENCE's own `save_pretrained` (llava_arch.py:181-250);
    tests/golden/ckpt_tiny_kat.npz holds the ids the reference's `generate()` produced on that very model.  Loaded through
    `load_pretrained_model` (fp32 engine) and through the HF-registry route, the engine reproduces them bit for bit."""
    import numpy as np
    from transformers import AutoConfig

    import spatialrgpt_amd
    from spatialrgpt_amd import load_pretrained_model
    from tests.util import GOLD

    root = os.path.join(GOLD, "ckpt_tiny")
    z = np.load(os.path.join(GOLD, "ckpt_tiny_kat.npz"))
    ids = torch.from_numpy(z["input_ids"]).cuda()
    images = (torch.from_numpy(z["images_q32"].astype(np.float32)) / 32).cuda()
    depths = (torch.from_numpy(z["depths_q32"].astype(np.float32)) / 32).expand(-1, 3, -1, -1).contiguous().cuda()
    masks = [torch.from_numpy(z["masks_u8"][i].astype(np.float32)).cuda() for i in range(z["masks_u8"].shape[0])]
    want = torch.from_numpy(z["new_ids"])
    tokenizer, model, image_processor, context_len = load_pretrained_model(root, "SpatialRGPT-tiny", dtype=torch.float32)
    assert model.dtype == torch.float32 and context_len == 2048 and len(tokenizer) == 122
    assert model.config.eos_token_id == [2, 9]  # llm/generation_config.json as the reference saved it
    kw = dict(images=images, depths=depths, masks=masks, do_sample=False, max_new_tokens=want.shape[1], use_cache=True,
              eos_token_id=None, pad_token_id=0, min_new_tokens=want.shape[1])  # the keywords the golden run used
    got = model.generate(ids, **kw).cpu()
    assert torch.equal(got, want), (got, want)
    # builder.py:142-158's own sequence on the same directory (nested sub-configs in the top-level config.json)
    config = AutoConfig.from_pretrained(root)
    config.resume_path = root
    config.model_dtype = "torch.float32"
    m2 = spatialrgpt_amd.LlavaLlamaModel(config=config, low_cpu_mem_usage=True)
    assert m2.dtype == torch.float32 and torch.equal(m2.generate(ids, **kw).cpu(), want)


def test_component_factories_compose_to_the_whole_model_bit_for_bit():
    """SURVEY 8b "factories": build_vision_tower / build_region_extractor / build_mm_projector / build_llm_and_tokenizer
    (llava/model/{multimodal_encoder,region_extractor,multimodal_projector,language_model}/builder.py) on the component directories
    the REFERENCE's save_pretrained wrote (tests/golden/ckpt_tiny/*).  Composed by hand the way llava_arch.py:387-411 composes them,
    every stage tensor equals the whole model's bit for bit, and the stand-alone LLM's generate(inputs_embeds=...) returns the
    reference's ids (ckpt_tiny_kat.npz).  Each factory engine holds ONLY its component: the others are refused."""
    import numpy as np

    from spatialrgpt_amd import load_pretrained_model
    from spatialrgpt_amd.factories import build_llm_and_tokenizer, build_mm_projector, build_region_extractor, build_vision_tower
    from tests.util import GOLD

    root = os.path.join(GOLD, "ckpt_tiny")
    z = np.load(os.path.join(GOLD, "ckpt_tiny_kat.npz"))
    ids = torch.from_numpy(z["input_ids"]).cuda()
    images = (torch.from_numpy(z["images_q32"].astype(np.float32)) / 32).cuda()
    depths = (torch.from_numpy(z["depths_q32"].astype(np.float32)) / 32).expand(-1, 3, -1, -1).contiguous().cuda()
    masks = [torch.from_numpy(z["masks_u8"][i].astype(np.float32)).cuda() for i in range(z["masks_u8"].shape[0])]
    want = torch.from_numpy(z["new_ids"])
    _, whole, _, _ = load_pretrained_model(root, "SpatialRGPT-tiny", dtype=torch.float32)
    st = {}
    embeds, _, _ = whole.engine.prepare_inputs(ids, images, depths, masks, None, stages=st)

    cfgns = type("Cfg", (), dict(mm_vision_select_layer=-2, mm_vision_select_feature="cls_patch"))()
    tower = build_vision_tower(os.path.join(root, "vision_tower"), cfgns, dtype=torch.float32)
    assert tower.is_loaded and tower.image_processor is not None and cfgns.mm_hidden_size == tower.hidden_size == whole.config.vit_hidden
    rex = build_region_extractor(os.path.join(root, "region_extractor"), cfgns, dtype=torch.float32)
    proj = build_mm_projector(os.path.join(root, "mm_projector"), cfgns, dtype=torch.float32)
    llm, tok = build_llm_and_tokenizer(os.path.join(root, "llm"), cfgns, attn_implementation="flash_attention_2", dtype=torch.float32)
    assert cfgns.hidden_size == whole.config.hidden and tok is not None

    # llava_arch.py:387-411
    tower_features = tower(images)
    depth_features = tower(depths)
    hres, lres = rex.feature_refinement(tower_features)
    mask_embeds, depth_embeds = rex(hres, depth_features, masks)
    image_features = proj(lres)
    assert torch.equal(tower_features, st["tower_features"]) and torch.equal(depth_features, st["depth_features"])
    assert torch.equal(hres, st["hres"]) and torch.equal(lres, st["lres"])
    assert all(torch.equal(a, b) for a, b in zip(mask_embeds, st["mask_embeds"]))
    assert all(torch.equal(a, b) for a, b in zip(depth_embeds, st["depth_embeds"]))
    assert torch.equal(image_features, st["image_features"])
    # the stand-alone LLM on the whole model's spliced embeddings: the reference's ids
    got = llm.generate(inputs_embeds=embeds, attention_mask=None, do_sample=False, max_new_tokens=want.shape[1], use_cache=True,
                       eos_token_id=None, pad_token_id=0, min_new_tokens=want.shape[1]).cpu()
    assert torch.equal(got, want)
    assert torch.equal(llm.get_input_embeddings()(ids.clamp_min(0)), whole.get_input_embeddings()(ids.clamp_min(0)))
    # a component engine refuses what it does not hold
    with pytest.raises(RuntimeError):
        rex._eng.mm_projector(lres)
    with pytest.raises(RuntimeError):
        proj._eng.vit(images)
    # a live module's state dict instead of a directory (how ONE component of a running reference model is replaced)
    from safetensors.torch import load_file

    sd = load_file(os.path.join(root, "region_extractor", "model.safetensors"))
    rex2 = build_region_extractor("regiongpt", cfgns, state_dict=sd, dtype=torch.float32)
    assert all(torch.equal(a, b) for a, b in zip(rex2(hres, depth_features, masks)[0], mask_embeds))
    with pytest.raises(ValueError):
        build_region_extractor("regiongpt", cfgns)  # a fresh random component is the training path
    with pytest.raises(ValueError):
        build_mm_projector("linear", cfgns, state_dict=sd)
    # the Auto-class route of the component directories (base_projector.py:97-98, base_extractor.py:176-177)
    from transformers import AutoModel

    proj2 = AutoModel.from_pretrained(os.path.join(root, "mm_projector"), torch_dtype=torch.float32)
    assert torch.equal(proj2(lres), image_features)
    rex3 = AutoModel.from_pretrained(os.path.join(root, "region_extractor"), torch_dtype=torch.float32)
    assert torch.equal(rex3.feature_refinement(tower_features)[0], hres)


@pytest.mark.parametrize("fmt", ["native", "fp8"])
def test_resize_token_embeddings_on_the_live_model(fmt):
    """builder.py:199 calls `model.resize_token_embeddings(len(tokenizer))` on the LIVE model: embed_tokens and lm_head follow (common
    rows kept bit for bit, new rows = the mean row), generation runs with ids in the new range, shrinking back restores the logits."""
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel
    from spatialrgpt_amd.weights import synth_state_dict

    cfg = SrgptConfig(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=42, patch_size=14, hidden=128, inter=256,
                      layers=2, heads=4, kv_heads=2, vocab=300, mask_token_id=298, depth_token_id=299)
    dt = torch.bfloat16
    model = LlavaLlamaModel(cfg, synth_state_dict(cfg, seed=4, dtype=dt, device="cuda"), device="cuda", dtype=dt, rope_positions=128,
                            llm_weight_format=fmt)
    w = model.engine.w
    ids = torch.tensor([[1, 17, 45, 250, 7]], device="cuda")
    am = torch.ones_like(ids)
    base = model(input_ids=ids, attention_mask=am, use_cache=False).logits.clone()
    emb0 = w.embed.clone()
    head0 = (w.dequantised("lm_head") if fmt == "fp8" else w.lm_head).clone()
    model.resize_token_embeddings(305)
    assert w.vocab == 305 == w.embed.shape[0] and model.config.vocab_size == 305
    assert torch.equal(w.embed[:300], emb0) and torch.equal(w.embed[300:], emb0.float().mean(0, keepdim=True).to(dt).expand(5, -1))
    head1 = w.dequantised("lm_head") if fmt == "fp8" else w.lm_head
    assert head1.shape[0] == 305 and torch.equal(head1[:300], head0)
    grown = model(input_ids=ids, attention_mask=am, use_cache=False).logits
    assert grown.shape[-1] == 305 and torch.equal(grown[..., :300], base)
    out = model.generate(torch.tensor([[1, 304, 302, 9]], device="cuda"), do_sample=False, max_new_tokens=3, eos_token_id=None)
    assert out.shape == (1, 3) and int(out.max()) < 305
    model.resize_token_embeddings(300)
    assert torch.equal(model(input_ids=ids, attention_mask=am, use_cache=False).logits, base)
    with pytest.raises(IndexError):
        model.generate(torch.tensor([[1, 302]], device="cuda"), do_sample=False, max_new_tokens=1)
