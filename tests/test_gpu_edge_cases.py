"""GPU: input-contract edge cases of generate()/prepare_inputs (llava_arch.py:333-650) against the oracle on the CPU,
and a true-width parity check (VILA1.5-8B layer geometry, truncated depth) in bf16."""
import pytest
import torch

from tests.util import assert_close, load_tiny

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(name="tiny_fp32.npz"):
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel

    cfgd, dtype, w, inp, ref = load_tiny(name)
    cfg = SrgptConfig.from_dict(cfgd)
    ocfg = so.SrgptConfig(**{k: v for k, v in cfgd.items() if k in so.SrgptConfig.__dataclass_fields__})
    model = LlavaLlamaModel(cfg, dict(w), device=DEV, dtype=dtype, rope_positions=1024)
    return so, ocfg, model, w, inp


def _embeds(model, ids, images, depths, masks, am=None):
    e, a, lens = model.engine.prepare_inputs(ids.to(DEV), images if isinstance(images, list) else images.to(DEV),
                                             None if depths is None else (depths if isinstance(depths, list) else depths.to(DEV)),
                                             None if masks is None else [None if m is None else m.to(DEV) for m in masks],
                                             None if am is None else am.to(DEV))
    return e, a, lens


def test_depths_none_and_masks_none():
    so, ocfg, model, w, inp = _both()
    ids, im, dp, mk = inp["input_ids"], inp["images"], inp["depths"], inp["masks"]
    # no depth maps: only <mask> rows are replaced, <depth> ids keep their token embedding (llava_arch.py:406-407)
    ref, _, _, _ = so.prepare_inputs(w, ocfg, ids, im, None, mk)
    got, _, _ = _embeds(model, ids, im, None, mk)
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "depths=None")
    # masks=[None]: nothing replaced, a diagnostic is printed (llava_arch.py:476-477)
    ref, _, _, _ = so.prepare_inputs(w, ocfg, ids, im, dp, [None])
    got, _, _ = _embeds(model, ids, im, dp, [None])
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "masks=[None]")
    ref, _, _, _ = so.prepare_inputs(w, ocfg, ids, im, dp, None)
    got, _, _ = _embeds(model, ids, im, dp, None)
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "masks=None")


def test_image_containers_and_attention_mask():
    so, ocfg, model, w, inp = _both()
    ids, im, dp, mk = inp["input_ids"], inp["images"], inp["depths"], inp["masks"]
    ref, _, _, _ = so.prepare_inputs(w, ocfg, ids, im, dp, mk)
    got, _, _ = _embeds(model, ids, [im.to(DEV)], [dp.to(DEV)], mk)  # list of [1,3,S,S]
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "list images")
    got, _, _ = _embeds(model, ids, im[None], dp[None], mk)  # 5-D [B, n, 3, S, S]
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "5-D images")
    am = torch.ones_like(ids)
    got, a, _ = _embeds(model, ids, im, dp, mk, am)
    assert a is not None and a.dtype == am.dtype and a.shape == got.shape[:2] and bool(a.bool().all())
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "explicit attention mask")


def test_two_images_one_prompt_and_masks_follow_image_index():
    """two <image> sentinels in one sample: T = P - 2 + 2*196; region embeds come from the sample's first image."""
    so, ocfg, model, w, inp = _both()
    ids, im, dp, mk = inp["input_ids"], inp["images"], inp["depths"], inp["masks"]
    ids2 = torch.cat([ids[:, :8], torch.tensor([[-200]]), ids[:, 8:]], dim=1)
    im2, dp2 = torch.cat([im, dp], 0), torch.cat([dp, im], 0)
    mk2 = [mk[0], mk[0].flip(0)]
    ref, _, _, _ = so.prepare_inputs(w, ocfg, ids2, im2, dp2, mk2)
    got, _, lens = _embeds(model, ids2, im2, dp2, mk2)
    assert got.shape[1] == ids2.shape[1] - 2 + 2 * 196 == lens[0]
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "two images")


def test_ragged_batch_right_and_left_padding():
    so, ocfg, model, w, inp = _both()
    ids, im, dp, mk = inp["input_ids"], inp["images"], inp["depths"], inp["masks"]
    P = ids.shape[1]
    short = ids[:, :P - 3]
    pad = torch.zeros((1, 3), dtype=ids.dtype)
    ids_b = torch.cat([ids, torch.cat([short, pad], 1)], 0)
    am = torch.ones_like(ids_b)
    am[1, P - 3:] = 0
    im2, dp2, mk2 = torch.cat([im, im], 0), torch.cat([dp, dp], 0), [mk[0], mk[0]]
    ref, ram, _, _ = so.prepare_inputs(w, ocfg, ids_b, im2, dp2, mk2, am)
    got, gam, lens = _embeds(model, ids_b, im2, dp2, mk2, am)
    assert lens == [P - 1 + 196, P - 4 + 196]
    assert torch.equal(gam.cpu(), ram)
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "ragged, right padded")
    assert float(got[1, lens[1]:].abs().max()) == 0.0  # padding rows are zeros
    # generate() on the ragged batch == each row on its own
    out = model.generate(ids_b.to(DEV), images=im2.to(DEV), depths=dp2.to(DEV), masks=[m.to(DEV) for m in mk2],
                         attention_mask=am.to(DEV), do_sample=False, max_new_tokens=5, eos_token_id=None)
    one = model.generate(ids.to(DEV), images=im.to(DEV), depths=dp.to(DEV), masks=[mk[0].to(DEV)], do_sample=False,
                         max_new_tokens=5, eos_token_id=None)
    two = model.generate(short.to(DEV), images=im.to(DEV), depths=dp.to(DEV), masks=[mk[0].to(DEV)], do_sample=False,
                         max_new_tokens=5, eos_token_id=None)
    assert torch.equal(out[0:1], one) and torch.equal(out[1:2], two)
    # left padding (llm.config.tokenizer_padding_side == "left", llava_arch.py:570-590)
    model.engine.cfg.padding_side = ocfg.padding_side = "left"
    ref, ram, _, _ = so.prepare_inputs(w, ocfg, ids_b, im2, dp2, mk2, am)
    got, gam, _ = _embeds(model, ids_b, im2, dp2, mk2, am)
    assert torch.equal(gam.cpu(), ram)
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "ragged, left padded")


def test_truncation_warns_like_reference():
    so, ocfg, model, w, inp = _both()
    ids, im, dp, mk = inp["input_ids"], inp["images"], inp["depths"], inp["masks"]
    model.engine.cfg.tokenizer_model_max_length = ocfg.tokenizer_model_max_length = 100
    with pytest.warns(UserWarning, match="Inputs truncated!"):
        got, _, lens = _embeds(model, ids, im, dp, mk)
    with pytest.warns(UserWarning):
        ref, _, _, _ = so.prepare_inputs(w, ocfg, ids, im, dp, mk)
    assert got.shape[1] == 100 and lens == [100]
    assert_close(got, ref, 2e-4 * float(ref.abs().max()), 0, "truncated")


def test_sampling_path_runs_and_respects_eos():
    so, ocfg, model, w, inp = _both()
    d = {k: (v.to(DEV) if torch.is_tensor(v) else [m.to(DEV) for m in v]) for k, v in inp.items()}
    torch.manual_seed(0)
    out = model.generate(d["input_ids"], images=d["images"], depths=d["depths"], masks=d["masks"], do_sample=True, temperature=0.2,
                         top_p=0.9, max_new_tokens=6, eos_token_id=None)
    assert out.shape == (1, 6) and int(out.min()) >= 0 and int(out.max()) < model.config.vocab
    # temperature -> 0 limit: top_k = 1 must reproduce greedy
    g = model.generate(d["input_ids"], images=d["images"], depths=d["depths"], masks=d["masks"], do_sample=False, max_new_tokens=6,
                       eos_token_id=None)
    s = model.generate(d["input_ids"], images=d["images"], depths=d["depths"], masks=d["masks"], do_sample=True, temperature=1.0,
                       top_k=1, max_new_tokens=6, eos_token_id=None)
    assert torch.equal(g, s)


def test_every_flag_combination_of_the_eval_clis_generates():
    """eval_spatial.py:224-236 / eval_region_cls.py:311-325 / model_vqa.py:66-80 pass `do_sample = temperature > 0`, `temperature`
    (default 0.2), `top_p` (default None), `num_beams` (default 1): every combination must generate (VERDICT r4 missing #2: the
    defaults + `--num_beams 3` are HF beam-SAMPLE and raised NotImplementedError).  Beam-sample is seeded by torch.manual_seed like
    HF's multinomial, stays inside the vocabulary and the token budget; beam search (temperature 0) is deterministic."""
    so, ocfg, model, w, inp = _both()
    d = {k: (v.to(DEV) if torch.is_tensor(v) else [m.to(DEV) for m in v]) for k, v in inp.items()}
    kw = dict(images=d["images"], depths=d["depths"], masks=d["masks"])
    G = 6
    for temperature in (0.2, 0.0):
        for top_p in (None, 0.9):
            for num_beams in (1, 3):
                torch.manual_seed(3)
                a = model.generate(d["input_ids"], do_sample=temperature > 0, temperature=temperature, top_p=top_p, num_beams=num_beams,
                                   max_new_tokens=G, use_cache=True, eos_token_id=None, **kw)
                assert a.shape == (1, G) and int(a.min()) >= 0 and int(a.max()) < model.config.vocab, (temperature, top_p, num_beams)
                torch.manual_seed(3)
                b = model.generate(d["input_ids"], do_sample=temperature > 0, temperature=temperature, top_p=top_p, num_beams=num_beams,
                                   max_new_tokens=G, use_cache=True, eos_token_id=None, **kw)
                assert torch.equal(a, b), (temperature, top_p, num_beams)
    # beam-sample with an EOS list: a stopped row ends with eos_token_id[0], the batch is cut at the longest row
    free = model.generate(d["input_ids"], do_sample=False, num_beams=3, max_new_tokens=G, eos_token_id=None, **kw)
    eos = [int(free[0, 2]), 5]
    torch.manual_seed(1)
    o = model.generate(d["input_ids"], do_sample=True, temperature=0.2, num_beams=3, max_new_tokens=G, eos_token_id=eos, pad_token_id=0, **kw)
    assert 1 <= o.shape[1] <= G
    row = o[0].tolist()
    hit = [j for j, t in enumerate(row) if t in eos]
    assert not hit or (row[hit[0]] == eos[0] and all(t == 0 for t in row[hit[0] + 1:]))


@pytest.mark.parametrize("geom", ["vila15_8b", "llama2_7b", "sheared_3b", "clip_l14_336", "vila15_8b-fp8"])
def test_true_width_truncated_depth_bf16_vs_oracle(geom, batch=1, regions=8, prompt_len=64, seed=2):
    """The three LLM layer geometries of the reference's recipes at TRUE width -- VILA1.5-8B (hidden 4096, GQA 32/8, inter
    14336), Llama-2-7B (MHA 32/32, inter 11008), Sheared-LLaMA-2.7B (hidden 2560, 20 heads, inter 6912) -- behind the
    SigLIP-so400m-width tower (plus the CLIP-L/14-336 tower in front of the 8B geometry), with 2 LLM / 2 ViT layers and a 16k vocab, bf16, one full request: stage tensors within bf16
    tolerance of the oracle, decode logits teacher-forced (every step) with a margin-aware argmax check."""
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel

    kw = dict(vit_layers=3, layers=2, vocab=16386, mask_token_id=16384, depth_token_id=16385)
    if geom == "llama2_7b":
        kw.update(hidden=4096, inter=11008, heads=32, kv_heads=32, rope_theta=10000.0)
    elif geom == "sheared_3b":
        kw.update(hidden=2560, inter=6912, heads=20, kv_heads=20, rope_theta=10000.0)
    elif geom == "clip_l14_336":  # CLIP-L/14-336 tower (class token, pre-LN, quick-GELU, "patch" features: 576 tokens -> 96x96 hres)
        kw.update(vit_hidden=1024, vit_inter=4096, vit_heads=16, image_size=336, vit_eps=1e-5, tower="clip", select_feature="patch")
    ocfg = so.SrgptConfig(**kw)
    w = so.synth_weights(ocfg, seed=11, dtype=torch.bfloat16)
    ids, images, depths, masks = so.synth_inputs(ocfg, batch=batch, regions=regions, prompt_len=prompt_len, seed=seed, dtype=torch.bfloat16)
    fp8 = geom.endswith("-fp8")  # weight-only fp8 LLM matrices: the oracle runs on dequant(quant(W)), the engine quantises itself
    model = LlavaLlamaModel(SrgptConfig(**kw), dict(w), device=DEV, dtype=torch.bfloat16, rope_positions=1024,
                            llm_weight_format="fp8" if fp8 else "native")
    if fp8:
        w = so.fp8_dequantised_weights(w)
    torch.set_num_threads(16)
    G = 6
    ref_ids, st = so.generate(w, ocfg, ids, images, depths, masks, max_new_tokens=G, return_stages=True, model_dtype=torch.bfloat16)
    got = {}
    emb, _, _ = model.engine.prepare_inputs(ids.to(DEV), images.to(DEV), depths.to(DEV), [m.to(DEV) for m in masks], None, stages=got)
    assert emb.shape == (batch, prompt_len - 1 + 196, kw.get("hidden", 4096))

    def chk(a, b, what, rel=4e-2):
        assert_close(a, b, rel * (float(b.float().abs().max()) + 1e-6), 0, what)

    chk(got["tower_features"], st["tower_features"], "tower (2 SigLIP-so400m layers)")
    chk(got["hres"], st["hres"], "hres 11664 x 1152")
    chk(torch.stack(got["mask_embeds"]), torch.stack(st["mask_embeds"]), "mask embeds")
    chk(torch.stack(got["depth_embeds"]), torch.stack(st["depth_embeds"]), "depth embeds")
    chk(got["image_features"], st["image_features"], "projector")
    chk(emb, st["inputs_embeds"], "inputs_embeds")
    stt, logits, _ = model.engine.prefill(st["inputs_embeds"].to(DEV), max_new=G + 1, all_logits=True)
    chk(logits, st["prefill_logits"], f"prefill logits (T = {emb.shape[1]})")
    # decode path under teacher forcing with the oracle's ids: every step compared, no exit at the first flip
    from tests.util import logit_parity_report, teacher_forced_decode_logits

    dec = teacher_forced_decode_logits(model.engine, stt, ref_ids)
    r = logit_parity_report(dec, st["step_logits"].float(), 4e-2, f"{geom}: decode logits, teacher forced")
    print("\nPARITY", r)
    assert r["max_abs_over_range"] <= 4e-2 and r["argmax_disagree_out_of_margin"] == 0, r


def test_device_preprocessing_equals_host_path():
    """process_images_device / process_regions_device (HIP kernels on raw uint8) == the host path
    (PIL bicubic + HF-style rescale/normalise; cv2-style nearest) bit for bit, in fp32 and after the bf16 cast."""
    import numpy as np
    from PIL import Image
    from types import SimpleNamespace

    from spatialrgpt_amd.mm_utils import (SrgptImageProcessor, process_images, process_images_device, process_regions,
                                          process_regions_device)

    rng = np.random.default_rng(3)
    proc = SrgptImageProcessor(size=384)
    for mode in ("resize", "pad", None):
        cfg = SimpleNamespace(image_aspect_ratio=mode, image_processor=proc)
        ims = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in [(480, 640), (200, 150), (384, 384)]]
        if mode == "resize":
            # the reference resizes with PIL first (mm_utils.py:441: Image.resize default = bicubic), then the processor
            # sees an image of the final size -> one bicubic pass, which is what the device path runs
            pass
        ref = process_images(ims, proc, cfg)
        got32 = process_images_device(ims, proc, cfg, device=DEV, dtype=torch.float32)
        assert got32.shape == (3, 3, 384, 384)
        assert torch.equal(got32.cpu(), ref), f"mode {mode}: max diff {float((got32.cpu() - ref).abs().max())}"
        got16 = process_images_device(ims, proc, cfg, device=DEV, dtype=torch.bfloat16)
        assert torch.equal(got16.cpu(), ref.to(torch.bfloat16))
    cfg = SimpleNamespace(image_aspect_ratio="resize", image_processor=proc)
    masks = [(rng.random((480, 640)) > 0.7).astype(np.uint8) for _ in range(5)]
    refm = process_regions(masks, proc, cfg)
    gotm = process_regions_device(masks, proc, cfg, device=DEV, dtype=torch.float32)
    assert gotm.shape == (5, 384, 384) and torch.equal(gotm.cpu(), refm)
    # sizes at which floor(dst * in / out) and OpenCV's floor(dst * (1 / (out / in))) pick different source pixels
    # (tests/golden/cv2_nearest_kat.json): host and device follow the published resizeNN arithmetic
    for h, w in [(72, 76), (333, 500), (1080, 1920)]:
        mk = [(rng.random((h, w)) > 0.5).astype(np.uint8) * 255 for _ in range(2)]
        assert torch.equal(process_regions_device(mk, proc, cfg, device=DEV, dtype=torch.float32).cpu(), process_regions(mk, proc, cfg))
    # pad mode (mm_utils.py:505-531): zero square + the processor's bicubic resize of the one-channel square -- soft edges; the
    # device path never materialises the square.  Landscape, portrait, square, 0/1 and 0/255 masks, fp32 and bf16.
    cfgp = SimpleNamespace(image_aspect_ratio="pad", image_processor=proc)
    for (h, w), hi in [((480, 640), 1), ((640, 480), 255), ((333, 500), 255), ((384, 384), 1), ((200, 731), 255), ((97, 97), 255)]:
        mk = [(rng.random((h, w)) > 0.6).astype(np.uint8) * hi for _ in range(3)]
        mk[1][h // 4:h // 2, w // 4:w // 2] = hi  # a solid box: interior stays `hi`, the rim goes soft
        refp = process_regions(mk, proc, cfgp)
        gotp = process_regions_device(mk, proc, cfgp, device=DEV, dtype=torch.float32)
        assert gotp.shape == (3, 384, 384) and torch.equal(gotp.cpu(), refp), ((h, w), float((gotp.cpu() - refp).abs().max()))
        assert torch.equal(process_regions_device(mk, proc, cfgp, device=DEV, dtype=torch.bfloat16).cpu(), refp.to(torch.bfloat16))
        if hi == 255 and h != w:
            assert 0 < float(((refp > 0) & (refp < 255)).float().mean()), "pad mode should produce soft (bicubic) mask edges"
    with pytest.raises(NotImplementedError):
        process_regions_device(masks, proc, SimpleNamespace(image_aspect_ratio="crop", image_processor=proc), device=DEV)


def test_raw_uint8_masks_go_straight_into_region_pooling():
    """SURVEY 8f-2, literal wording: raw uint8 [K, H, W] masks handed to generate() / the pooling kernel (nearest resize to the
    processor size, float(uint8) and the bilinear resample fused in srgpt_region_pool_u8) == the host path
    (process_regions -> float masks -> MaskPooling) bit for bit, for both feature grids (108^2 refined RGB, 27^2 depth)."""
    import numpy as np
    from types import SimpleNamespace

    from spatialrgpt_amd import ops
    from spatialrgpt_amd.mm_utils import SrgptImageProcessor, process_regions

    rng = np.random.default_rng(5)
    proc = SrgptImageProcessor(size=384)
    cfgp = SimpleNamespace(image_aspect_ratio="resize", image_processor=proc)
    raw = [(rng.random((480, 640)) > 0.6).astype(np.uint8) for _ in range(5)] + [np.zeros((480, 640), np.uint8)]
    host = process_regions(raw, proc, cfgp)  # float [6, 384, 384] (cv2-nearest restatement + processor)
    raw_t = torch.from_numpy(np.stack(raw, 0)).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    for dtype in (torch.bfloat16, torch.float32):
        for fw in (108, 27):
            feat = torch.randn((fw * fw, 64), generator=g, device=DEV).to(dtype)
            want = ops.region_pool(feat, host.to(DEV))
            got = ops.region_pool_u8(feat, raw_t, 384)
            assert torch.equal(got, want), f"{dtype} fw={fw}: max diff {float((got.float() - want.float()).abs().max())}"
    # through the model surface: generate() with uint8 masks == generate() with the host-processed float masks
    so, ocfg, model, w, inp = _both()
    S = model.config.image_size
    rawm = (rng.random((inp["masks"][0].shape[0], 200, 300)) > 0.5).astype(np.uint8)
    proc2 = SrgptImageProcessor(size=S)
    hostm = process_regions(list(rawm), proc2, SimpleNamespace(image_aspect_ratio="resize", image_processor=proc2))
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp.items()}
    a = model.generate(d["input_ids"], images=d["images"], depths=d["depths"], masks=[torch.from_numpy(rawm).to(DEV)],
                       do_sample=False, max_new_tokens=5, eos_token_id=None)
    b = model.generate(d["input_ids"], images=d["images"], depths=d["depths"], masks=[hostm.to(DEV)], do_sample=False,
                       max_new_tokens=5, eos_token_id=None)
    assert torch.equal(a, b)
    # pad-mode models: raw uint8 masks take the fused pad + bicubic kernel, then the float-mask pooling == host process_regions(pad)
    model.engine.cfg.image_aspect_ratio = "pad"
    try:
        hostp = process_regions(list(rawm), proc2, SimpleNamespace(image_aspect_ratio="pad", image_processor=proc2))
        a = model.generate(d["input_ids"], images=d["images"], depths=d["depths"], masks=[torch.from_numpy(rawm).to(DEV)],
                           do_sample=False, max_new_tokens=5, eos_token_id=None)
        b = model.generate(d["input_ids"], images=d["images"], depths=d["depths"], masks=[hostp.to(DEV)], do_sample=False,
                           max_new_tokens=5, eos_token_id=None)
        assert torch.equal(a, b)
        feat = torch.randn((108 * 108, 64), generator=g, device=DEV).to(model.engine.dtype)
        mp = model.engine.mask_pooling(feat[None], [torch.from_numpy(rawm).to(DEV)])[0]
        assert torch.equal(mp, ops.region_pool(feat, hostp.to(DEV)))
    finally:
        model.engine.cfg.image_aspect_ratio = "resize"
