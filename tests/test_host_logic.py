"""CPU: host-side logic of the product package (no kernels): tokenizer_image_token against the reference's own
outputs (golden), mask preprocessing, config, sharding helpers, weight naming."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import GOLD


class FakeTok:
    bos_token_id = 1

    def __call__(self, text):
        class R:
            pass

        r = R()
        r.input_ids = [1] + [3 + (sum(map(ord, wd)) % 90) for wd in text.split()]
        return r

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(i)) for i in row) for row in ids]


def test_tokenizer_image_token_matches_reference():
    from spatialrgpt_amd import tokenizer_image_token

    cases = json.load(open(os.path.join(GOLD, "tokenizer_kat.json")))
    assert len(cases) >= 5
    for c in cases:
        assert tokenizer_image_token(c["prompt"], FakeTok()) == c["ids"], c["prompt"]
        assert tokenizer_image_token(c["prompt"], FakeTok(), lstrip=True) == c["ids_lstrip"], c["prompt"]
    t = tokenizer_image_token("a <image> b", FakeTok(), return_tensors="pt")
    assert t.dtype == torch.long and t.tolist().count(-200) == 1
    with pytest.raises(ValueError):
        tokenizer_image_token("a", FakeTok(), return_tensors="np")


def test_keywords_stopping_criteria():
    from spatialrgpt_amd import KeywordsStoppingCriteria

    tok = FakeTok()
    prompt = torch.zeros((1, 50), dtype=torch.long)
    kw = "stop now"
    crit = KeywordsStoppingCriteria([kw], tok, prompt)
    kid = tok(kw).input_ids[1:]
    assert crit(torch.tensor([[5, 6] + kid]), None) is True
    assert crit(torch.tensor([[5, 6, 7, 8]]), None) is False


def test_process_regions_nearest_and_processor():
    from types import SimpleNamespace

    from spatialrgpt_amd.mm_utils import SrgptImageProcessor, _nearest_resize, process_regions

    m = np.zeros((100, 50), np.uint8)
    m[10:60, 5:20] = 1
    r = _nearest_resize(m, 384, 384)
    assert r.shape == (384, 384) and set(np.unique(r)) == {0, 1}
    assert r[int(10 * 3.84) + 1, int(5 * 7.68) + 1] == 1 and r[0, 0] == 0
    proc = SrgptImageProcessor(size=384)
    out = process_regions([m, m], proc, SimpleNamespace(image_aspect_ratio="resize", image_processor=proc))
    assert out.shape == (2, 384, 384) and out.dtype == torch.float32
    assert set(out.unique().tolist()) == {0.0, 1.0}


def test_config_geometry():
    from spatialrgpt_amd import SrgptConfig

    c = SrgptConfig.vila15_8b()
    assert c.head_dim == 128 and c.grid == 27 and c.vit_layers_run == 26
    assert SrgptConfig.sheared_3b().head_dim == 128 and SrgptConfig.llama2_7b().kv_heads == 32
    assert SrgptConfig.from_dict(c.to_dict()) == c


def test_weight_shapes_cover_checkpoint_names():
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.weights import weight_shapes

    from tests.util import load_tiny

    cfgd, dtype, w, inp, ref = load_tiny("tiny_fp32.npz")
    shapes = weight_shapes(SrgptConfig.from_dict(cfgd))
    for k, s in shapes.items():
        assert k in w and tuple(w[k].shape) == tuple(s), k
    n = sum(int(np.prod(s)) for k, s in weight_shapes(SrgptConfig.vila15_8b()).items() if k.startswith("llm.") and "embed" not in k)
    assert abs(n - 7.50e9) < 0.02e9  # 7.50 B streamed params per token


def test_chunking_matches_reference_semantics():
    from spatialrgpt_amd.dist import get_chunk, split_list

    lst = list(range(10))
    assert split_list(lst, 4) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert get_chunk(lst, 4, 3) == [9] and get_chunk(lst, 8, 7) == []
    assert sum((get_chunk(lst, 3, k) for k in range(3)), []) == lst


def test_missing_extension_fails_loudly(monkeypatch):
    from spatialrgpt_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsrgpt_hip.so")
    with pytest.raises(_lib.SrgptNativeError):
        _lib.load()


def test_no_cpu_fallback():
    from spatialrgpt_amd import ops

    with pytest.raises(RuntimeError):
        ops.rmsnorm(torch.zeros(2, 8), torch.ones(8), 1e-5)


def test_product_does_not_import_oracle():
    import re

    root = os.path.join(os.path.dirname(GOLD), "..", "spatialrgpt_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_pil_bicubic_tables_reproduce_pillow_bit_exactly():
    """the resampling tables the device preprocessing uses (mm_utils.pil_bicubic_tables) are Pillow's own: a numpy
    two-pass integer convolution with them equals Image.resize(BICUBIC) bit for bit (up-, down-scaling, identity axis)."""
    from PIL import Image

    from spatialrgpt_amd.mm_utils import cv2_nearest_index, pil_bicubic_tables

    def resize_np(img, oh, ow):
        H, W, C = img.shape
        b, k = pil_bicubic_tables(W, ow)
        out = np.zeros((H, ow, C), np.uint8)
        for xx in range(ow):
            x0, n = b[xx]
            acc = (img[:, x0:x0 + n, :].astype(np.int64) * k[xx, :n][None, :, None]).sum(1) + (1 << 21)
            out[:, xx, :] = np.clip(acc >> 22, 0, 255)
        b, k = pil_bicubic_tables(H, oh)
        out2 = np.zeros((oh, ow, C), np.uint8)
        for yy in range(oh):
            y0, n = b[yy]
            acc = (out[y0:y0 + n].astype(np.int64) * k[yy, :n][:, None, None]).sum(0) + (1 << 21)
            out2[yy] = np.clip(acc >> 22, 0, 255)
        return out2

    rng = np.random.default_rng(0)
    for H, W, oh, ow in [(240, 320, 96, 96), (50, 77, 96, 96), (96, 200, 96, 96), (300, 96, 96, 96), (97, 31, 64, 80)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(resize_np(img, oh, ow), ref), (H, W, oh, ow)
    # cv2.INTER_NEAREST indices: floor(dst * in/out) clipped, monotone, covering [0, in)
    for n_in, n_out in [(480, 384), (100, 384), (384, 384), (7, 3)]:
        idx = cv2_nearest_index(n_in, n_out)
        assert idx[0] == 0 and idx.max() <= n_in - 1 and np.all(np.diff(idx) >= 0)
        assert np.array_equal(idx, np.minimum((np.arange(n_out) * n_in) // n_out, n_in - 1))  # exact rational floor agrees here


def test_pil_bicubic_tables_random_sizes_match_pillow():
    """property check over random (in, out) sizes: one image row resized with the tables == Pillow (covers extreme up/down
    scaling ratios, sizes 1..700, where the tap count and the clipped bounds at the borders change)."""
    from PIL import Image

    from spatialrgpt_amd.mm_utils import pil_bicubic_tables

    rng = np.random.default_rng(7)
    for _ in range(60):
        w_in, w_out = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        row = rng.integers(0, 256, (1, w_in, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(row).resize((w_out, 1), Image.BICUBIC))
        b, k = pil_bicubic_tables(w_in, w_out)
        out = np.zeros((1, w_out, 3), np.uint8)
        for xx in range(w_out):
            x0, n = b[xx]
            acc = (row[:, x0:x0 + n, :].astype(np.int64) * k[xx, :n][None, :, None]).sum(1) + (1 << 21)
            out[:, xx, :] = np.clip(acc >> 22, 0, 255)
        assert np.array_equal(out, ref), (w_in, w_out)


def test_two_pass_resize_order_follows_the_installed_pillow():
    """Whole images, both passes: the integer two-pass resize the device kernels run (mm_utils._pil_resize_u8_restated on
    pil_bicubic_tables) == Image.resize of the INSTALLED Pillow for random sizes, including images more than 100 times taller than
    wide that shrink vertically -- recent releases resize those rows first (found by tests/test_gpu_fuzz_shapes.py: up to 22 codes
    off with the passes in the usual order), which mm_utils.pil_resizes_tall_images_vertically_first asks the library itself."""
    from PIL import Image

    from spatialrgpt_amd.mm_utils import _pil_resize_u8_restated, pil_resizes_tall_images_vertically_first

    rows_first = pil_resizes_tall_images_vertically_first()
    rng = np.random.default_rng(11)
    shapes = [(1078, 4), (500, 3), (4, 1078), (600, 5), (600, 8), (379, 3), (400, 4), (401, 4), (202, 2), (150, 1)]
    shapes += [(int(rng.integers(1, 500)), int(rng.integers(1, 500))) for _ in range(12)]
    for h, w in shapes:
        oh, ow = (int(rng.integers(1, 400)), int(rng.integers(1, 400))) if rng.random() < 0.5 else (96, 96)
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(arr).resize((ow, oh), Image.BICUBIC))
        vf = rows_first and h > 100 * w and oh < h
        assert np.array_equal(_pil_resize_u8_restated(arr, oh, ow, vertical_first=vf), ref), ((h, w), (oh, ow), vf)


def test_bench_presets_map_to_baseline_configs(monkeypatch):
    """bench.py's presets / flags name the BASELINE.json configuration they measure (no GPU needed: argument plumbing only)."""
    import importlib
    import sys as _sys

    bench = importlib.import_module("bench")
    from spatialrgpt_amd.config import SrgptConfig

    def wl(argv):
        monkeypatch.setattr(_sys, "argv", ["bench.py"] + argv)
        a = bench.parse()
        cfg = bench.make_cfg(a.model)
        return a, bench.workload_name(a, cfg, a.prompt_len - 1 + 196)

    a, name = wl([])
    assert a.gpus == 1 and a.batch == 1 and "BASELINE configs[1]" in name and "T=259" in name
    a, name = wl(["--preset", "config2"])
    assert a.batch == 4 and "configs[2]" in name
    a, name = wl(["--preset", "config3"])
    assert a.model == "llama2_7b" and a.regions == 16 and a.prompt_len == 512 and "configs[3]" in name and "T=707" in name
    a, name = wl(["--preset", "config4"])  # "fp8 weights on CDNA4 fp8 MFMA" = the W8A8-prefill mode README / DESIGN quote
    assert a.weights == "fp8_w8a8" and a.batch == 8 and "configs[4]" in name and "fp8 matrix pipe" in name
    a, name = wl(["--preset", "config4_w8a16"])
    assert a.weights == "fp8" and a.batch == 8 and "configs[4]" in name and "fp8 matrix pipe" not in name
    a, name = wl(["--batch", "3"])
    assert "non-BASELINE" in name
    assert isinstance(bench.make_cfg("vila15_8b_clip336"), SrgptConfig)


def test_fp8_quantisation_rules_agree_between_host_and_oracle():
    """The power-of-two row scale + round-to-nearest-even e4m3fn rule exists three times: spatialrgpt_amd.ops.quantize_fp8_rows
    (weights at load, the reference for the device quantiser srgpt_quant_rows_e4m3 in the GPU suite), the oracle's
    fp8_dequantised_weights (log2 / ceil formulation) and its fp8_rowwise_fake_quant (the W8A8 restatement).  They must give the
    same dequantised values on rows that sit exactly on a scale boundary (max = 448 * 2^k), one ulp either side, zeros, and
    random rows; the default llama_forward (act_quant=None) stays the reference's arithmetic."""
    import inspect
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd import ops

    g = torch.Generator().manual_seed(9)
    x = torch.randn((40, 256), generator=g) * torch.exp2(torch.randint(-8, 8, (40, 1), generator=g).float())
    x[0] = 0
    x[1, 3] = 448.0 * 2.0 ** -5       # exactly on the boundary: codes reach 448
    x[2, 3] = 452.0 * 2.0 ** 3        # the next bf16 value above a boundary: the scale doubles
    x[3, 3] = 446.0 * 2.0 ** 3
    x[1:4] = x[1:4].clamp(-1e9, 1e9)
    x[1, :3] = 0.01
    xb = x.to(torch.bfloat16)
    for r, peak in ((1, 448.0 * 2.0 ** -5), (2, 452.0 * 2.0 ** 3), (3, 446.0 * 2.0 ** 3)):
        xb[r] = (xb[r].float().clamp(-peak, peak)).to(torch.bfloat16)
    q8, sc, deq = ops.quantize_fp8_rows(xb)
    assert float(q8.view(torch.float8_e4m3fn).float().abs().max()) <= 448.0
    assert torch.equal(torch.log2(sc), torch.log2(sc).round()), "scales are powers of two"
    amax = xb.float().abs().amax(dim=1)
    nz = amax > 0
    assert bool(((amax / sc)[nz] <= 448.0).all()) and bool(((amax / sc)[nz] > 224.0).all()), "the smallest admissible power of two"
    fake = so.fp8_rowwise_fake_quant(xb)
    assert torch.equal(fake, deq), "oracle activation fake-quant == host quantise / dequantise"
    wq = so.fp8_dequantised_weights({"llm.model.layers.0.mlp.down_proj.weight": xb})["llm.model.layers.0.mlp.down_proj.weight"]
    assert torch.equal(wq, deq), "oracle weight rule (log2 / ceil form) == host rule (frexp form)"
    # a 3-d activation tensor quantises per token (last dimension)
    x3 = xb.view(5, 8, 256)
    assert torch.equal(so.fp8_rowwise_fake_quant(x3).view(40, 256), deq)
    assert inspect.signature(so.llama_forward).parameters["act_quant"].default is None


def test_rope_tables_follow_the_vendored_linear_scaling_class_bit_for_bit():
    """weights.rope_tables (the tables srgpt_rope_kv_append / srgpt_decode_attention index) == cos / sin of the reference's vendored
    LlamaLinearScalingRotaryEmbedding (modeling_llama.py:133-140) under rope_scaling {linear, 3.0}, fp32 and bf16, positions
    0 .. 63 of a model with max_position_embeddings 32 (tests/golden/vendored_llama_kat.npz; context_length_extension,
    language_model/builder.py:31-38)."""
    import json
    import os

    import numpy as np

    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.weights import rope_tables
    from tests.util import GOLD

    z = np.load(os.path.join(GOLD, "vendored_llama_kat.npz"))
    geo = json.loads(bytes(z["geo_json"]).decode())
    cfg = SrgptConfig(hidden=geo["hidden_size"], heads=geo["num_attention_heads"], kv_heads=geo["num_key_value_heads"],
                      rope_theta=geo["rope_theta"], rope_factor=geo["rope_factor"])
    c32, s32 = rope_tables(cfg, 64, torch.float32, "cpu")
    assert torch.equal(c32, torch.from_numpy(z["rope.cos_f32"])) and torch.equal(s32, torch.from_numpy(z["rope.sin_f32"]))
    c16, s16 = rope_tables(cfg, 64, torch.bfloat16, "cpu")
    assert torch.equal(c16.float(), torch.from_numpy(z["rope.cos_bf16"])) and torch.equal(s16.float(), torch.from_numpy(z["rope.sin_bf16"]))


def test_cv2_nearest_tables_follow_opencv_resizenn():
    """`cv2_nearest_index` (the tables of the device mask resize AND of the host process_regions) == OpenCV's published resizeNN
    arithmetic, min(cvFloor(x * (1. / ((double)out / in))), in - 1), derived with exact rationals by oracle/make_cv2_nearest_kat.py
    (cv2 is not installed here: stated in INTEGRATION.md).  6 of the 20 size pairs differ from floor(x * in / out) -- the formula
    round 3 used on the host."""
    from types import SimpleNamespace

    from spatialrgpt_amd.mm_utils import SrgptImageProcessor, cv2_nearest_index, process_regions

    kat = json.load(open(os.path.join(GOLD, "cv2_nearest_kat.json")))
    differing = 0
    for c in kat["cases"]:
        idx = cv2_nearest_index(c["in"], c["out"])
        assert idx.tolist() == c["index"], (c["in"], c["out"])
        differing += bool(c["differs_from_floor_x_in_over_out_at"])
    assert differing >= 5
    # the host process_regions gathers through the same tables: a ramp mask reads back the KAT's indices
    c = next(c for c in kat["cases"] if c["in"] == 72 and c["out"] == 224)
    proc = SrgptImageProcessor(size=224)
    ramp = np.tile(np.arange(72, dtype=np.uint8)[None, :], (72, 1))
    out = process_regions([ramp, ramp.T.copy()], proc, SimpleNamespace(image_aspect_ratio="resize", image_processor=proc))
    assert out.shape == (2, 224, 224)
    assert out[0, 0].to(torch.int64).tolist() == c["index"] and out[1, :, 0].to(torch.int64).tolist() == c["index"]


def test_pad_mode_masks_follow_pillow_and_the_hf_processor_bit_for_bit():
    """process_regions with image_aspect_ratio == "pad" (mm_utils.py:505-531): pad_to_square, then the processor's resize of the
    [1, side, side] uint8 array -- HF resizes it as an 8-bit PIL image (bicubic, rounded and clipped).  The host path == Pillow
    == transformers' own SiglipImageProcessor on the same array (the device path is pinned to the host path by
    tests/test_gpu_edge_cases.py)."""
    import copy
    from types import SimpleNamespace

    from PIL import Image

    from spatialrgpt_amd.mm_utils import SrgptImageProcessor, process_regions

    rng = np.random.default_rng(0)
    proc = SrgptImageProcessor(size=384)
    for (h, w), hi in [((333, 500), 255), ((640, 480), 1), ((97, 97), 255)]:
        m = (rng.random((h, w)) > 0.6).astype(np.uint8) * hi
        out = process_regions([m], proc, SimpleNamespace(image_aspect_ratio="pad", image_processor=proc))
        side = max(h, w)
        p = np.zeros((side, side), np.uint8)
        p[(side - h) // 2:(side - h) // 2 + h, (side - w) // 2:(side - w) // 2 + w] = m
        pil = np.asarray(Image.fromarray(p, "L").resize((384, 384), Image.BICUBIC)).astype(np.float32)
        assert out.shape == (1, 384, 384) and out.dtype == torch.float32
        assert torch.equal(out[0], torch.from_numpy(pil))
        try:
            from transformers import SiglipImageProcessor
            hf = SiglipImageProcessor(size={"height": 384, "width": 384}, resample=3, do_rescale=True, rescale_factor=1 / 255,
                                      do_normalize=True, image_mean=[0.5] * 3, image_std=[0.5] * 3)
        except Exception:  # pragma: no cover
            continue
        mp = copy.deepcopy(hf)  # process_regions' own edits of the processor (mm_utils.py:479-482)
        mp.do_normalize, mp.do_convert_rgb, mp.rescale_factor = False, False, 1.0
        ref = mp.preprocess(p[None, ...], return_tensors="pt")["pixel_values"][0]
        assert torch.equal(out, torch.as_tensor(ref).float())


def test_the_gpu_tests_host_splice_loop_equals_the_reference_row_source_maps():
    """tests/test_gpu_splice.py::splice_reference (the checker of the randomised device-splice cases) against the reference's own
    row-source maps (tests/golden/splice_kat.npz): the checker is pinned, not only what it checks."""
    import types
    import warnings

    import pytest
    import torch

    from tests.test_gpu_splice import splice_reference
    from tests.util import splice_kat_cases, splice_kat_tables
    for n, c in enumerate(splice_kat_cases()):
        cfg = types.SimpleNamespace(mask_token_id=c["mask_token_id"], depth_token_id=c["depth_token_id"], enable_region=True,
                                    enable_depth=True, tokenizer_model_max_length=c["max_length"], padding_side=c["padding_side"])
        embed, feats, me, de, expect = splice_kat_tables(c, 8, seed=n)
        args = (cfg, c["vocab"], embed, c["input_ids"], c["attention_mask"], feats, me, de, c["have_depths"], c["labels"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if c["raises"]:
                with pytest.raises(RuntimeError, match="shape mismatch"):
                    splice_reference(*args)
                continue
            out, am, lens, new_labels = splice_reference(*args)
        assert torch.equal(out, expect()), c["name"]
        if c["attention_mask_out"] is not None:
            assert torch.equal(am, c["attention_mask_out"]), c["name"]
        if c["labels"] is not None:
            assert torch.equal(new_labels, c["new_labels"]), c["name"]
