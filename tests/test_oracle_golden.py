"""CPU: pins oracle/srgpt_oracle.py against the golden vectors minted from the REAL reference
(oracle/make_golden.py).  If this fails the oracle is not a faithful restatement and no GPU parity claim holds."""
import numpy as np
import pytest
import torch

from oracle import srgpt_oracle as so
from tests.util import assert_close, load_kat, load_tiny


def _cfg(d):
    names = so.SrgptConfig.__dataclass_fields__
    return so.SrgptConfig(**{k: v for k, v in d.items() if k in names})


@pytest.mark.parametrize("name,atol,rtol", [("tiny_fp32.npz", 1e-5, 1e-5), ("tiny_bf16.npz", 0.0, 2e-2),
                                            ("tiny_clip_fp32.npz", 1e-5, 1e-5)])
def test_oracle_reproduces_reference_stages(name, atol, rtol):
    cfgd, dtype, w, inp, ref = load_tiny(name)
    cfg = _cfg(cfgd)
    ids, st = so.generate(w, cfg, inp["input_ids"], inp["images"], inp["depths"], inp["masks"], max_new_tokens=12,
                          return_stages=True, model_dtype=dtype)
    for k in ("tower_features", "depth_features", "lres", "image_features", "inputs_embeds"):
        assert_close(st[k], ref[k], atol, rtol, k)
    idx = ref["hres_rows_idx"].long()
    assert_close(st["hres"][:, idx], ref["hres_rows"], atol, rtol, "hres rows")
    assert_close(torch.stack(st["mask_embeds"]), ref["mask_embeds"], atol, rtol, "mask_embeds")
    assert_close(torch.stack(st["depth_embeds"]), ref["depth_embeds"], atol, rtol, "depth_embeds")
    assert_close(st["prefill_logits"], ref["prefill_logits"], 1e-4 if dtype == torch.float32 else 2e-2, rtol, "prefill_logits")
    if dtype == torch.float32:
        assert torch.equal(ids, ref["new_ids"]), (ids, ref["new_ids"])  # greedy ids bit-exact
    assert st["inputs_embeds"].shape[1] == inp["input_ids"].shape[1] - 1 + 196  # T = P - 1 + 196


def test_oracle_region_kats():
    z = load_kat()
    for tag, fw in [("rgb108", 108), ("depth27", 27), ("soft336_to_108", 108), ("soft336_to_96", 96), ("up56_to_108", 108)]:
        feat = torch.from_numpy(z[f"pool.{tag}.feat_q64"].astype(np.float32) / 64)
        masks = torch.from_numpy(z[f"pool.{tag}.masks_q16"].astype(np.float32) / 16)
        out = so.mask_pooling(feat[None], [masks])[0]
        assert_close(out, torch.from_numpy(z[f"pool.{tag}.out"]), 1e-6, 0, tag)
    # closed forms: all-ones mask = mean of features; empty mask = zeros (denorm = 1e-8)
    feat = torch.from_numpy(z["pool.rgb108.feat_q64"].astype(np.float32) / 64)
    out = torch.from_numpy(z["pool.rgb108.out"])
    assert_close(out[3], feat.mean(0), 1e-5, 0, "all-ones mask == feature mean")
    assert float(out[4].abs().max()) == 0.0


def test_oracle_s2d_and_refinement():
    z = load_kat()
    x = torch.from_numpy(z["s2d.in"])
    n, L, c = x.shape
    assert torch.equal(so.flat_square(x.reshape(n, 27, 27, c)).reshape(n, -1, 4 * c), torch.from_numpy(z["s2d.out"]))
    for tag in ("27", "24"):
        w = {so.RE + "feature_refinement_module." + k[len(f"refine{tag}.w."):]: torch.from_numpy(z[k])
             for k in z.files if k.startswith(f"refine{tag}.w.")}
        h, l = so.feature_refinement(w, torch.from_numpy(z[f"refine{tag}.in"]))
        idx = torch.from_numpy(z[f"refine{tag}.hres_rows_idx"]).long()
        assert_close(h[:, idx], torch.from_numpy(z[f"refine{tag}.hres_rows"]), 1e-5, 0, "hres")
        assert_close(l, torch.from_numpy(z[f"refine{tag}.lres"]), 1e-5, 0, "lres")


def test_oracle_kv_cache_equals_full_forward():
    """size-independent property: incremental decode == teacher-forced full forward."""
    cfgd, dtype, w, inp, ref = load_tiny("tiny_fp32.npz")
    cfg = _cfg(cfgd)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 9, cfg.hidden), generator=g)
    pos = torch.arange(9)[None]
    full = so.llama_forward(w, cfg, x, pos, so.KVCache(cfg.layers))
    kv = so.KVCache(cfg.layers)
    a = so.llama_forward(w, cfg, x[:, :6], pos[:, :6], kv)
    outs = [a]
    for t in range(6, 9):
        outs.append(so.llama_forward(w, cfg, x[:, t:t + 1], pos[:, t:t + 1], kv))
    assert_close(torch.cat(outs, 1), full, 2e-5, 0, "incremental vs full")


def test_oracle_fp8_weight_quantisation_properties():
    """fp8_dequantised_weights (BASELINE config 5 semantics): only LLM Linear weights change; row scales are powers of two
    with max|row|/scale in (224, 448]; the map is idempotent; the error is bounded by half an e4m3 ulp of the row maximum."""
    import torch

    from oracle import srgpt_oracle as so

    cfg = so.SrgptConfig(vit_hidden=32, vit_inter=64, vit_layers=2, vit_heads=2, image_size=28, patch_size=14, hidden=64, inter=128,
                         layers=2, heads=4, kv_heads=2, vocab=128, mask_token_id=120, depth_token_id=121)
    w = so.synth_weights(cfg, seed=5, dtype=torch.bfloat16)
    q = so.fp8_dequantised_weights(w)
    changed = {k for k in w if not torch.equal(w[k], q[k])}
    assert changed and all(k.startswith("llm.") and (k.endswith("_proj.weight") or k == "llm.lm_head.weight") for k in changed)
    assert torch.equal(q["llm.model.embed_tokens.weight"], w["llm.model.embed_tokens.weight"])
    again = so.fp8_dequantised_weights(q)
    for k in changed:
        assert torch.equal(again[k], q[k]), f"{k}: quantisation is not idempotent"
        a, b = w[k].float(), q[k].float()
        amax = a.abs().amax(dim=1, keepdim=True)
        assert float(((a - b).abs() / amax).max()) <= 2.0 ** -4 + 2.0 ** -8  # 3 mantissa bits (+ the bf16 store)
        # codes recovered from the dequantised values sit on a power-of-two grid: max|row| / scale in (224, 448]
        sc = torch.ldexp(torch.ones(a.shape[0]), torch.ceil(torch.log2(b.abs().amax(dim=1).double() / 448.0)).to(torch.int32))
        r = b.abs().amax(dim=1) / sc
        assert bool(((r > 223.9) & (r <= 448.0)).all())


def test_oracle_labels_and_loss_match_reference():
    """teacher-forced forward with labels (llava_llama.py:100-192): the oracle's spliced labels / attention mask equal the
    reference's (tests/golden/labels_kat.npz, minted from the real reference) and its causal-LM loss matches to 2e-5."""
    import os

    from tests.util import GOLD

    cfgd, dtype, w, inp, ref = load_tiny("tiny_fp32.npz")
    cfg = _cfg(cfgd)
    z = np.load(os.path.join(GOLD, "labels_kat.npz"))
    ids, am, labels = (torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "labels"))
    im2, dp2 = torch.cat([inp["images"]] * 2, 0), torch.cat([inp["depths"]] * 2, 0)
    mk2 = [inp["masks"][0], inp["masks"][0]]
    image_features, mask_embeds, depth_embeds, _ = so.encode_visual(w, cfg, im2, dp2, mk2)
    emb, am_out, pid, new_labels = so.splice(w, cfg, ids, am, image_features, mask_embeds, depth_embeds, have_depths=True,
                                             labels=labels)
    assert torch.equal(new_labels, torch.from_numpy(z["new_labels"]))
    assert torch.equal(am_out, torch.from_numpy(z["attention_mask_out"]))
    logits = so.llama_forward(w, cfg, emb, pid, so.KVCache(cfg.layers), key_padding_mask=am_out.bool())
    loss = float(so.causal_lm_loss(logits, new_labels))
    assert abs(loss - float(z["loss"])) <= 2e-5 * max(1.0, abs(float(z["loss"])))
    last = torch.stack([logits[b, int(am_out[b].sum()) - 1] for b in range(2)])
    assert_close(last, torch.from_numpy(z["logits_valid_last"]), 1e-4, 0, "last valid logits")


def test_oracle_splice_equals_the_reference_row_source_maps():
    """The splice restatement against tests/golden/splice_kat.npz -- the reference's own prepare_inputs_labels_for_multimodal
    (llava_arch.py:333-650) run on index-coded rows: several images per prompt, text-only rows, None mask entries, surplus /
    missing region embeddings, depths=None, left padding, truncation, attention masks with holes, list / 5-D images, labels.
    Integer work: rows, attention mask and labels compare exactly; the error case raises."""
    import warnings

    from tests.util import splice_kat_cases, splice_kat_tables
    n = 0
    for c in splice_kat_cases():
        cfg = so.SrgptConfig(hidden=16, vocab=c["vocab"], mask_token_id=c["mask_token_id"], depth_token_id=c["depth_token_id"],
                             padding_side=c["padding_side"], tokenizer_model_max_length=c["max_length"])
        embed, feats, me, de, expect = splice_kat_tables(c, 16, seed=n)
        w = {so.LM + "model.embed_tokens.weight": embed}
        args = (w, cfg, c["input_ids"], c["attention_mask"], feats, me, de)
        n += 1
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if c["raises"]:
                with pytest.raises(RuntimeError):
                    so.splice(*args, have_depths=c["have_depths"], labels=c["labels"])
                continue
            o = so.splice(*args, have_depths=c["have_depths"], labels=c["labels"])
        assert torch.equal(o[0], expect()), c["name"]
        if c["attention_mask_out"] is None:
            assert o[1] is None, c["name"]
        else:
            assert torch.equal(o[1].bool(), c["attention_mask_out"]), c["name"]
        if c["labels"] is not None:
            assert torch.equal(o[3], c["new_labels"]), c["name"]
    assert n >= 14


def _vendored():
    import json
    import os

    from tests.util import GOLD
    z = np.load(os.path.join(GOLD, "vendored_llama_kat.npz"))
    geo = json.loads(bytes(z["geo_json"]).decode())
    cfg = so.SrgptConfig(hidden=geo["hidden_size"], inter=geo["intermediate_size"], layers=geo["num_hidden_layers"],
                         heads=geo["num_attention_heads"], kv_heads=geo["num_key_value_heads"], vocab=geo["vocab_size"],
                         rms_eps=geo["rms_norm_eps"], rope_theta=geo["rope_theta"], rope_factor=geo["rope_factor"],
                         mask_token_id=126, depth_token_id=127)
    w = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    return z, cfg, w


def test_oracle_reproduces_the_vendored_llama_with_linear_rope_scaling():
    """tests/golden/vendored_llama_kat.npz = the reference's VENDORED modeling_llama.py (LlamaFlashAttention2 hard-wired,
    LlamaLinearScalingRotaryEmbedding, rope_scaling {linear, 3.0}) run on CPU with oracle/flash_attn_cpu.py standing in for the
    flash-attn extension (oracle/make_golden.py vendored).  fp32: the oracle's logits are BIT-identical at every prompt position,
    on 10 greedy steps over the cache at positions beyond max_position_embeddings, and on the right-padded ragged batch."""
    z, cfg, w = _vendored()
    ids = torch.from_numpy(z["single.ids"])
    T = ids.shape[1]
    kv = so.KVCache(cfg.layers)
    emb = torch.nn.functional.embedding(ids, w["llm.model.embed_tokens.weight"])
    logits = so.llama_forward(w, cfg, emb, torch.arange(T)[None], kv)
    assert torch.equal(logits, torch.from_numpy(z["single.prefill_logits"]))
    ref_ids, ref_steps = torch.from_numpy(z["single.new_ids"]), torch.from_numpy(z["single.step_logits"])
    nxt = logits[:, -1].argmax(-1)
    for t in range(ref_ids.shape[1]):
        assert int(nxt) == int(ref_ids[0, t])
        e = torch.nn.functional.embedding(nxt[:, None], w["llm.model.embed_tokens.weight"])
        lg = so.llama_forward(w, cfg, e, torch.tensor([[T + t]]), kv, last_only=True)[:, -1]
        assert torch.equal(lg, ref_steps[:, t]), f"step {t}"
        nxt = lg.argmax(-1)
    ids2, lens = torch.from_numpy(z["ragged.ids"]), z["ragged.lens"].tolist()
    am = torch.zeros(ids2.shape, dtype=torch.bool)
    pos = torch.zeros(ids2.shape, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, :n] = True
        pos[b, :n] = torch.arange(n)
    got = so.llama_forward(w, cfg, torch.nn.functional.embedding(ids2, w["llm.model.embed_tokens.weight"]), pos,
                           so.KVCache(cfg.layers), key_padding_mask=am)
    for b, n in enumerate(lens):
        assert torch.equal(got[b, :n], torch.from_numpy(z["ragged.logits"])[b, :n]), f"ragged row {b}"
    # with the scaling applied the transformers >= 4.45 way (inv_freq / factor) the fixture is NOT reproduced bit for bit:
    # the fixture tells the two orders apart
    c_ok, _ = so.rope_cos_sin(cfg, torch.arange(64)[None], torch.float32)
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d)) / cfg.rope_factor
    c_other = (torch.arange(64).float()[:, None] * inv[None]).cos()
    assert torch.equal(c_ok[0, :, :d // 2], torch.from_numpy(z["rope.cos_f32"])) and not torch.equal(c_other, torch.from_numpy(z["rope.cos_f32"]))
