"""GPU: parity under OUTLIER STATISTICS (VERDICT r4 missing #3).  Every other parity number of the suite is measured on N(0, 0.02)
or peaked-margin weights; trained checkpoints are heavy-tailed (tests.util.make_heavy_tailed says how and plants it).  Two layers:

  kernels   the product kernels on heavy-tailed operands against fp64 torch on the same bf16-valued inputs, with a bound derived from
            the arithmetic, element by element:  |got - exact| <= 2^-8 |exact|  (the bf16 rounding of the output, doubled)
                                                          + 2^-19 sum_k |a_k b_k|  (fp32 accumulation, any order: K 2^-24 worst case at K <= 32)
            plus, where an RMSNorm is fused in front, one bf16 ulp on the largest normalised term (the statistic's summation order may
            flip a rounding).  Flash / decode attention with logits of +-60 against fp64 softmax attention (2e-2 of the output range).
            The e4m3 quantisers exactly (power-of-two scales: no rounding freedom).
  model     a 4-layer TRUE-WIDTH request (Llama-3-8B layer geometry behind 3 SigLIP-so400m-width layers, 16k vocabulary) with the
            planted statistics, bf16 / fp8 W8A16 / fp8 W8A8: every stage tensor finite, prefill and teacher-forced decode logits
            against the oracle with the NOISE-FLOOR bars of tests/test_gpu_fulldepth_parity.py (floor = oracle_bf16 - oracle_fp32 on
            the same weights): engine vs bf16 oracle <= 2 x floor (max and rms), engine vs fp32 oracle rms <= 1.25 x floor rms.
            W8A8 is compared with ITS oracle (per-token e4m3 activations in the prefill) under the same bars, and its distance from
            the un-quantised fp32 model is RECORDED: an outlier channel sets a token's scale and flushes the rest of the row, which is
            a property of the format, reported next to the configs[4] number (README), not hidden by a tolerance.
Measured numbers -> gpurun_out/r05_outlier_parity.json (committed under profiles/)."""
import json
import os
import sys

import pytest
import torch

from tests.util import logit_parity_report, make_heavy_tailed, teacher_forced_decode_logits

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPORT = os.path.join(ROOT, "gpurun_out", "r05_outlier_parity.json")


# "spec": the statistics VERDICT r4 #3 lists (gains log-uniform in [0.1, 50], massive channels x 300, rows x 30, logits x 16) -- planted
# into EVERY norm and projection of an untrained model they compound: the bf16 oracle itself then sits 13 % rms (> 100 % max) of the
# logit range away from its own fp32 run, so the floor-relative bars hold but discriminate little.  "moderate" keeps bf16 arithmetic
# meaningful (a floor of a few per cent), which is the regime a trained checkpoint lives in, and is where the bars bite.
LEVELS = {"spec": dict(gains=(0.1, 50.0), n_massive=6, massive=300.0, hot_rows=3, hot_row_gain=30.0, attn_gain=4.0),
          "moderate": dict(gains=(0.25, 8.0), n_massive=4, massive=100.0, hot_rows=3, hot_row_gain=8.0, attn_gain=2.0)}


def _record(section, key, value):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
    data.setdefault(section, {})[key] = value
    json.dump(data, open(REPORT, "w"), indent=1)


def _heavy(shape, seed, massive_cols=6, massive=300.0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * scale
    cols = torch.randperm(shape[-1], generator=g)[:massive_cols]
    x[..., cols] *= massive
    return x.to(dtype)


def _gains(n, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    import math
    return torch.exp(torch.rand((n,), generator=g) * (math.log(50.0) - math.log(0.1)) + math.log(0.1)).to(dtype)


def _hot_rows(w, seed, n=3, gain=30.0):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randperm(w.shape[0], generator=g)[:n]
    w = w.clone()
    w[rows] = (w[rows].float() * gain).to(w.dtype)
    return w


def _bound_check(got, a64, b64, what, extra=None):
    """|got - a b^T| <= 2^-8 |exact| + 2^-19 sum |a||b| (+ extra), element by element; records the worst ratio err / bound"""
    exact = a64 @ b64.T
    bound = exact.abs() * 2.0 ** -8 + (a64.abs() @ b64.abs().T) * 2.0 ** -19
    if extra is not None:
        bound = bound + extra
    got = got.double().cpu()
    assert bool(torch.isfinite(got).all()), f"{what}: non-finite output"
    ratio = float(((got - exact).abs() / bound.clamp_min(1e-300)).max())
    _record("kernels", what, {"worst_err_over_bound": ratio, "exact_absmax": float(exact.abs().max()),
                              "rel_err_of_max": float((got - exact).abs().max() / exact.abs().max())})
    assert ratio <= 1.0, f"{what}: error {ratio:.3f} x the arithmetic bound"


@pytest.mark.parametrize("M,N,K", [(259, 512, 4096), (259, 256, 14336), (64, 1000, 1152), (1458, 384, 1152)])
def test_gemm_on_heavy_tailed_operands(M, N, K):
    from spatialrgpt_amd import ops

    a = _heavy((M, K), 1)
    w = _hot_rows(_heavy((N, K), 2, massive_cols=0, scale=0.03), 3)
    _bound_check(ops.gemm(a.to(DEV), w.to(DEV)), a.double(), w.double(), f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("B", [1, 2, 8, 16])
@pytest.mark.parametrize("fp8", [False, True])
def test_decode_products_on_heavy_tailed_operands(B, fp8):
    """the decode GEMV / batched MFMA product with the fused RMSNorm (gains log-uniform in [0.1, 50], 6 massive channels x 300),
    hot weight rows x 30, bf16 and fp8 weights; with the row-statistics hand-off where the batch size has it"""
    from spatialrgpt_amd import ops

    K, N = 4096, 1536
    x = _heavy((B, K), 11, scale=0.05)
    g = _gains(K, 12)
    w = _hot_rows(_heavy((N, K), 13, massive_cols=0, scale=0.03), 14)
    xd, gd, wd = x.to(DEV), g.to(DEV), w.to(DEV)
    if fp8:
        q8, sc, _ = ops.quantize_fp8_rows(wd)
        w64 = (q8.cpu().view(torch.float8_e4m3fn).double() * sc.cpu().double()[:, None])
        got = ops.gemv_w8(xd, q8, sc, norm_w=gd, eps=1e-5)
        plain = ops.gemv_w8(xd, q8, sc)
    else:
        w64 = w.double()
        got = ops.gemv(xd, wd, norm_w=gd, eps=1e-5)
        plain = ops.gemv(xd, wd)
    _bound_check(plain, x.double(), w64, f"decode product B={B} fp8={fp8}")
    xf = x.float()
    xn = (g.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16).float()).to(torch.bfloat16)
    flip = (xn.double().abs()[:, None, :] * w64.abs()[None, :, :]).amax(-1) * 2.0 ** -7   # one bf16 ulp of the largest term, twice
    _bound_check(got, xn.double(), w64, f"RMSNorm + decode product B={B} fp8={fp8}", extra=flip)
    if ops.gemv_rowss_supported(B, fp8):
        kw = dict(w8=q8, wscale=sc) if fp8 else dict(w=wd)
        table = torch.zeros((B, 512), device=DEV)
        table[:, 0] = xd.float().pow(2).sum(-1)
        _bound_check(ops.gemv_rowss(xd, norm_w=gd, eps=1e-5, rowss_in=table, **kw), xn.double(), w64,
                     f"RMSNorm (published statistics) + decode product B={B} fp8={fp8}", extra=flip)


@pytest.mark.parametrize("T,Hq,Hkv,D,causal", [(259, 32, 8, 128, True), (729, 16, 16, 72, False), (2048, 8, 8, 128, True)])
def test_flash_attention_with_logits_of_60(T, Hq, Hkv, D, causal):
    from spatialrgpt_amd import ops

    g = torch.Generator().manual_seed(5)
    q = torch.randn((1, T, Hq, D), generator=g)
    k = torch.randn((1, T, Hkv, D), generator=g)
    v = _heavy((1, T, Hkv, D), 6, massive_cols=3, massive=100.0).float()
    # q, k ~ N(0, 20): q.k / sqrt(D) has standard deviation 20, its extremes over T x T pairs pass +-60
    q, k = (q * 20.0 ** 0.5).bfloat16(), (k * 20.0 ** 0.5).bfloat16()
    v = v.bfloat16()
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal)
    rep = Hq // Hkv
    qq, kk, vv = q.double().transpose(1, 2), k.double().transpose(1, 2).repeat_interleave(rep, 1), v.double().transpose(1, 2).repeat_interleave(rep, 1)
    s = qq @ kk.transpose(-1, -2) / D ** 0.5
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), float("-inf"))
    ref = (s.softmax(-1) @ vv).transpose(1, 2)
    got = out.double().cpu()
    assert bool(torch.isfinite(got).all())
    err = float((got - ref).abs().max() / ref.abs().max())
    _record("kernels", f"flash T={T} D={D} causal={causal}", {"logit_absmax": float(s[torch.isfinite(s)].abs().max()), "max_err_over_range": err})
    assert float(s[torch.isfinite(s)].abs().max()) >= 60.0 and err <= 2e-2, err


def test_decode_attention_with_logits_of_60():
    from spatialrgpt_amd import ops
    from spatialrgpt_amd.config import SrgptConfig as PC
    from spatialrgpt_amd.weights import rope_tables

    B, Hq, Hkv, D, P, max_pos = 2, 32, 8, 128, 700, 1024
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(8)
    QW = (Hq + 2 * Hkv) * D
    allq = torch.randn((B, P + 1, QW), generator=g)
    allq[..., :(Hq + Hkv) * D] *= 20.0 ** 0.5   # q and k parts ~ N(0, 20): logits of standard deviation 20 (RoPE keeps norms)
    allq = allq.to(dtype).to(DEV)
    cos_t, sin_t = rope_tables(PC(hidden=Hq * D, heads=Hq, kv_heads=Hkv, rope_theta=500000.0), max_pos, dtype, DEV)
    kc = torch.zeros((B, Hkv, max_pos, D), device=DEV, dtype=dtype)
    vc = torch.zeros_like(kc)
    ops.rope_kv_append(allq[:, :P].contiguous(), kc, vc, cos_t, sin_t, B, P, Hq, Hkv, D)
    pos = torch.full((B,), P, device=DEV, dtype=torch.int32)
    out = ops.decode_attention(allq[:, P].contiguous(), kc, vc, pos, cos_t, sin_t, Hq, Hkv, D)
    # reference: the roped q of the new token against the cache the kernel just completed (rope itself is pinned elsewhere)
    from oracle import srgpt_oracle as so
    qn = allq[:, P, :Hq * D].reshape(B, 1, Hq, D).transpose(1, 2).cpu()
    kn = allq[:, P, Hq * D:(Hq + Hkv) * D].reshape(B, 1, Hkv, D).transpose(1, 2).cpu()
    pid = torch.full((B, 1), P)
    cos, sin = so.rope_cos_sin(so.SrgptConfig(hidden=Hq * D, heads=Hq, kv_heads=Hkv, rope_theta=500000.0), pid, dtype)
    qr, _ = so.apply_rope(qn, kn, cos, sin)
    K_, V_ = kc[:, :, :P + 1].double().cpu(), vc[:, :, :P + 1].double().cpu()
    rep = Hq // Hkv
    s = (qr.double() @ K_.repeat_interleave(rep, 1).transpose(-1, -2)) / D ** 0.5
    ref = (s.softmax(-1) @ V_.repeat_interleave(rep, 1)).transpose(1, 2).reshape(B, Hq * D)
    got = out.double().cpu().reshape(B, Hq * D)
    assert bool(torch.isfinite(got).all())
    err = float((got - ref).abs().max() / ref.abs().max())
    _record("kernels", "decode attention P=700", {"logit_absmax": float(s.abs().max()), "max_err_over_range": err})
    # logits of +-84: ONE bf16 ulp on the new token's roped q (the kernel ropes in fp32 and rounds once, the reference's rotary
    # arithmetic rounds twice) moves a logit by up to 2^-9 x 84 = 0.16, i.e. a probability by 18 % -- the bar carries that term
    smax = float(s.abs().max())
    assert smax >= 60.0 and err <= 2e-2 + 2.0 ** -9 * smax / 4, err


def test_e4m3_row_quantiser_on_heavy_tailed_rows_is_exact():
    """per-token e4m3 codes and power-of-two scales of rows with massive channels == the oracle's fake quantiser, bit for bit (the
    flush of the small entries of such a row is the format's, identically on both sides)"""
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd import ops

    x = _heavy((37, 4096), 21, scale=0.05)
    q, sc = ops.quant_rows_e4m3(x.to(DEV))
    deq = (q.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None]).to(torch.bfloat16)
    ref = so.fp8_rowwise_fake_quant(x)
    assert torch.equal(deq, ref)
    small = x.float().abs() < x.float().abs().amax(-1, keepdim=True) * 2.0 ** -10
    _record("kernels", "e4m3 rows", {"fraction_of_entries_flushed_to_zero": float((ref.float()[small] == 0).float().mean()),
                                     "fraction_small": float(small.float().mean())})


@pytest.mark.parametrize("level", ["moderate", "spec"])
@pytest.mark.parametrize("mode", ["bf16", "fp8", "fp8_w8a8"])
def test_true_width_4_layer_request_with_outlier_statistics(mode, level):
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel

    kw = dict(vit_layers=4, layers=4, vocab=16386, mask_token_id=16384, depth_token_id=16385)
    ocfg = so.SrgptConfig(**kw)
    dtype = torch.bfloat16
    w = so.synth_weights(ocfg, seed=11, dtype=dtype)
    planted = make_heavy_tailed(w, ocfg, **LEVELS[level])
    ids, images, depths, masks = so.synth_inputs(ocfg, batch=1, regions=8, prompt_len=64, seed=2, dtype=dtype)
    model = LlavaLlamaModel(SrgptConfig(**kw), dict(w), device=DEV, dtype=dtype, rope_positions=1024,
                            llm_weight_format={"bf16": "native"}.get(mode, mode))
    eng = model.engine
    wq = so.fp8_dequantised_weights(w) if mode != "bf16" else w
    aq = so.fp8_rowwise_fake_quant if mode == "fp8_w8a8" else None
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    G = 6
    ref_ids, st = so.generate(wq, ocfg, ids, images, depths, masks, max_new_tokens=G, return_stages=True, model_dtype=dtype,
                              prefill_act_quant=aq)
    # fp32 oracle on the same (bf16-valued) weights, same quantisers, teacher forced: the noise floor of bf16 arithmetic itself
    w32 = {k: v.float() for k, v in wq.items()}
    emb32, _, _, _ = so.prepare_inputs(w32, ocfg, ids, images.float(), depths.float(), [m.float() for m in masks])
    kv = so.KVCache(ocfg.layers)
    T = emb32.shape[1]
    pre32 = so.llama_forward(w32, ocfg, emb32, torch.arange(T)[None], kv, act_quant=aq)
    steps32 = [pre32[:, -1]]
    for t_ in range(G - 1):
        e = torch.nn.functional.embedding(ref_ids[:, t_:t_ + 1], w32["llm.model.embed_tokens.weight"])
        steps32.append(so.llama_forward(w32, ocfg, e, torch.tensor([[T + t_]]), kv, last_only=True)[:, -1])
    steps32 = torch.stack(steps32, 1)
    got = {}
    emb, _, _ = eng.prepare_inputs(ids.to(DEV), images.to(DEV), depths.to(DEV), [m.to(DEV) for m in masks], None, stages=got)
    report = {"mode": mode, "level": level, "level_settings": LEVELS[level], "planted": {k: (len(v) if isinstance(v, list) and k == "gains" else v) for k, v in planted.items()}, "stages": {}}
    for name in ("tower_features", "depth_features", "hres", "lres", "image_features"):
        a, b = got[name].float().cpu(), st[name].float()
        assert bool(torch.isfinite(a).all()), name
        report["stages"][name] = {"max_abs_over_max": float((a - b).abs().max() / b.abs().max()), "ref_absmax": float(b.abs().max())}
    for name in ("mask_embeds", "depth_embeds"):
        a, b = torch.stack(got[name]).float().cpu(), torch.stack(st[name]).float()
        assert bool(torch.isfinite(a).all()), name
        report["stages"][name] = {"max_abs_over_max": float((a - b).abs().max() / b.abs().max()), "ref_absmax": float(b.abs().max())}
    a, b = emb.float().cpu(), st["inputs_embeds"].float()
    report["stages"]["inputs_embeds"] = {"max_abs_over_max": float((a - b).abs().max() / b.abs().max()), "ref_absmax": float(b.abs().max())}
    # the LLM on the ORACLE's embeddings: layer arithmetic under test, not the accumulated vision noise
    stt, logits, _ = eng.prefill(st["inputs_embeds"].to(DEV), max_new=G + 1, all_logits=True)
    dec = teacher_forced_decode_logits(eng, stt, ref_ids)
    assert bool(torch.isfinite(logits).all()) and bool(torch.isfinite(dec).all())
    ok = True
    for name, got_l, ref16, ref32 in (("prefill", logits, st["prefill_logits"], pre32), ("decode", dec, st["step_logits"], steps32)):
        floor = logit_parity_report(ref16, ref32, 1.0, f"{name}: noise floor")
        tol_max, tol_rms = 2 * floor["max_abs_over_range"], 2 * floor["rms_over_range"]
        r16 = logit_parity_report(got_l, ref16, tol_max, f"{name}: engine vs oracle_bf16")
        r32 = logit_parity_report(got_l, ref32, tol_max, f"{name}: engine vs oracle_fp32")
        report[name] = {"noise_floor": floor, "vs_bf16_oracle": r16, "vs_fp32_oracle": r32, "tol_max": tol_max, "tol_rms": tol_rms}
        ok &= r16["max_abs_over_range"] <= tol_max and r16["rms_over_range"] <= tol_rms and r16["argmax_disagree_out_of_margin"] == 0
        ok &= r32["rms_over_range"] <= 1.25 * floor["rms_over_range"]
    if mode != "bf16":
        # how far the quantised model sits from the un-quantised one (fp32, original weights): the format's own effect
        w32o = {k: v.float() for k, v in w.items()}
        kv2 = so.KVCache(ocfg.layers)
        pre_plain = so.llama_forward(w32o, ocfg, emb32, torch.arange(T)[None], kv2)
        d = logit_parity_report(pre32, pre_plain, 1.0, "quantised fp32 oracle vs un-quantised fp32 oracle (prefill)")
        report["quantisation_effect_prefill"] = {"rms_over_range": d["rms_over_range"], "max_abs_over_range": d["max_abs_over_range"],
                                                 "argmax_agree": d["argmax_agree"], "rows": d["rows"]}
    _record("model", f"{mode}/{level}", report)
    print("\nOUTLIER", json.dumps(report)[:2000])
    assert ok, {k: report[k] for k in ("prefill", "decode")}


# ------------------------------------------------------------------------------------------------------------------------------
# Per-layer, teacher-forced.  The end-to-end bars above are noise-floor relative, and four layers of an
# untrained outlier model amplify bf16 noise until the floor itself is a large fraction of the logit range: a wrong layer could hide
# under `2 x floor`.  Here every layer is run ALONE on the oracle's input of that layer -- the engine as a ONE-layer model carrying that
# layer's weights (LLM: srgpt_llm_prefill's hidden-state hook; ViT: srgpt_vit_forward with a zero patch projection and the layer input
# as the position embedding) -- and compared with the oracle's output of the layer.  The floor is then ONE layer's rounding:
#   fp32 run of the same layer on the same (bf16-valued) input = the exact answer up to fp32 accumulation;
#   floor = oracle_bf16 - fp32 (what ONE layer of correctly rounded bf16 arithmetic costs on these statistics);
#   rms(engine - fp32) <= RMS_BAR x rms(floor)  over the whole layer output,   max|engine - fp32| <= MAX_BAR x max|floor|.
# What the bars resolve is measured next to them: the SAME fp32 layer with its output projection scaled by 1 + 2^-6 (a 1.6 % error
# in one of seven products) sits at `wrong_projection_rms_ratio` x the floor -- recorded per planting, and asserted to fail the bar.
# Measured (profiles/r06_outlier_per_layer.json): rms ratios 0.61 - 1.10 on every layer / mode / planting -- the engine is as close to
# the exact layer output as the reference arithmetic is; a layer with a wrong rounding point, a dropped term or a mis-scaled
# projection lands at many times the floor.  The error in bf16 ulps of the OUTPUT scale is recorded, not asserted: the planted hot
# rows and massive channels make the layer's INTERNAL values hundreds of times the output's typical size, and one ulp of those is
# tens of ulps of an ordinary output channel -- for the oracle exactly as for the engine.
REPORT6 = os.path.join(ROOT, "gpurun_out", "r06_outlier_per_layer.json")
RMS_BAR, MAX_BAR = 1.25, 2.0   # (the worst single token row's ratio is recorded, not asserted: a row whose own floor happens to be
                               # small turns an ordinary error into a large ratio -- up to 23 on one row of 729)


def _record6(key, value):
    os.makedirs(os.path.dirname(REPORT6), exist_ok=True)
    data = json.load(open(REPORT6)) if os.path.exists(REPORT6) else {}
    data[key] = value
    json.dump(data, open(REPORT6, "w"), indent=1)


def _layer_metrics(got, ref16, ref32):
    got, ref16, ref32 = got.double().cpu(), ref16.double(), ref32.double()
    assert bool(torch.isfinite(got).all())
    rms = lambda t: float(t.pow(2).mean().sqrt())  # noqa: E731
    floor, e32, e16 = rms(ref16 - ref32), rms(got - ref32), rms(got - ref16)
    robust = 1.48 * ref16.abs().median(dim=-1, keepdim=True).values
    bound = 2.0 ** -8 * torch.maximum(ref16.abs(), robust)
    ulps = float(((got - ref16).abs() / bound).max())
    ulps_floor = float(((ref32 - ref16).abs() / bound).max())
    row = lambda t: t.reshape(-1, t.shape[-1]).pow(2).mean(-1).sqrt()  # noqa: E731
    rf, re = row(ref16 - ref32), row(got - ref32)
    keep = rf >= 0.1 * floor  # (rows whose own floor is far below the layer's say nothing about a ratio)
    return {"out_rms": rms(ref32), "out_absmax": float(ref32.abs().max()), "floor_rms": floor, "engine_vs_fp32_rms": e32,
            "engine_vs_bf16_oracle_rms": e16, "rms_ratio_to_floor": e32 / max(floor, 1e-30),
            "worst_row_rms_ratio_to_floor": float((re[keep] / rf[keep]).max()),
            "max_ratio_to_floor": float((got - ref32).abs().max() / (ref16 - ref32).abs().max()),
            "recorded_ulps_of_output_scale": {"engine_vs_bf16_oracle": ulps, "fp32_vs_bf16_oracle": ulps_floor},
            "row_scale_median": float(robust.median())}


def _layer_ok(m):
    return m["rms_ratio_to_floor"] <= RMS_BAR and m["max_ratio_to_floor"] <= MAX_BAR


def _one_layer(w, prefix, i, n_layers_out=1):
    """weights dict with layer i of `prefix` renamed to layer 0 (and copied to layers 1.. n_layers_out - 1), the other layers dropped"""
    out = {}
    for k, v in w.items():
        if k.startswith(prefix):
            rest = k[len(prefix):]
            j, tail = rest.split(".", 1)
            if int(j) == i:
                for d in range(n_layers_out):
                    out[f"{prefix}{d}.{tail}"] = v
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("level", ["moderate", "spec"])
@pytest.mark.parametrize("mode", ["bf16", "fp8", "fp8_w8a8"])
def test_every_llm_layer_alone_on_the_oracles_input(mode, level):
    import dataclasses

    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine

    kw = dict(vit_layers=4, layers=4, vocab=16386, mask_token_id=16384, depth_token_id=16385)
    ocfg = so.SrgptConfig(**kw)
    dtype = torch.bfloat16
    w = so.synth_weights(ocfg, seed=11, dtype=dtype)
    make_heavy_tailed(w, ocfg, **LEVELS[level])
    ids, images, depths, masks = so.synth_inputs(ocfg, batch=1, regions=8, prompt_len=64, seed=2, dtype=dtype)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    wq = so.fp8_dequantised_weights(w) if mode != "bf16" else w
    aq = so.fp8_rowwise_fake_quant if mode == "fp8_w8a8" else None
    emb, _, _, _ = so.prepare_inputs(w, ocfg, ids, images, depths, masks)
    T = emb.shape[1]
    pos = torch.arange(T)[None]
    _, hiddens = so.llama_forward(wq, ocfg, emb, pos, so.KVCache(ocfg.layers), collect_hidden=True, act_quant=aq)
    LP = "llm.model.layers."
    ocfg1 = dataclasses.replace(ocfg, layers=1)
    ecfg1 = SrgptConfig(**dict(kw, layers=1))
    rep, ok = {}, True
    for i in range(ocfg.layers):
        x_in, ref16 = hiddens[i], hiddens[i + 1]
        w32 = {k: v.float() for k, v in _one_layer(wq, LP, i).items() if k.startswith("llm.")}
        _, h32 = so.llama_forward(w32, ocfg1, x_in.float(), pos, so.KVCache(1), collect_hidden=True, act_quant=aq)
        sd = {k: v for k, v in _one_layer(w, LP, i).items() if k.startswith("llm.")}
        eng = SrgptEngine(ecfg1, sd, device=DEV, dtype=dtype, rope_positions=1024, parts=("llm",),
                          llm_weight_format={"bf16": "native"}.get(mode, mode))
        _, _, hs = eng.prefill(x_in.to(DEV), max_new=1, hidden_states=True)
        m = _layer_metrics(hs[1], ref16, h32[1])
        if i == 0:  # negative control: what a 1.6 % error in ONE projection of this layer looks like against the same floor
            wbad = dict(w32)
            wbad[LP + "0.self_attn.o_proj.weight"] = w32[LP + "0.self_attn.o_proj.weight"] * (1 + 2.0 ** -6)
            _, hbad = so.llama_forward(wbad, ocfg1, x_in.float(), pos, so.KVCache(1), collect_hidden=True, act_quant=aq)
            m["wrong_projection_rms_ratio"] = float((hbad[1].double() - h32[1].double()).pow(2).mean().sqrt()) / max(m["floor_rms"], 1e-30)
        rep[f"layer {i}"] = m
        ok &= _layer_ok(m)
        del eng
    _record6(f"llm/{mode}/{level}", rep)
    print("\nPER-LAYER", mode, level, json.dumps(rep))
    assert ok, rep


@pytest.mark.parametrize("level", ["moderate", "spec"])
def test_every_vit_layer_alone_on_the_oracles_input(level):
    import dataclasses

    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine

    kw = dict(vit_layers=4, layers=4, vocab=16386, mask_token_id=16384, depth_token_id=16385)
    ocfg = so.SrgptConfig(**kw)
    dtype = torch.bfloat16
    w = so.synth_weights(ocfg, seed=11, dtype=dtype)
    make_heavy_tailed(w, ocfg, **LEVELS[level])
    _, images, _, _ = so.synth_inputs(ocfg, batch=1, regions=8, prompt_len=64, seed=2, dtype=dtype)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ocfg_all = dataclasses.replace(ocfg, select_layer=ocfg.vit_layers)  # run (and collect) all four layers
    hiddens = []
    so.vit_forward(w, ocfg_all, images, collect_hidden=hiddens)
    assert len(hiddens) == ocfg.vit_layers + 1
    VP = so.VT + "encoder.layers."
    pe, pw, pb = so.VT + "embeddings.position_embedding.weight", so.VT + "embeddings.patch_embedding.weight", so.VT + "embeddings.patch_embedding.bias"
    ocfg1 = dataclasses.replace(ocfg, vit_layers=2, select_layer=-2)   # hidden_states[-2] of a two-layer tower = ONE layer
    ecfg1 = SrgptConfig(**dict(kw, vit_layers=2, select_layer=-2))
    rep, ok = {}, True
    for i in range(ocfg.vit_layers):
        x_in, ref16 = hiddens[i][0], hiddens[i + 1][0]
        # a tower whose embeddings ARE the layer input: zero patch projection, the input as the position embedding
        w1 = {k: v for k, v in _one_layer(w, VP, i, n_layers_out=2).items() if k.startswith("vision_tower.")}
        w1[pw], w1[pb], w1[pe] = torch.zeros_like(w[pw]), torch.zeros_like(w[pb]), x_in.clone()
        h32 = []
        so.vit_forward({k: v.float() for k, v in w1.items()}, ocfg1, images.float(), collect_hidden=h32)
        eng = SrgptEngine(ecfg1, dict(w1), device=DEV, dtype=dtype, parts=("vit",))
        got = eng.vit(images.to(DEV))[0]
        m = _layer_metrics(got, ref16, h32[1][0])
        rep[f"layer {i}"] = m
        ok &= _layer_ok(m)
        del eng
    _record6(f"vit/bf16/{level}", rep)
    print("\nPER-LAYER vit", level, json.dumps(rep))
    assert ok, rep
