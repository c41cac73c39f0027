"""GPU: LONG-CONTEXT parity against the CPU oracle (VERDICT r3 "missing" #2 / "next" #4).

The shipped recipes run `--model_max_length 4096` (scripts/srgpt/llama3_8b/3_sft.sh:58), `model_vqa.py:77` asks for 1024 new tokens,
the demo defaults to 512 over a growing multi-turn history (demo/gradio_web_server_multi.py:330), and `context_length_extension`
(llava/model/language_model/builder.py:31-38) stretches the rotary positions linearly when `model_max_length` exceeds the LLM's
`max_position_embeddings`.  Everything the other model-level tests run stays below ~830 positions; this file covers

  (0) the reference's VENDORED Llama under rope_scaling {linear, 3.0} (tests/golden/vendored_llama_kat.npz: the real
      modeling_llama.py executed on CPU, oracle/make_golden.py vendored): fp32 engine logits at every prompt position, the ragged
      right-padded batch, and 10 greedy steps at positions beyond max_position_embeddings -- ids equal, logits at fp32 tolerance;
  (i) Llama-3-8B geometry (hidden 4096, GQA 32/8, inter 14336, vocab 128258) truncated to 4 layers behind a tiny tower, the whole
      request (tower, pooling, projector, splice) with a prompt spliced to T = 2048 and T = 4000, then 64 TEACHER-FORCED decode
      steps: inputs_embeds, the hidden state after every layer at EVERY position, the last-position prefill logits and the logits
      of all 64 steps against the bf16 oracle, with the noise-floor-calibrated tolerances of tests/test_gpu_fulldepth_parity.py
      (the oracle run a second time in fp32 on the same bf16-valued weights gives the floor);
 (ii) the same at T = 4000 with rope_factor = 2.0 (max_position_embeddings 2048 -> model_max_length 4096);
(iii) a 1024-token free-running greedy run through `model.generate` (the graph-captured loop) on `tests.util.make_peaked`
      weights: contexts 259 .. 1282 (the decode attention goes from 5 to 21 live 64-key splits -- past the 16 the merge prefetches
      in one batch -- and the cache crosses eight 128-row granules); all 1024 ids equal to the oracle's, margin >= 10 x the measured
      bf16 noise floor asserted from the oracle's logits.
Measured numbers -> gpurun_out/longctx_parity.json (committed as profiles/r04_longctx_parity.json)."""
import json
import os
import sys
import time

import numpy as np
import pytest
import torch

from tests.util import GOLD, assert_close, logit_parity_report, make_peaked, teacher_forced_decode_logits

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TINY_TOWER = dict(vit_hidden=64, vit_inter=176, vit_layers=3, vit_heads=4, image_size=378, patch_size=14)
LLAMA3_4L = dict(hidden=4096, inter=14336, layers=4, heads=32, kv_heads=8, vocab=128258, rope_theta=500000.0,
                 mask_token_id=128256, depth_token_id=128257)
LOGIT_MAX_CAP, LOGIT_RMS_CAP = 0.15, 2.5e-2
THREADS = 16
FLOOR_STEPS = 6
REPORT = os.path.join(ROOT, "gpurun_out", "longctx_parity.json")


def _record(tag, rep):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    allr = {}
    if os.path.exists(REPORT):
        with open(REPORT) as f:
            allr = json.load(f)
    allr[tag] = rep
    with open(REPORT, "w") as f:
        json.dump(allr, f, indent=1)


# ------------------------------------------------------------------------------------------------ (0)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_engine_matches_the_vendored_llama_under_linear_rope_scaling(dtype):
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine
    from spatialrgpt_amd.weights import synth_state_dict

    z = np.load(os.path.join(GOLD, "vendored_llama_kat.npz"))
    geo = json.loads(bytes(z["geo_json"]).decode())
    cfg = SrgptConfig(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=42, patch_size=14,
                      hidden=geo["hidden_size"], inter=geo["intermediate_size"], layers=geo["num_hidden_layers"],
                      heads=geo["num_attention_heads"], kv_heads=geo["num_key_value_heads"], vocab=geo["vocab_size"],
                      rms_eps=geo["rms_norm_eps"], rope_theta=geo["rope_theta"], rope_factor=geo["rope_factor"],
                      mask_token_id=126, depth_token_id=127, max_position_embeddings=96)
    sd = synth_state_dict(cfg, seed=1, dtype=dtype, device=DEV)
    for k in z.files:
        if k.startswith("w."):
            sd[k[2:]] = torch.from_numpy(z[k]).to(device=DEV, dtype=dtype)
    eng = SrgptEngine(cfg, sd, device=DEV, dtype=dtype)
    assert eng.w.rope_len == 96
    bf = dtype == torch.bfloat16
    pre = torch.from_numpy(z["bf16.prefill_logits" if bf else "single.prefill_logits"])
    steps = torch.from_numpy(z["bf16.step_logits" if bf else "single.step_logits"])
    new_ids = torch.from_numpy(z["bf16.new_ids" if bf else "single.new_ids"])
    rng = float(pre.abs().max())
    tol = (3e-2 if bf else 5e-5) * rng
    ids = torch.from_numpy(z["single.ids"]).to(DEV)
    st, logits, _ = eng.prefill(eng.embed_tokens(ids), max_new=new_ids.shape[1] + 1, all_logits=True)
    assert_close(logits, pre, tol, 0, "vendored: all-position prefill logits")
    # decode steps: the vendored run fed its own argmax; step t consumes new_ids[t] at position T + t
    dec = teacher_forced_decode_logits(eng, st, torch.cat([new_ids, new_ids[:, -1:]], 1))[:, 1:]
    assert_close(dec, steps, tol, 0, "vendored: decode-step logits (positions beyond max_position_embeddings)")
    if not bf:
        assert torch.equal(logits[:, -1].argmax(-1).cpu(), new_ids[:, 0])
        assert torch.equal(dec.argmax(-1).cpu()[:, :-1], new_ids[:, 1:]), "greedy ids differ from the vendored Llama's"
        # ragged right-padded batch (the unpad / varlen branch of the vendored flash forward)
        ids2, lens = torch.from_numpy(z["ragged.ids"]).to(DEV), z["ragged.lens"].tolist()
        _, lg2, _ = eng.prefill(eng.embed_tokens(ids2), max_new=1, all_logits=True, lens=torch.tensor(lens))
        for b, n in enumerate(lens):
            assert_close(lg2[b, :n], torch.from_numpy(z["ragged.logits"])[b, :n], tol, 0, f"vendored: ragged row {b}")


# ------------------------------------------------------------------------------------------------ (i), (ii)
def _teacher_forced_case(tag, T_target, rope_factor, mpe, G=64):
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel
    from spatialrgpt_amd.weights import synth_state_dict

    dtype = torch.bfloat16
    cfg = SrgptConfig(**TINY_TOWER, **LLAMA3_4L, rope_factor=rope_factor, max_position_embeddings=mpe)
    ocfg = so.SrgptConfig(**{k: v for k, v in cfg.to_dict().items() if k in so.SrgptConfig.__dataclass_fields__})
    assert ocfg.rope_factor == rope_factor
    sd = synth_state_dict(cfg, seed=3, dtype=dtype, device=DEV)
    w_cpu = {k: v.cpu() for k, v in sd.items()}
    model = LlavaLlamaModel(cfg, sd, device=DEV, dtype=dtype, consume_state_dict=True)
    del sd
    eng = model.engine
    assert eng.w.rope_len == mpe
    prompt_len = T_target - 196 + 1
    ids, images, depths, masks = so.synth_inputs(ocfg, batch=1, regions=8, prompt_len=prompt_len, seed=4, dtype=dtype)
    torch.set_num_threads(min(THREADS, os.cpu_count() or 1))

    def oracle_pass(w, wdt, teacher=None, steps=G):
        emb, _, _, st = so.prepare_inputs(w, ocfg, ids, images.to(wdt), depths.to(wdt), [m.to(wdt) for m in masks])
        kv = so.KVCache(ocfg.layers)
        T = emb.shape[1]
        lg, hid = so.llama_forward(w, ocfg, emb, torch.arange(T)[None], kv, last_only=True, collect_hidden=True)
        out, new = [lg[:, -1]], []
        for t in range(steps - 1):
            nxt = out[-1].argmax(-1) if teacher is None else teacher[:, t]
            new.append(nxt)
            e = torch.nn.functional.embedding(nxt[:, None], w["llm.model.embed_tokens.weight"])
            out.append(so.llama_forward(w, ocfg, e, torch.tensor([[T + t]]), kv, last_only=True)[:, -1])
        new.append(out[-1].argmax(-1))
        return emb, torch.stack(hid, 0), torch.stack(out, 1), torch.stack(new, 1)

    t0 = time.perf_counter()
    emb_ref, hid_ref, steps_ref, ref_ids = oracle_pass(w_cpu, dtype)
    t_oracle = time.perf_counter() - t0
    T = emb_ref.shape[1]
    assert T == T_target
    t0 = time.perf_counter()
    w32 = {k: v.float() for k, v in w_cpu.items()}
    del w_cpu
    _, hid32, steps32, _ = oracle_pass(w32, torch.float32, teacher=ref_ids, steps=FLOOR_STEPS)
    del w32
    t_oracle32 = time.perf_counter() - t0

    d = lambda t: t.to(DEV)  # noqa: E731
    emb, _, lens = eng.prepare_inputs(d(ids), d(images), d(depths), [d(m) for m in masks], None)
    assert emb.shape == (1, T, cfg.hidden) and lens == [T]
    rep = {"tag": tag, "T": T, "G": G, "rope_factor": rope_factor, "max_position_embeddings": mpe, "layers": cfg.layers,
           "oracle_s": round(t_oracle, 1), "oracle_fp32_s": round(t_oracle32, 1), "stages": {}}

    def chk(a, b, what, rel=2.5e-2):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        scale = float(b.abs().max()) + 1e-6
        rep["stages"][what] = {"max_abs_over_max": float((a - b).abs().max()) / scale,
                               "rms_over_max": float((a - b).pow(2).mean().sqrt()) / scale, "tol": rel}
        assert_close(a, b, rel * scale, 0, what)

    chk(emb, emb_ref, "inputs_embeds")
    st, _, hs = eng.prefill(emb, max_new=G + 1, hidden_states=True)
    assert st.max_pos >= T + G
    # hidden states, all T positions x 4096 channels (8 - 16 M elements per layer): a fixed max-abs bound over that many bf16
    # values sits on the tail of the rounding noise (the first GPU run: 3 of 8.4 M elements of layer 3 at 2.8e-2 of the max), so
    # the bound is calibrated like the logits -- floor = oracle_bf16 - oracle_fp32 of the same layer: max <= 2 x floor max (never
    # tighter than 2.5e-2 of the tensor's max), rms <= 2 x floor rms
    for i in range(cfg.layers + 1):
        ref16, ref32, gotl = hid_ref[i].float(), hid32[i].float(), hs[i].float().cpu()
        scale = float(ref16.abs().max()) + 1e-6
        fl_max, fl_rms = float((ref16 - ref32).abs().max()), float((ref16 - ref32).pow(2).mean().sqrt())
        e_max, e_rms = float((gotl - ref16).abs().max()), float((gotl - ref16).pow(2).mean().sqrt())
        rep["stages"][f"hidden state after layer {i}"] = {"max_abs_over_max": e_max / scale, "rms_over_max": e_rms / scale,
                                                          "floor_max_over_max": fl_max / scale, "floor_rms_over_max": fl_rms / scale}
        assert e_max <= max(2.5e-2 * scale, 2 * fl_max), (i, rep["stages"])
        assert e_rms <= 2 * fl_rms + 1e-6 * scale, (i, rep["stages"])
    del hs, hid32
    dec = teacher_forced_decode_logits(eng, st, ref_ids)
    floor = logit_parity_report(steps_ref[:, :FLOOR_STEPS], steps32, 1.0, "NOISE FLOOR oracle_bf16 vs oracle_fp32")
    tol_max = min(LOGIT_MAX_CAP, 2 * floor["max_abs_over_range"])
    tol_rms = min(LOGIT_RMS_CAP, 2 * floor["rms_over_range"])
    r16 = logit_parity_report(dec, steps_ref, tol_max, "last prompt position + decode steps: engine vs oracle_bf16")
    r32 = logit_parity_report(dec[:, :FLOOR_STEPS], steps32, tol_max, "engine vs oracle_fp32")
    r16.update(tol_max=tol_max, tol_rms=tol_rms)
    rep.update(noise_floor=floor, vs_bf16_oracle=r16, vs_fp32_oracle=r32,
               engine_argmax_equals_oracle_ids=int((dec.argmax(-1).cpu() == ref_ids).sum()))
    _record(tag, rep)
    print("\nLONGCTX", json.dumps({k: v for k, v in rep.items() if k != "stages"}))
    assert r16["max_abs_over_range"] <= tol_max and r16["rms_over_range"] <= tol_rms, rep
    assert r16["argmax_disagree_out_of_margin"] == 0, rep
    assert r32["rms_over_range"] <= 1.25 * floor["rms_over_range"], rep
    del model
    torch.cuda.empty_cache()


def test_llama3_geometry_prompt_spliced_to_2048_positions_64_teacher_forced_steps():
    _teacher_forced_case("T2048", 2048, 1.0, 4096)


def test_llama3_geometry_prompt_spliced_to_4000_positions_64_teacher_forced_steps():
    _teacher_forced_case("T4000", 4000, 1.0, 4096 + 128)


def test_llama3_geometry_4000_positions_with_linear_rope_scaling_2048_to_4096():
    """context_length_extension: max_position_embeddings 2048, model_max_length 4096 -> rope_scaling {linear, 2.0}; the loader
    sizes the tables for model_max_length positions (spatialrgpt_amd/builder.py)."""
    _teacher_forced_case("T4000_rope2", 4000, 2.0, 4096)


# ------------------------------------------------------------------------------------------------ (iii)
def test_1024_free_running_greedy_tokens_bit_identical_to_the_oracle():
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel
    from spatialrgpt_amd.weights import synth_state_dict

    G, dtype = 1024, torch.bfloat16
    cfg = SrgptConfig(**TINY_TOWER, **LLAMA3_4L, max_position_embeddings=2048)
    ocfg = so.SrgptConfig(**{k: v for k, v in cfg.to_dict().items() if k in so.SrgptConfig.__dataclass_fields__})
    sd = synth_state_dict(cfg, seed=0, dtype=dtype, device=DEV)
    make_peaked(sd, cfg)
    w_cpu = {k: v.cpu() for k, v in sd.items()}
    model = LlavaLlamaModel(cfg, sd, device=DEV, dtype=dtype, consume_state_dict=True)
    del sd
    ids, images, depths, masks = so.synth_inputs(ocfg, batch=1, regions=8, prompt_len=64, seed=2, dtype=dtype)
    torch.set_num_threads(min(THREADS, os.cpu_count() or 1))
    t0 = time.perf_counter()
    ref_ids, st = so.generate(w_cpu, ocfg, ids, images, depths, masks, max_new_tokens=G, return_stages=True, model_dtype=dtype)
    t_oracle = time.perf_counter() - t0
    assert ref_ids.shape == (1, G) and len(set(ref_ids[0].tolist())) == G
    step_ref = st["step_logits"].float()
    top2 = step_ref.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    # noise floor: fp32 oracle, teacher forced, first FLOOR_STEPS steps and the LAST ones (longest contexts)
    w32 = {k: v.float() for k, v in w_cpu.items()}
    del w_cpu
    emb32, _, _, _ = so.prepare_inputs(w32, ocfg, ids, images.float(), depths.float(), [m.float() for m in masks])
    kv = so.KVCache(ocfg.layers)
    T = emb32.shape[1]
    f32 = [so.llama_forward(w32, ocfg, emb32, torch.arange(T)[None], kv, last_only=True)[:, -1]]
    for t in range(FLOOR_STEPS - 1):
        e = torch.nn.functional.embedding(ref_ids[:, t:t + 1], w32["llm.model.embed_tokens.weight"])
        f32.append(so.llama_forward(w32, ocfg, e, torch.tensor([[T + t]]), kv, last_only=True)[:, -1])
    del w32, kv
    floor = logit_parity_report(step_ref[:, :FLOOR_STEPS], torch.stack(f32, 1), 1.0, "NOISE FLOOR")
    floor_abs = floor["max_abs_over_range"] * floor["logit_range"]
    rep = {"T": int(T), "G": G, "oracle_s": round(t_oracle, 1), "margin_min": float(margin.min()), "noise_floor_max_abs": floor_abs,
           "margin_over_floor": float(margin.min()) / max(floor_abs, 1e-9)}
    assert rep["margin_over_floor"] >= 10.0, rep
    d = lambda t: t.to(DEV)  # noqa: E731
    out = model.generate(d(ids), images=d(images), depths=d(depths), masks=[d(m) for m in masks], do_sample=False,
                         max_new_tokens=G, eos_token_id=None).cpu()
    same = int((out == ref_ids).sum())
    bad = (out != ref_ids).nonzero()
    rep.update(ids_equal=same, first_mismatch_step=None if same == G else int(bad[0, 1]), contexts=[int(T), int(T) + G - 1])
    _record("greedy1024", rep)
    print("\nLONGCTX-GREEDY", json.dumps(rep))
    assert same == G, rep
