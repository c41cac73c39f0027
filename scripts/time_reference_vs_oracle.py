"""CPU timing of the REAL reference generate() beside the oracle (the CPU restatement bench.py's `cpu_baseline` times), same
weights, same inputs, same thread count -- build container only (needs /root/reference; nothing here runs on the GPU box).

  python scripts/time_reference_vs_oracle.py            -> profiles/r02_cpu_reference_vs_oracle.txt

The model is a scaled-down VILA geometry (the reference's own builders, seeded random weights) large enough for the timings to
be GEMM- rather than Python-bound: Llama 8 layers / 1024 / GQA 16:4 / inter 2816 / vocab 8192 behind a SigLIP 6 layers / 384 /
378 px tower, bf16 (the eval dtype, eval_spatial.py:221), 8 region masks, 64-id prompt, 32 greedy tokens.  Checks that both
produce the same ids under fp32 first."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden as mg  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402
from oracle import srgpt_oracle as so  # noqa: E402

LLM = dict(vocab_size=8192, hidden_size=1024, intermediate_size=2816, num_hidden_layers=8, num_attention_heads=16,
           num_key_value_heads=4, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=500000.0,
           tie_word_embeddings=False, bos_token_id=1, eos_token_id=2, attention_bias=False)
VIT = dict(hidden_size=384, intermediate_size=1536, num_hidden_layers=6, num_attention_heads=6, image_size=378,
           patch_size=14, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh")
G = 32


def timed(fn, reps=2):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return out, best


def main():
    threads = int(os.environ.get("SRGPT_CPU_THREADS", str(min(16, os.cpu_count() or 1))))
    torch.set_num_threads(threads)
    lines = [f"# real reference generate() vs the oracle (oracle/srgpt_oracle.py) on CPU, {threads} threads, "
             f"{os.cpu_count()} cores visible; scripts/time_reference_vs_oracle.py",
             f"# model: Llama {LLM['num_hidden_layers']}L/{LLM['hidden_size']}/GQA {LLM['num_attention_heads']}:{LLM['num_key_value_heads']}"
             f"/inter {LLM['intermediate_size']}/vocab {LLM['vocab_size']} + SigLIP {VIT['num_hidden_layers']}L/{VIT['hidden_size']}/378px, "
             f"8 regions, 64-id prompt (T = 259), {G} greedy tokens"]
    with tempfile.TemporaryDirectory() as td:
        model, tok = rh.build_tiny_reference_model(td, llm=LLM, vit=VIT, dtype="torch.float32", seed=0)
    for dtype in (torch.float32, torch.bfloat16):
        model = model.to(dtype)
        cfg = mg.cfg_from(model, tok)
        w = mg.canonical_state_dict(model)
        ids, images, depths, masks = so.synth_inputs(cfg, batch=1, regions=8, prompt_len=64, seed=1, dtype=dtype)

        def ref():
            with torch.no_grad():
                return model.generate(input_ids=ids, images=images, depths=depths, masks=masks, do_sample=False,
                                      max_new_tokens=G, use_cache=True, eos_token_id=None, pad_token_id=0, min_new_tokens=G)

        def ora():
            return so.generate(w, cfg, ids, images, depths, masks, max_new_tokens=G, model_dtype=dtype)

        r_ids, r_t = timed(ref)
        o_ids, o_t = timed(ora)
        o_ids = o_ids[0] if isinstance(o_ids, tuple) else o_ids
        same = bool(torch.equal(r_ids, o_ids))
        agree = float((r_ids == o_ids).float().mean())
        if dtype == torch.float32:
            assert same, "fp32 greedy ids differ between the reference and the oracle"
        lines.append(f"{str(dtype):15s} reference generate() {r_t:7.3f} s = {G / r_t:6.2f} tok/s | oracle {o_t:7.3f} s = {G / o_t:6.2f} tok/s | "
                     f"oracle/reference time {o_t / r_t:.2f} | ids identical: {same} (agreement {agree:.2f})")
        print(lines[-1], flush=True)
    lines.append("# the oracle is the same arithmetic as the reference's path (bit-identical stages, oracle/make_golden.py) without the HF "
                 "module / generation-loop overhead, so it is an upper bound on the reference's CPU throughput for this path: timing it as "
                 "bench.py's cpu_baseline (kind \"port\") does not flatter the GPU/CPU ratio.")
    out = os.path.join(ROOT, "profiles", "r02_cpu_reference_vs_oracle.txt")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
