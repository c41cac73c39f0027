"""Decoding modes of the interactive callers, timed on the headline geometry (VILA1.5-8B, 8 regions, bs 1, 128 new tokens):
greedy (the benchmarked loop), the demo's sampling settings (demo/gradio_web_server_multi.py:202-213: temperature 0.2, top_k 50 by the
transformers 4.37.2 default), sampling + top-p, Gumbel-max (no top-k), each with and without a stopping criterion (host Python after
every token, judged one step behind the device: engine._decode_loop_run_ahead), and the torch-op path a setting outside the device
sampler takes.   python scripts/experiments/time_sampling.py > gpurun_out/sampling.txt   -> profiles/r04_sampling.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from spatialrgpt_amd.config import SrgptConfig  # noqa: E402
from spatialrgpt_amd.model import LlavaLlamaModel  # noqa: E402
from spatialrgpt_amd.weights import synth_state_dict  # noqa: E402

dev = "cuda"
cfg = SrgptConfig.vila15_8b()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev)
model = LlavaLlamaModel(cfg, sd, device=dev, dtype=torch.bfloat16, rope_positions=1024, consume_state_dict=True)
req = bench.synth_request(cfg, 8, 64, 1, dev, torch.bfloat16)
G = 128


def never(ids, scores):  # a criterion that never fires: the whole budget runs, the host is consulted after every token
    return False


def run(**kw):
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = model.generate(req[0], images=req[1], depths=req[2], masks=req[3], max_new_tokens=G, **{'eos_token_id': None, **kw})
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    assert out.shape == (1, G)
    return best


rows = [("greedy", dict(do_sample=False)),
        ("greedy, EOS id set (never met: judged every step, one behind)", dict(do_sample=False, eos_token_id=2)),
        ("greedy + stopping criterion (run-ahead)", dict(do_sample=False, stopping_criteria=[never])),
        ("sample T=0.2 top_k=50 (the demo)", dict(do_sample=True, temperature=0.2)),
        ("sample T=0.2 top_k=50 + stopping criterion (the demo)", dict(do_sample=True, temperature=0.2, stopping_criteria=[never])),
        ("sample T=0.2 top_k=50 top_p=0.9", dict(do_sample=True, temperature=0.2, top_p=0.9)),
        ("sample T=0.7 no top-k (Gumbel-max)", dict(do_sample=True, temperature=0.7, top_k=0)),
        ("sample T=0.2 top_p=0.9 without top-k (torch ops per token)", dict(do_sample=True, temperature=0.2, top_k=0, top_p=0.9))]
base = None
print(f"VILA1.5-8B geometry, bf16, 8 regions, bs 1, {G} new tokens, best of 3; whole request (vision + prefill + decode)")
for name, kw in rows:
    dt = run(**kw)
    base = base or dt
    print(f"{name:62s} {dt * 1e3:8.1f} ms  {G / dt:7.1f} tok/s  {base / dt:5.3f} x greedy")
