"""Probe: CPU oracle decode-step time vs thread count / dtype on the bench host (sizing the cpu_baseline leg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import srgpt_oracle as so

cfg = so.SrgptConfig(layers=2, vit_layers=3)
for dtype in (torch.bfloat16, torch.float32):
    w = {k: v for k, v in so.synth_weights(so.SrgptConfig(layers=2, vit_layers=1, vocab=128258), seed=0, dtype=dtype).items()}
    for nt in (8, 16, 32, 64, 128):
        torch.set_num_threads(nt)
        kv = so.KVCache(2)
        x = torch.randn((1, 259, 4096)).to(dtype)
        with torch.no_grad():
            t0 = time.perf_counter(); so.llama_forward(w, cfg, x, torch.arange(259)[None], kv); tp = time.perf_counter() - t0
            e = torch.randn((1, 1, 4096)).to(dtype)
            t0 = time.perf_counter()
            for s in range(3):
                so.llama_forward(w, cfg, e, torch.tensor([[259 + s]]), kv, last_only=True)
            td = (time.perf_counter() - t0) / 3
        print(f"dtype={dtype} threads={nt} prefill(2L+head)={tp:.3f}s decode_step(2L+head)={td*1e3:.1f}ms", flush=True)
