"""Microbenchmark of the decode GEMV at the VILA1.5-8B shapes inside a captured graph (per-kernel us, TB/s).
Each measurement replays a graph of 32 launches over 32 distinct weight matrices (cold in the 256 MB L3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import ops

dev = "cuda"
torch.manual_seed(0)
shapes = [("qkv+norm", 6144, 4096, dict(norm=True)), ("o+res", 4096, 4096, dict(res=True)), ("gateup+norm", 14336, 4096, dict(norm=True, swiglu=True)), ("down+res", 4096, 14336, dict(res=True)), ("n2048", 2048, 4096, {})]
L = 32
side = torch.cuda.Stream()
for name, N, K, opt in shapes:
    rows = 2 * N if opt.get("swiglu") else N
    Ws = [torch.randn((rows, K), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(L)]
    x = torch.randn((1, K), device=dev, dtype=torch.bfloat16)
    g = (1 + 0.1 * torch.randn((K,), device=dev)).to(torch.bfloat16) if opt.get("norm") else None
    res = torch.randn((1, N), device=dev, dtype=torch.bfloat16) if opt.get("res") else None
    out = torch.empty((1, N), device=dev, dtype=torch.bfloat16)

    def run():
        for W in Ws:
            ops.gemv(x, W, norm_w=g, eps=1e-5, residual=res, swiglu=bool(opt.get("swiglu")), out=out)

    run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        run()
    for _ in range(2):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * L)
    mb = rows * K * 2 / 1e6
    print(f"{name:14s} N={N:6d} K={K:6d} {mb:7.1f} MB  {us:7.2f} us/launch  {mb / us / 1e6 * 1e6 / 1e6 * 1e0:6.3f} TB/s  (fixed vs 7.05TB/s: {us - mb / 7.05:5.2f} us)", flush=True)
    del Ws
