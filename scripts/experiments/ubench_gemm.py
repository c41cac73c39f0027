"""GEMM microbenchmark at the prefill / ViT / extractor shapes (graph-captured, distinct weights per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
if os.environ.get("SRGPT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SRGPT_LIB"])
from spatialrgpt_amd import ops

dev = "cuda"
shapes = [("llm qkv", 259, 6144, 4096), ("llm o", 259, 4096, 4096), ("llm gate/up", 259, 28672, 4096), ("llm down", 259, 4096, 14336),
          ("vit qkv", 1458, 3456, 1152), ("vit out", 1458, 1152, 1152), ("vit fc1", 1458, 4304, 1152), ("vit fc2", 1458, 1152, 4304),
          ("proj 1", 196, 4096, 4608), ("deconv1", 729, 4608, 1152), ("deconv2", 2916, 4608, 1152), ("sq 4096", 4096, 4096, 4096)]
if len(sys.argv) > 1:  # shapes from the command line: name:M:N:K ...
    shapes = [(a.split(":")[0], *map(int, a.split(":")[1:])) for a in sys.argv[1:]]
side = torch.cuda.Stream()
tot = 0.0
for name, M, N, K in shapes:
    L = 8 if N * K < 2e8 else 4
    Ws = [torch.randn((N, K), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(L)]
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)

    def run():
        for W in Ws:
            ops.gemm(a, W, out=out)

    run(); torch.cuda.synchronize()
    if os.environ.get("UBENCH_CHECK"):  # output of the last launch against fp32 torch (relative to the output's max)
        ref = a.float() @ Ws[-1].float().T
        print(f"   check: max err / max = {float((out.float() - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * L)
    print(f"{name:12s} M={M:5d} N={N:6d} K={K:6d}  {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s  W {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
    del Ws
