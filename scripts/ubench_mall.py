"""Does a prefetch pass leave weights in the 256 MiB Infinity Cache for the next (nt-load) GEMV?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatialrgpt_amd import ops, _lib as L

lib = L.load()
dev = "cuda"
L_ = 32
side = torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * L_)


for name, N, K in [("o", 4096, 4096), ("qkv", 6144, 4096), ("down", 4096, 14336), ("gateup/2", 14336, 4096)]:
    Ws = [torch.randn((N, K), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(L_)]
    x = torch.randn((1, K), device=dev, dtype=torch.bfloat16)
    out = torch.empty((1, N), device=dev, dtype=torch.bfloat16)
    nbytes = N * K * 2

    def gemv_only():
        for W in Ws:
            ops.gemv(x, W, out=out)

    def pf_only(blocks=512):
        for W in Ws:
            L.check(lib.srgpt_prefetch(W.data_ptr(), nbytes, blocks, torch.cuda.current_stream().cuda_stream))

    def pf_then_gemv():
        for W in Ws:
            L.check(lib.srgpt_prefetch(W.data_ptr(), nbytes, 512, torch.cuda.current_stream().cuda_stream))
            ops.gemv(x, W, out=out)

    a, b, c = timed(gemv_only), timed(pf_only), timed(pf_then_gemv)
    print(f"{name:9s} {nbytes / 1e6:6.1f} MB  gemv cold {a:6.2f} us | prefetch {b:6.2f} us | prefetch+gemv {c:6.2f} us -> gemv warm ~{c - b:6.2f} us", flush=True)
    del Ws
