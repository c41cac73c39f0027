"""GEMM at the shapes that fill the chip (batched ViT / batched prefill / large squares).  `--tuning` loads the A/B build
(libsrgpt_hip_tuning.so, `make -C spatialrgpt_amd/csrc TUNING=1`) so that SRGPT_GEMM_FORCE_256=-1 / 1 selects the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
if "--tuning" in sys.argv:
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libsrgpt_hip.so", "libsrgpt_hip_tuning.so")
from spatialrgpt_amd import ops
dev = "cuda"
shapes = [("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
          ("vit b8 qkv", 11664, 3456, 1152), ("vit b8 out", 11664, 1152, 1152), ("vit b8 fc1", 11664, 4304, 1152),
          ("vit b8 fc2 (K pad)", 11664, 1152, 4352), ("vit b4 fc1", 5832, 4304, 1152), ("vit b1 fc1", 1458, 4304, 1152),
          ("prefill b8 qkv", 2072, 6144, 4096), ("prefill b8 o", 2072, 4096, 4096), ("prefill b8 gate/up", 2072, 28672, 4096),
          ("prefill b8 down", 2072, 4096, 14336), ("prefill b4 gate/up", 1036, 28672, 4096), ("prefill b4 down", 1036, 4096, 14336),
          ("deconv2 b8", 23328, 4608, 1152)]
only = None
if "--only" in sys.argv:
    only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
for name, M, N, K in shapes:
    if only is not None and name not in only:
        continue
    Ws = [torch.randn((N, K), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(2)]
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for W in Ws:
        ops.gemm(a, W, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        for W in Ws:
            ops.gemm(a, W, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"{name:22s} M={M:6d} N={N:6d} K={K:6d} {us:9.1f} us {2 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)
    del Ws, a, out
