import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/spatialrgpt_amd") else os.getcwd())
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
from spatialrgpt_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(259, 512, 4096), (259, 6144, 4096), (96, 128, 64), (97, 130, 200), (300, 1000, 1152), (259, 4096, 14336)]:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.03).to(torch.bfloat16)
    b = torch.randn((N,), device="cuda").to(torch.bfloat16)
    ref = (a.float() @ w.float().T + b.float())
    out = ops.gemm(a, w, b)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"M={M} N={N} K={K}: rel max err {err:.2e}", "OK" if err < 2e-2 else "BAD")
