"""A forced kernel variant of srgpt_gemm (tuning build: SRGPT_GEMM_FORCE_BM / _NBUF / _SPLITS / _256 in the environment) against
the product library on the same inputs: the variants only change tiling / pipelining, so un-split results must be BIT-identical
and split-K results equal up to the fp32 slab order (reported as max |diff| in bf16 ulps of the largest output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from spatialrgpt_amd import _lib, ops

prod = ops.gemm
shapes = [(259, 512, 4096), (259, 6144, 4096), (96, 128, 64), (97, 130, 200), (300, 1000, 1152), (259, 4096, 14336), (1458, 4304, 1152),
          (1458, 1152, 4304), (64, 256, 128), (259, 28672, 4096), (5, 130, 72)]
torch.manual_seed(0)
cases = []
for (M, N, K) in shapes:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.03).to(torch.bfloat16)
    b = torch.randn((N,), device="cuda").to(torch.bfloat16)
    r = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    cases.append((a, w, b, r, prod(a, w, b, residual=r).clone()))
# second library instance: the tuning build with the knobs of the environment
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
_lib._lib = None  # drop the cached handle: the next ops call loads the tuning build
bad = 0
for (a, w, b, r, ref) in cases:
    out = ops.gemm(a, w, b, residual=r)
    d = (out.float() - ref.float()).abs().max().item()
    same = torch.equal(out, ref)
    print(f"M={a.shape[0]} N={w.shape[0]} K={a.shape[1]}: bit-identical {same}, max |diff| {d:.3e} (max |out| {ref.float().abs().max().item():.2f})")
    bad += 0 if (same or d <= 2 ** -6 * ref.float().abs().max().item()) else 1
print("VARIANT_CHECK", "OK" if bad == 0 else f"BAD ({bad})", {k: v for k, v in os.environ.items() if k.startswith("SRGPT_GEMM")})
