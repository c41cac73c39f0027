"""Per-phase timestamps of gemm256 (tuning build, variant 8): waves 0 (early half) and 4 (late half) of block 0, K tiles 8..11.
stamps per phase: 0 phase start, 1 fragment reads issued, 2 DMA issued + waits done (barrier entry), 3 barrier exit,
4 fragments landed (lgkmcnt 0), 5 MFMAs issued (barrier entry), 6 barrier exit."""
import ctypes, os, sys
os.environ["SRGPT_GEMM256_ABLATE"] = "8"
os.environ["SRGPT_GEMM_FORCE_256"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libsrgpt_hip.so", "libsrgpt_hip_tuning.so")
from spatialrgpt_amd import ops
M = N = K = 4096
a = torch.randn((M, K), device="cuda", dtype=torch.bfloat16)
w = torch.randn((N, K), device="cuda", dtype=torch.bfloat16) * 0.02
for _ in range(3):
    ops.gemm(a, w)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_uint64 * (2 * 4 * 4 * 8))()
assert lib.srgpt_gemm256_debug_ts(buf) == 0
t0 = min(x for x in buf if x)
names = ["start", "reads", "dma+wait", "bar1 out", "lgkm0", "mfma iss", "bar2 out"]
for g in range(2):
    print(f"--- wave {4 * g} ({'late' if g else 'early'} half)")
    for t in range(4):
        for p in range(4):
            base = ((g * 4 + t) * 4 + p) * 8
            st = [buf[base + i] - t0 for i in range(7)]
            d = [st[i + 1] - st[i] for i in range(6)]
            print(f"tile {8 + t} p{p}: start @{st[0]:7d}  " + "  ".join(f"{names[i + 1]}+{d[i]:4d}" for i in range(6)))
