"""phase stamps (s_memtime, shader cycles) of block 0 / wave 0 of the skinny decode kernel for the four projections of a layer.
   python scripts/ubench_skinny_stamps.py [batch] [bf16|fp8] [pub]     -- tuning build only
   pub: the row-statistics hand-off of round 5 (srgpt_gemv_rowss: the RMSNorm products read a published table, the residual products publish one)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
from spatialrgpt_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fp8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"
pub = len(sys.argv) > 3 and sys.argv[3] == "pub"
lib = C.CDLL(_lib.LIB_PATH)
names = ["entry", "statistics / first stage requested", "first slice done (NSU stages)", "K loop done", "barrier", "reduced + stored"]
for name, N, K, norm, res, sw in [("qkv", 6144, 4096, 1, 0, 0), ("o", 4096, 4096, 0, 1, 0), ("gate/up", 14336, 4096, 1, 0, 1), ("down", 4096, 14336, 0, 1, 0)]:
    x = torch.randn((B, K), device="cuda").to(torch.bfloat16)
    g = torch.ones((K,), device="cuda", dtype=torch.bfloat16) if norm else None
    r = torch.randn((B, N), device="cuda").to(torch.bfloat16) if res else None
    ws = [torch.randn((N * (2 if sw else 1), K), device="cuda").to(torch.bfloat16) * 0.02 for _ in range(6)]
    if fp8:
        qs = [ops.quantize_fp8_rows(w)[:2] for w in ws]
    table = None
    if pub and norm:
        table = torch.zeros((B, _lib.ROWSS_STRIDE), device="cuda")
        table[:, 0] = x.float().pow(2).sum(-1)
    for rep in range(2):
        for i in range(6):  # cold weights each time
            if pub:
                kw = dict(w8=qs[i][0], wscale=qs[i][1]) if fp8 else dict(w=ws[i])
                ops.gemv_rowss(x, norm_w=g, eps=1e-5, residual=r, swiglu=bool(sw), rowss_in=table, publish=bool(res), **kw)
            elif fp8:
                ops.gemv_w8(x, qs[i][0], qs[i][1], norm_w=g, eps=1e-5, residual=r, swiglu=bool(sw))
            else:
                ops.gemv(x, ws[i], norm_w=g, eps=1e-5, residual=r, swiglu=bool(sw))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert lib.srgpt_skinny_debug_stamps(buf, 16) == 0
    v = list(buf)
    print(f"{name} (batch {B}, {'fp8' if fp8 else 'bf16'}{', published statistics' if pub else ''}): " + " | ".join(f"{names[i]} +{v[i] - v[i-1]}" for i in range(1, 6)) + f" | total {v[5] - v[0]}")
