"""phase stamps (s_memtime) inside region_pool_kernel at the headline shape (108 x 108 x 1152 bf16, 8 masks) and at the depth shape
(27 x 27) -- tuning build only.   python scripts/experiments/ubench_region_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
from spatialrgpt_amd import ops
lib = C.CDLL(_lib.LIB_PATH)
names = ["entry", "first rows requested", "denominators", "weights in LDS", "rows consumed (FMA)", "in-block reduce", "partials issued",
         "stores drained + barrier", "ticket drawn"]
for fw in (108, 54, 27):
    feat = torch.randn((fw * fw, 1152), device="cuda").to(torch.bfloat16)
    masks = (torch.rand((8, 384, 384), device="cuda") > 0.5).to(torch.bfloat16)
    for _ in range(5):
        out = ops.region_pool(feat, masks)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        out = ops.region_pool(feat, masks)
    e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    assert lib.srgpt_region_debug_stamps(buf, 32) == 0
    v = list(buf)
    print(f"feature grid {fw} x {fw}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per region_pool call (2 launches + torch allocs); block (0,0), cycles since entry")
    for i in range(1, 9):
        print(f"  {names[i]:28s} +{v[i] - v[i - 1]:7d}   (t = {v[i] - v[0]})")
    print(f"  last arriver of channel slab 0: slab sums + store +{v[17] - v[16]:7d}")
