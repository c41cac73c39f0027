"""phase stamps (s_memtime) of one key tile inside flash2_bf16_kernel at the SigLIP shape -- tuning build only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
from spatialrgpt_amd import ops
lib = C.CDLL(_lib.LIB_PATH)
names = ["step entry", "barrier passed", "next tile parked + requests issued", "QK^T products in registers", "softmax done", "PV issued (step end)"]
for B in (2, 16):
    qkv = (torch.randn((B, 729, 48, 72), device="cuda") * 0.7).to(torch.bfloat16)
    q, k, v = qkv[:, :, :16], qkv[:, :, 16:32], qkv[:, :, 32:]
    for _ in range(5):
        o = ops.attention(q, k, v)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert lib.srgpt_flash_debug_stamps(buf, 16) == 0
    v_ = list(buf)
    print(f"B = {B}: key tile 5 of block (0,0,0), wave 0; ticks")
    for i in range(1, 6):
        print(f"  {names[i]:38s} +{v_[i] - v_[i - 1]:6d}   (t = {v_[i] - v_[0]})")
