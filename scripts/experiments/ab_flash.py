"""Timing shapes of the MFMA flash attention kernel (flash.hip): prints a checksum of every output and runs each shape 30 times
for `rocprofv3 --kernel-trace` (scripts/experiments/ab_flash.sh).  The round-3 / round-4 A/B in profiles/r04_flash_attention.txt was taken with
this script while both kernels were in the tuning build (knob SRGPT_FLASH_V2, gone with the round-3 kernel)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "spatialrgpt_amd", "libsrgpt_hip_tuning.so"))
from spatialrgpt_amd import ops
torch.manual_seed(0)
# (B, Tq, Tk, Hq, Hkv, D, causal): SigLIP so400m (2 passes as one batch), 8 requests' worth, CLIP-L/336, Llama-3 prefill, long prefill
SHAPES = ((2, 729, 729, 16, 16, 72, 0), (16, 729, 729, 16, 16, 72, 0), (2, 577, 577, 16, 16, 64, 0), (1, 259, 259, 32, 8, 128, 1),
          (1, 2048, 2048, 32, 8, 128, 1), (3, 100, 333, 8, 2, 32, 1))
sel = os.environ.get("AB_SHAPE")
for B, Tq, Tk, Hq, Hkv, D, causal in (SHAPES if sel is None else (SHAPES[int(sel)],)):
    qkv = (torch.randn((B, Tk, Hq + 2 * Hkv, D), device="cuda") * 0.7).to(torch.bfloat16)
    q, k, v = qkv[:, Tk - Tq:, :Hq], qkv[:, :, Hq:Hq + Hkv], qkv[:, :, Hq + Hkv:]
    kv_len = torch.tensor([Tk - 7 * i for i in range(B)], device="cuda", dtype=torch.int32) if B == 3 else None
    for _ in range(30):
        o = ops.attention(q, k, v, causal=bool(causal), kv_len=kv_len)
    torch.cuda.synchronize()
    hx = hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    print(f"B={B} Tq={Tq} Tk={Tk} Hq={Hq} Hkv={Hkv} D={D} causal={causal}: sha256 {hx}  nan={bool(torch.isnan(o.float()).any())}")
