"""flash attention (srgpt_attention) at the ViT and prefill shapes, us per call (graph-free, events around 20 back-to-back calls).
  SRGPT_LIB=<libsrgpt_hip*.so> python scripts/experiments/ubench_attention.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
if os.environ.get("SRGPT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SRGPT_LIB"])
from spatialrgpt_amd import ops

def run(name, B, T, Hq, Hkv, D, causal):
    QW = (Hq + 2 * Hkv) * D
    qkv = torch.randn((B, T, QW), device="cuda").to(torch.bfloat16)
    q = qkv[:, :, :Hq * D].view(B, T, Hq, D)
    k = qkv[:, :, Hq * D:(Hq + Hkv) * D].view(B, T, Hkv, D)
    v = qkv[:, :, (Hq + Hkv) * D:].view(B, T, Hkv, D)
    for _ in range(3):
        ops.attention(q, k, v, causal=causal)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            ops.attention(q, k, v, causal=causal)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    fl = 4.0 * B * Hq * T * T * D * (0.5 if causal else 1.0)
    print(f"{os.path.basename(_lib.LIB_PATH):28s} {name:34s} {best:8.2f} us  {fl / best / 1e6:7.1f} TF/s", flush=True)

run("ViT SigLIP 2 img 729x16x72", 2, 729, 16, 16, 72, False)
run("ViT SigLIP 16 img", 16, 729, 16, 16, 72, False)
run("ViT CLIP-336 2 img 577x16x64", 2, 577, 16, 16, 64, False)
run("prefill T=259 32/8 x128 causal", 1, 259, 32, 8, 128, True)
run("prefill 8 x T=259 causal", 8, 259, 32, 8, 128, True)
run("prefill T=707 32/32 x128 causal", 1, 707, 32, 32, 128, True)
