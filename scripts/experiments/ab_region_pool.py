"""A/B of the pooling launch: SRGPT_REGION_MFMA = 0 (VALU kernel) / 1, 2, 4 (MFMA kernel, that many chunks per wave) / 3 (by map size) -- tuning build.
Run under `rocprofv3 --kernel-trace` for the per-kernel durations; prints max |diff| against a torch fp32 restatement.
    SRGPT_REGION_MFMA=3 python scripts/experiments/ab_region_pool.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "spatialrgpt_amd", "libsrgpt_hip_tuning.so"))
from spatialrgpt_amd import ops
torch.manual_seed(0)
SHAPES = ((108, 8), (108, 16), (27, 8), (54, 3), (64, 5))
sel = os.environ.get("AB_SHAPE")
for fw, M in (SHAPES if sel is None else (SHAPES[int(sel)],)):
    feat = torch.randn((fw * fw, 1152), device="cuda").to(torch.bfloat16)
    masks = (torch.rand((M, 384, 384), device="cuda") > 0.5).to(torch.bfloat16)
    for _ in range(30):
        out = ops.region_pool(feat, masks)
    torch.cuda.synchronize()
    m = torch.nn.functional.interpolate(masks[None].float(), size=(fw, fw), mode="bilinear", align_corners=False)[0].to(torch.bfloat16)
    mf = m.reshape(M, -1)
    den = (mf.float().sum(1).to(torch.bfloat16) + 1e-8).to(torch.bfloat16)
    w = (mf.float() / den.float()[:, None]).to(torch.bfloat16).float()
    ref = w @ feat.float()
    d = (out.float() - ref).abs().max().item()
    print(f"mode {os.environ.get('SRGPT_REGION_MFMA', 'default')}  grid {fw}x{fw} M={M}: max|out - fp32 ref| = {d:.3e}  (|ref| max {ref.abs().max().item():.3e})")
