"""Shared helpers for the test-suite: golden-fixture loading and comparison."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _t(a, dtype):
    """npz array -> torch tensor (uint16 arrays hold raw bf16 bits)."""
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t


def load_tiny(name):
    """-> (cfg_dict, dtype, weights{name: tensor}, inputs dict, ref dict)"""
    z = np.load(os.path.join(GOLD, name))
    cfg = json.loads(bytes(z["cfg_json"]).decode())
    dtype = {"torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16}[bytes(z["dtype"]).decode()]
    w, ref = {}, {}
    for k in z.files:
        if k.startswith("w."):
            w[k[2:]] = _t(z[k], dtype)
        elif k.startswith("ref."):
            ref[k[4:]] = _t(z[k], dtype)
    images = (torch.from_numpy(z["in.images_q32"].astype(np.float32)) / 32).to(dtype)
    d1 = torch.from_numpy(z["in.depths_q32"].astype(np.float32)) / 32
    depths = d1.expand(-1, 3, -1, -1).contiguous().to(dtype)
    masks = [torch.from_numpy(z["in.masks_u8"][i].astype(np.float32)).to(dtype) for i in range(z["in.masks_u8"].shape[0])]
    # masks stacked as [n_img=1 stacked...]: stored shape [1, M, S, S]
    inputs = dict(input_ids=torch.from_numpy(z["in.input_ids"]), images=images, depths=depths, masks=masks)
    return cfg, dtype, w, inputs, ref


def load_kat():
    return np.load(os.path.join(GOLD, "region_kat.npz"))


def max_err(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def assert_close(a, b, atol, rtol=0.0, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance, "
                                 f"max err {err.max().item():.4e} (ref max {b.abs().max().item():.4e})")
