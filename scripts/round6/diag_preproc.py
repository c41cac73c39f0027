"""which (mode, size) of the preprocessing fuzz differs between the device and the host path, and where"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from types import SimpleNamespace
from spatialrgpt_amd.mm_utils import SrgptImageProcessor, process_images, process_images_device, process_regions, process_regions_device
import PIL
print("Pillow", PIL.__version__)
rng = np.random.default_rng(1)
cases = [(137, 1277), (1078, 4), (36, 768), (4, 1078), (2, 2), (3, 500), (500, 3), (10, 10), (1300, 1300), (1080, 1920), (40, 1200), (5, 5), (8, 300)]
for size in (378, 384, 224):
    proc = SrgptImageProcessor(size=size)
    for mode in ("resize", "pad", None):
        cfg = SimpleNamespace(image_aspect_ratio=mode, image_processor=proc)
        for h, w in cases:
            im = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))]
            ref = process_images(im, proc, cfg)
            got = process_images_device(im, proc, cfg, device="cuda", dtype=torch.float32).cpu()
            d = (got - ref).abs()
            if float(d.max()) > 0:
                nz = torch.nonzero(d[0].amax(0) > 0)
                print(f"size {size} mode {mode} image {h}x{w}: max diff {float(d.max()):.4f} = {float(d.max()) * 127.5:.1f} codes, {int((d > 0).sum())} of {d.numel()} values, rows {int(nz[:,0].min())}..{int(nz[:,0].max())} cols {int(nz[:,1].min())}..{int(nz[:,1].max())}")
    for mode in ("resize", "pad"):
        cfg = SimpleNamespace(image_aspect_ratio=mode, image_processor=proc)
        for h, w in cases:
            mk = [(rng.random((h, w)) > 0.6).astype(np.uint8) * 255]
            ref = process_regions(mk, proc, cfg)
            got = process_regions_device(mk, proc, cfg, device="cuda", dtype=torch.float32).cpu()
            d = (got - ref).abs()
            if float(d.max()) > 0:
                print(f"size {size} MASK mode {mode} {h}x{w}: max diff {float(d.max()):.4f}, {int((d > 0).sum())} of {d.numel()} values")
print("done")
