"""Host-side input contract of the hot path (reference: llava/mm_utils.py:477-617).  Stays Python/CPU like
the reference: prompt tokenisation with the <image> sentinel, image / mask preprocessing, stopping criteria."""
from __future__ import annotations

import copy
import functools
import math
from typing import List, Sequence

import numpy as np
import torch

from .constants import IMAGE_TOKEN_INDEX


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None, lstrip=False):
    """mm_utils.py:545-570: tokenise each "<image>"-separated chunk on its own, keep only the first BOS, join the
    chunks with the image sentinel."""
    chunks = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    ids: List[int] = []
    offset = 0
    if lstrip:
        offset = 1
    elif len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    pieces = []
    for i, c in enumerate(chunks):
        pieces.append(c)
        if i + 1 < len(chunks):
            pieces.append(sep)
    for i, x in enumerate(pieces):
        if i == 0 and lstrip:
            ids.extend(x)
        else:
            ids.extend(x[offset:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def get_model_name_from_path(model_path: str) -> str:
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria:
    """mm_utils.py:586-617 (behaviour kept as is, including the start_len quirk of SURVEY 9.11)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for keyword in keywords:
            cur = tokenizer(keyword).input_ids
            if len(cur) > 1 and cur[0] == tokenizer.bos_token_id:
                cur = cur[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(cur))
            self.keyword_ids.append(torch.tensor(cur))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids, scores, **kwargs) -> bool:
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        self.keyword_ids = [k.to(output_ids.device) for k in self.keyword_ids]
        for k in self.keyword_ids:
            if output_ids.shape[1] >= k.shape[0] and (output_ids[0, -k.shape[0]:] == k).all():
                return True
        outputs = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(keyword in outputs for keyword in self.keywords)

    def __call__(self, output_ids, scores, **kwargs) -> bool:
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))


class SrgptImageProcessor:
    """Built-in processor for when transformers cannot build its own (no torchvision here).  Two shapes:
      * SigLIP (siglip_encoder.py:10 `SiglipImageProcessor`): resize to size x size (bicubic) -> rescale 1/255 -> normalise
      * CLIP   (clip_encoder.py:11 `CLIPImageProcessor`; `center_crop=True`): resize the SHORTEST edge to `shortest_edge`
        keeping the aspect ratio (bicubic) -> centre crop size x size -> rescale -> normalise; exposes `crop_size`, which is how
        the reference's helpers tell the two apart (mm_utils.py:437-441)."""

    def __init__(self, size=384, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), rescale_factor=1 / 255.0,
                 do_normalize=True, do_convert_rgb=True, resample=3, shortest_edge=None, center_crop=False):
        self.center_crop = bool(center_crop)
        if self.center_crop:
            self.crop_size = {"height": size, "width": size}
            self.size = {"shortest_edge": shortest_edge or size}
        else:
            self.crop_size = None
            self.size = {"height": size, "width": size}
        self.image_mean, self.image_std = list(image_mean), list(image_std)
        self.rescale_factor, self.do_normalize, self.do_convert_rgb, self.resample = rescale_factor, do_normalize, do_convert_rgb, resample

    def _target(self, h, w):
        """-> ((resize_h, resize_w), (crop_h, crop_w) | None)"""
        if not self.center_crop:
            return (self.size["height"], self.size["width"]), None
        se = self.size["shortest_edge"]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = se, int(se * long / short)  # HF get_resize_output_image_size(default_to_square=False)
        rh, rw = (new_long, new_short) if w <= h else (new_short, new_long)
        return (rh, rw), (self.crop_size["height"], self.crop_size["width"])

    def preprocess(self, image, return_tensors="pt"):
        from PIL import Image

        if isinstance(image, np.ndarray):
            src = image[None] if image.ndim == 2 else image
            (rh, rw), crop = self._target(src.shape[-2], src.shape[-1])
            if src.shape[-2:] != (rh, rw):
                if src.dtype == np.uint8:
                    # HF converts a uint8 array to a PIL image and resizes THERE: 8 bits per channel, rounded and clipped after
                    # each pass (process_regions' pad mode feeds [1, side, side] uint8 masks, mm_utils.py:505-531) -- pinned to
                    # transformers' own SiglipImageProcessor by tests/test_host_logic.py
                    chans = [np.asarray(Image.fromarray(np.ascontiguousarray(c), mode="L").resize((rw, rh), self.resample)) for c in src]
                else:
                    chans = [np.asarray(Image.fromarray(c.astype(np.float32), mode="F").resize((rw, rh), self.resample)) for c in src]
                src = np.stack(chans, 0)
            arr = src.astype(np.float32)
        else:
            if self.do_convert_rgb:
                image = image.convert("RGB")
            (rh, rw), crop = self._target(image.size[1], image.size[0])
            if (image.size[1], image.size[0]) != (rh, rw):
                image = image.resize((rw, rh), self.resample)
            arr = np.asarray(image).astype(np.float32).transpose(2, 0, 1)
        if crop is not None:
            ch, cw = crop
            top, left = (arr.shape[-2] - ch) // 2, (arr.shape[-1] - cw) // 2
            if top < 0 or left < 0:
                raise ValueError("centre crop larger than the resized image")
            arr = arr[..., top:top + ch, left:left + cw]
        arr = arr * self.rescale_factor
        if self.do_normalize:
            m = np.asarray(self.image_mean, np.float32)[:, None, None]
            s = np.asarray(self.image_std, np.float32)[:, None, None]
            arr = (arr - m) / s
        return {"pixel_values": [torch.from_numpy(np.ascontiguousarray(arr))]}


def _crop_size(image_processor):
    if hasattr(image_processor, "crop_size") and image_processor.crop_size is not None:
        return image_processor.crop_size  # CLIP tower
    assert hasattr(image_processor, "size")
    return image_processor.size  # SigLIP tower


def process_image(image, data_args, image_folder=None):
    """mm_utils.py:421-474 for PIL inputs."""
    processor = data_args.image_processor
    image = image.convert("RGB")
    if data_args.image_aspect_ratio == "resize":
        cs = _crop_size(processor)
        image = image.resize((cs["height"], cs["width"]))
    if data_args.image_aspect_ratio == "pad":
        from PIL import Image

        w, h = image.size
        if w != h:
            side = max(w, h)
            bg = Image.new(image.mode, (side, side), tuple(int(x * 255) for x in processor.image_mean))
            bg.paste(image, ((side - w) // 2, (side - h) // 2))
            image = bg
    return processor.preprocess(image, return_tensors="pt")["pixel_values"][0]


def process_images(images, image_processor, model_cfg):
    """mm_utils.py:535-542."""
    model_cfg.image_processor = image_processor
    new_images = [process_image(im, model_cfg, None) for im in images]
    if all(x.shape == new_images[0].shape for x in new_images):
        new_images = torch.stack(new_images, dim=0)
    return new_images


def _nearest_resize(m: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(m, (out_w, out_h), interpolation=cv2.INTER_NEAREST) (mm_utils.py:520): OpenCV's resizeNN gathers
    src[min(floor(dst * (1 / (out / in))), in - 1)] -- the RECIPROCAL of the rounded scale, not in / out: the two differ at exact
    multiples for many sizes (72 -> 224, 76 -> 336, ...; tests/golden/cv2_nearest_kat.json).  Same tables as the device path."""
    h, w = m.shape
    return m[cv2_nearest_index(h, out_h)][:, cv2_nearest_index(w, out_w)]


def process_regions(masks: Sequence[np.ndarray], image_processor, data_args):
    """mm_utils.py:477-532: uint8 [H,W] masks -> float [M, S, S] in processor geometry."""
    mp = copy.deepcopy(image_processor)
    mp.do_normalize = False
    mp.do_convert_rgb = False
    mp.rescale_factor = 1.0
    out = []
    for m in masks:
        m = np.asarray(m)
        if data_args.image_aspect_ratio == "resize":
            cs = _crop_size(data_args.image_processor if hasattr(data_args, "image_processor") else image_processor)
            m = _nearest_resize(m, cs["height"], cs["width"])
        if data_args.image_aspect_ratio == "pad":
            H, W = m.shape
            side = max(H, W)
            p = np.zeros((side, side), dtype=np.uint8)
            p[(side - H) // 2:(side - H) // 2 + H, (side - W) // 2:(side - W) // 2 + W] = m
            m = p
        out.append(mp.preprocess(m[None, ...], return_tensors="pt")["pixel_values"][0])
    return torch.vstack([torch.as_tensor(o) for o in out]).float()


# ------------------------------------------------------------------------------------------------
# Device-side request preprocessing: raw uint8 image / masks in, model-ready tensors on the GPU out.
# The host computes only the small resampling tables; pixels are touched by the HIP kernels (csrc/preproc.hip).
# ------------------------------------------------------------------------------------------------
def _bicubic(x: float, a: float = -0.5) -> float:
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=64)
def pil_bicubic_tables(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (libImaging/Resample.c), restated in
    double exactly in Pillow's operation order: -> (bounds int32 [out, 2] = (first tap, taps), coef int32 [out, ksize]).
    A dimension that is not resized is skipped by Pillow: identity table."""
    if in_size == out_size:
        b = np.stack([np.arange(out_size, dtype=np.int32), np.ones(out_size, dtype=np.int32)], 1)
        return np.ascontiguousarray(b), np.full((out_size, 1), 1 << 22, dtype=np.int32)
    scale = in_size / out_size
    fs = scale if scale >= 1.0 else 1.0
    support = 2.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / fs
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                k[x] /= ww
        for x in range(ksize):
            v = k[x]
            coef[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, coef


def _pil_resize_u8_restated(arr: np.ndarray, out_h: int, out_w: int, vertical_first: bool = False) -> np.ndarray:
    """Pillow's two-pass 8-bit bicubic resize of a uint8 [H, W, C] array as integer arithmetic on pil_bicubic_tables (what the device
    kernels run), in either pass order.  Host-side check / calibration only (numpy, slow)."""
    def one_pass(a, axis, out_size):
        bounds, coef = pil_bicubic_tables(a.shape[axis], out_size)
        a = np.moveaxis(a, axis, 0).astype(np.int64)
        out = np.empty((out_size,) + a.shape[1:], np.uint8)
        for i in range(out_size):
            x0, n = int(bounds[i, 0]), int(bounds[i, 1])
            acc = (1 << 21) + np.tensordot(coef[i, :n].astype(np.int64), a[x0:x0 + n], axes=(0, 0))
            out[i] = np.clip(acc >> 22, 0, 255)
        return np.moveaxis(out, 0, axis)
    if vertical_first:
        return one_pass(one_pass(arr, 0, out_h), 1, out_w)
    return one_pass(one_pass(arr, 1, out_w), 0, out_h)


@functools.lru_cache(maxsize=1)
def pil_resizes_tall_images_vertically_first() -> bool:
    """Image.resize of recent Pillow releases (seen in 12.2: `if self.size[1] > self.size[0] * 100 and size[1] < self.size[1]`) shrinks
    an image more than 100 times taller than wide in two separate resizes, rows first -- the intermediate rounding then happens in
    the other order and the pixels differ from the horizontal-first result by up to ~20 codes.  The release that introduced it is
    not pinned anywhere (the reference does not pin Pillow), so the INSTALLED library is asked once with a 303 x 3 probe image."""
    from PIL import Image
    arr = ((np.arange(303 * 3 * 3, dtype=np.int64) * 2654435761) % 251).astype(np.uint8).reshape(303, 3, 3)
    ref = np.asarray(Image.fromarray(arr).resize((7, 101), Image.BICUBIC))
    if np.array_equal(ref, _pil_resize_u8_restated(arr, 101, 7, vertical_first=False)):
        return False
    if np.array_equal(ref, _pil_resize_u8_restated(arr, 101, 7, vertical_first=True)):
        return True
    raise RuntimeError("the installed Pillow resizes a 303 x 3 image like neither pass order of its published two-pass resampler; "
                       "process_images_device cannot promise the host path's pixels for such shapes")


@functools.lru_cache(maxsize=64)
def cv2_nearest_index(in_size: int, out_size: int):
    """cv2.resize(..., INTER_NEAREST) source index per destination index: min(floor(dst * (1 / (out / in))), in - 1),
    in double like cv2's resizeNN."""
    ifx = 1.0 / (out_size / in_size)
    return np.minimum(np.floor(np.arange(out_size, dtype=np.float64) * ifx).astype(np.int64), in_size - 1).astype(np.int32)


def _dev_tables(tables, device):
    return [torch.from_numpy(np.ascontiguousarray(t)).to(device) for t in tables]


def process_images_device(images, image_processor, model_cfg, device="cuda", dtype=torch.bfloat16):
    """Device counterpart of process_images (mm_utils.py:535-542 / process_image :421-474) for PIL images or uint8 HWC
    arrays: returns [N, 3, S, S] on the GPU in `dtype`, bit-identical to the host path followed by `.to(dtype)`.
    Aspect modes "resize" and "pad" (expand2square with the mean colour) and the processor default (plain resize)."""
    from . import _lib as L, ops

    cs = _crop_size(image_processor)
    S_h, S_w = cs["height"], cs["width"]
    mean = torch.tensor(list(image_processor.image_mean), dtype=torch.float32, device=device)
    std = torch.tensor(list(image_processor.image_std), dtype=torch.float32, device=device)
    lib = L.load()
    outs = []
    for im in images:
        arr = np.asarray(im.convert("RGB")) if hasattr(im, "convert") else np.asarray(im)
        if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
            raise ValueError("process_images_device: expected PIL images or uint8 [H, W, 3] arrays")
        if getattr(model_cfg, "image_aspect_ratio", None) == "pad" and arr.shape[0] != arr.shape[1]:
            h, w = arr.shape[:2]
            side = max(h, w)
            bg = np.empty((side, side, 3), np.uint8)
            bg[:] = np.asarray([int(x * 255) for x in image_processor.image_mean], np.uint8)
            bg[(side - h) // 2:(side - h) // 2 + h, (side - w) // 2:(side - w) // 2 + w] = arr
            arr = bg
        H, W = arr.shape[:2]
        # an image more than 100 times taller than wide that shrinks vertically: recent Pillow resizes its rows FIRST (see
        # pil_resizes_tall_images_vertically_first).  Same kernels on the transposed image -- their first pass then runs along the
        # original rows' axis -- and the result transposed back
        rows_first = H > 100 * W and S_h < H and pil_resizes_tall_images_vertically_first()
        if rows_first:
            arr, H, W, S_h, S_w = arr.transpose(1, 0, 2), W, H, S_w, S_h
        src = torch.from_numpy(np.array(arr, dtype=np.uint8, order="C")).to(device)  # PIL buffers are read-only: copy
        hb, hc = _dev_tables(pil_bicubic_tables(W, S_w), device)
        vb, vc = _dev_tables(pil_bicubic_tables(H, S_h), device)
        tmp = torch.empty((H, S_w, 3), dtype=torch.uint8, device=device)
        out = torch.empty((3, S_h, S_w), dtype=dtype, device=device)
        L.check(lib.srgpt_image_resize_normalize(src.data_ptr(), H, W, 3, hb.data_ptr(), hc.data_ptr(), hc.shape[1], vb.data_ptr(),
                                                 vc.data_ptr(), vc.shape[1], S_h, S_w, tmp.data_ptr(), out.data_ptr(),
                                                 mean.data_ptr(), std.data_ptr(), float(image_processor.rescale_factor),
                                                 int(bool(getattr(image_processor, "do_normalize", True))), ops.dt_code(out),
                                                 ops._stream()))
        if rows_first:
            out, S_h, S_w = out.transpose(1, 2).contiguous(), S_w, S_h
        outs.append(out)
    return torch.stack(outs, 0)


def process_regions_device(masks: Sequence[np.ndarray], image_processor, data_args, device="cuda", dtype=torch.bfloat16):
    """Device counterpart of process_regions (mm_utils.py:477-532): uint8 [H, W] masks -> [M, S, S] on the GPU in `dtype`,
    bit-identical to the host path followed by `.to(dtype)`.
      image_aspect_ratio == "resize" (the SpatialRGPT configuration): cv2.INTER_NEAREST gather to the processor size (values are
        the mask's own 0/1 or 0/255);
      image_aspect_ratio == "pad": pad_to_square (zeros, centred) + the processor's Pillow-bicubic resize of the one-channel
        square, fused (the square is never materialised)."""
    from . import _lib as L, ops

    mode = getattr(data_args, "image_aspect_ratio", None)
    if mode not in ("resize", "pad"):
        raise NotImplementedError(f"process_regions_device: image_aspect_ratio {mode!r} (the reference handles 'resize' and 'pad')")
    cs = _crop_size(data_args.image_processor if hasattr(data_args, "image_processor") else image_processor)
    S_h, S_w = cs["height"], cs["width"]
    ms = [np.asarray(m) for m in masks]
    if not ms or any(m.ndim != 2 or m.dtype != np.uint8 or m.shape != ms[0].shape for m in ms):
        raise ValueError("process_regions_device: expected equally sized uint8 [H, W] masks")
    src = torch.from_numpy(np.ascontiguousarray(np.stack(ms, 0))).to(device)
    return (masks_pad_resize_device if mode == "pad" else masks_nearest_device)(src, S_h, S_w, dtype)


def masks_nearest_device(src: torch.Tensor, S_h: int, S_w: int, dtype) -> torch.Tensor:
    """uint8 [K, H, W] on the GPU -> [K, S_h, S_w]: the cv2.INTER_NEAREST step of process_regions ("resize" mode)."""
    from . import _lib as L, ops

    K, H, W = src.shape
    ys, xs = _dev_tables((cv2_nearest_index(H, S_h), cv2_nearest_index(W, S_w)), src.device)
    out = torch.empty((K, S_h, S_w), dtype=dtype, device=src.device)
    L.check(L.load().srgpt_mask_resize_nearest(src.data_ptr(), K, H, W, ys.data_ptr(), xs.data_ptr(), S_h, S_w, out.data_ptr(),
                                               ops.dt_code(out), ops._stream()))
    return out


def masks_pad_resize_device(src: torch.Tensor, S_h: int, S_w: int, dtype) -> torch.Tensor:
    """uint8 [K, H, W] on the GPU -> [K, S_h, S_w]: pad_to_square + the processor's bicubic resize ("pad" mode, mm_utils.py:505-531)."""
    from . import _lib as L, ops

    K, H, W = src.shape
    side = max(H, W)
    hb, hc = _dev_tables(pil_bicubic_tables(side, S_w), src.device)
    vb, vc = _dev_tables(pil_bicubic_tables(side, S_h), src.device)
    tmp = torch.empty((K, side, S_w), dtype=torch.uint8, device=src.device)
    out = torch.empty((K, S_h, S_w), dtype=dtype, device=src.device)
    L.check(L.load().srgpt_mask_pad_resize(src.contiguous().data_ptr(), K, H, W, hb.data_ptr(), hc.data_ptr(), hc.shape[1], vb.data_ptr(),
                                           vc.data_ptr(), vc.shape[1], S_h, S_w, tmp.data_ptr(), out.data_ptr(), ops.dt_code(out),
                                           ops._stream()))
    return out
