"""Host-side input contract of the hot path (reference: llava/mm_utils.py:477-617).  Stays Python/CPU like
the reference: prompt tokenisation with the <image> sentinel, image / mask preprocessing, stopping criteria."""
from __future__ import annotations

import copy
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
    """Minimal SigLIP-style processor (resize -> rescale 1/255 -> normalise) for when no HF processor is at hand."""

    def __init__(self, size=384, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), rescale_factor=1 / 255.0,
                 do_normalize=True, do_convert_rgb=True, resample=3):
        self.size = {"height": size, "width": size}
        self.image_mean, self.image_std = list(image_mean), list(image_std)
        self.rescale_factor, self.do_normalize, self.do_convert_rgb, self.resample = rescale_factor, do_normalize, do_convert_rgb, resample

    def preprocess(self, image, return_tensors="pt"):
        from PIL import Image

        if isinstance(image, np.ndarray):
            arr = image.astype(np.float32)
            if arr.ndim == 2:
                arr = arr[None]
            if arr.shape[-2:] != (self.size["height"], self.size["width"]):
                chans = [np.asarray(Image.fromarray(c, mode="F").resize((self.size["width"], self.size["height"]), self.resample))
                         for c in arr]
                arr = np.stack(chans, 0)
        else:
            if self.do_convert_rgb:
                image = image.convert("RGB")
            image = image.resize((self.size["width"], self.size["height"]), self.resample)
            arr = np.asarray(image).astype(np.float32).transpose(2, 0, 1)
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
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src = min(floor(dst * in/out), in - 1)."""
    h, w = m.shape
    ys = np.minimum((np.arange(out_h) * (h / out_h)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(out_w) * (w / out_w)).astype(np.int64), w - 1)
    return m[ys][:, xs]


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
