"""Generation-config resolution and the logits warpers of the sampling path.

The reference decodes through `self.llm.generate(inputs_embeds=..., **generation_kwargs)` (llava_llama.py:212) on an LLM built by
`AutoModelForCausalLM.from_pretrained(<ckpt>/llm)` (language_model/builder.py:41-59), i.e. HF `GenerationMixin.generate`
(transformers==4.37.2, pyproject.toml:17) with

  * `llm.generation_config` = `<ckpt>/llm/generation_config.json` when the file exists, else `GenerationConfig.from_model_config`
    of `<ckpt>/llm/config.json` (bos / eos / pad and any generation field stored there),
  * every keyword of the call written over it -- INCLUDING explicit `None`s (`generation_config.update(**kwargs)`:
    eval_spatial.py:224-236 passes `top_p=None`, which switches the top-p warper off),
  * `eos_token_id` an int or a LIST (Llama-3: [128001, 128009]); `pad_token_id` defaulting to the first EOS id,
  * sampling warpers in the order temperature -> top-k -> top-p (`_get_logits_warper`), each only when it would change anything
    (temperature != 1, top_k != 0, top_p < 1); `top_k` defaults to 50 in 4.37.2.

`resolve_generation` reproduces that resolution; `warp_logits` is the three warpers on plain torch tensors (pinned against HF's own
`TemperatureLogitsWarper` / `TopKLogitsWarper` / `TopPLogitsWarper` on CPU by tests/test_host_generation.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

# GenerationConfig defaults of the pinned transformers==4.37.2 (generation/configuration_utils.py); newer releases moved to
# `None` placeholders, the reference's behaviour is the pinned one
HF_DEFAULTS = dict(do_sample=False, temperature=1.0, top_k=50, top_p=1.0, max_length=20, max_new_tokens=None, min_new_tokens=None,
                   num_beams=1, eos_token_id=None, pad_token_id=None, bos_token_id=None)

_GEN_KEYS = tuple(HF_DEFAULTS)

NOT_GIVEN = object()  # a keyword the caller did not pass (an explicit None is a value: it overrides the stored config)


def generation_config_from_files(llm_config: dict, generation_json: Optional[dict]) -> dict:
    """what `llm.generation_config` holds after `from_pretrained`: the json when present, else the generation fields of the
    model config (`GenerationConfig.from_model_config`)."""
    src = generation_json if generation_json is not None else llm_config
    return {k: src[k] for k in _GEN_KEYS if k in src and src[k] is not None}


@dataclass
class ResolvedGeneration:
    do_sample: bool
    temperature: Optional[float]
    top_k: Optional[int]
    top_p: Optional[float]
    max_new_tokens: int
    min_new_tokens: Optional[int]
    num_beams: int
    eos_token_ids: Optional[List[int]]  # None = never stop on a token; order preserved (pad falls back to the FIRST)
    pad_token_id: Optional[int]


def _eos_list(eos) -> Optional[List[int]]:
    if eos is None:
        return None
    if isinstance(eos, torch.Tensor):
        eos = eos.tolist()
    if isinstance(eos, (list, tuple, set)):
        out = [int(e) for e in eos]
        return out or None
    return [int(eos)]


def resolve_generation(stored: Optional[dict], **kwargs) -> ResolvedGeneration:
    """stored = the checkpoint's generation config (dict, possibly empty); kwargs = the keywords of the generate() call, NOT_GIVEN
    for the ones the caller left out."""
    cfg = dict(HF_DEFAULTS)
    cfg.update(stored or {})
    for k, v in kwargs.items():
        if v is not NOT_GIVEN and k in cfg:
            cfg[k] = v
    if cfg["max_new_tokens"] is not None:
        max_new = int(cfg["max_new_tokens"])
    else:
        # generate() fed `inputs_embeds` only: max_length counts new tokens alone (there are no input ids)
        max_new = max(1, int(cfg["max_length"] if cfg["max_length"] is not None else HF_DEFAULTS["max_length"]))
    eos = _eos_list(cfg["eos_token_id"])
    pad = cfg["pad_token_id"]
    if pad is None and eos is not None:
        pad = eos[0]  # "Setting `pad_token_id` to `eos_token_id`" (first element of a list)
    mn = cfg["min_new_tokens"]
    return ResolvedGeneration(do_sample=bool(cfg["do_sample"]), temperature=cfg["temperature"], top_k=cfg["top_k"], top_p=cfg["top_p"],
                              max_new_tokens=max_new, min_new_tokens=None if mn is None else int(mn),
                              num_beams=int(cfg["num_beams"] or 1), eos_token_ids=eos, pad_token_id=None if pad is None else int(pad))


def warp_logits(logits: torch.Tensor, temperature=None, top_k=None, top_p=None, min_tokens_to_keep: int = 1) -> torch.Tensor:
    """[B, V] float logits -> warped logits (filtered entries = -inf), HF's order and rules:
    temperature (when not None and != 1) -> top-k (when not None and != 0) -> top-p (when not None and < 1)."""
    scores = logits
    if temperature is not None and temperature != 1.0:
        scores = scores / temperature
    if top_k is not None and top_k != 0:
        k = min(max(int(top_k), min_tokens_to_keep), scores.shape[-1])
        kth = torch.topk(scores, k, dim=-1).values[..., -1, None]
        scores = scores.masked_fill(scores < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sl, si = torch.sort(scores, descending=False, dim=-1)
        cp = sl.softmax(dim=-1).cumsum(dim=-1)
        rm = cp <= (1 - top_p)
        rm[..., -min_tokens_to_keep:] = False
        scores = scores.masked_fill(rm.scatter(-1, si, rm), float("-inf"))
    return scores
