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


def beam_search(first_logits: torch.Tensor, step, batch: int, num_beams: int, max_new_tokens: int, eos_token_ids: Optional[List[int]],
                pad_token_id: Optional[int], length_penalty: float = 1.0, early_stopping=False, stopping_criteria=None) -> torch.Tensor:
    """`generate(num_beams > 1, do_sample=False)` as the reference's callers can ask for it (`--num_beams`: eval_spatial.py:234,
    eval_region_cls.py:321, model_vqa.py:75; default 1): HF beam search over a decoder whose prompt was fed as `inputs_embeds`
    (llava_llama.py:212) -- so the id sequences start EMPTY, `max_length` counts new tokens only, and the length penalty divides
    by the number of generated tokens.  Semantics restated from transformers' GenerationMixin beam search (4.37.2's
    BeamSearchScorer with num_beam_groups 1, length_penalty 1.0, early_stopping False, one returned sequence; the vectorised form
    of the installed release) and pinned to the reference's own generate(num_beams=3) on the tiny model (tests/golden/beam_kat.npz):

      * every step adds log_softmax(logits) to the running beam scores, takes the K = max(2, 1 + #eos) * num_beams best (beam, token)
        continuations per batch row (beam 0 alone is live at the first step);
      * a continuation that ends in an EOS id (or reaches max_new_tokens, or satisfies a stopping criterion) leaves the running set;
        if it ranks among the first num_beams it becomes a FINISHED hypothesis scored sum_logprob / n_generated ** length_penalty,
        and the num_beams best finished hypotheses are kept;
      * the num_beams best remaining continuations run on (their KV-cache rows are gathered by `step`);
      * the loop ends when no running beam can beat the worst kept hypothesis (best running score / cur_len ** length_penalty), or
        nothing can continue; the best finished hypothesis of every row is returned, padded with pad_token_id (else the first EOS).

    first_logits: fp32 [batch * num_beams, V] (last prompt position; the num_beams rows of a batch item are replicas).
    step(tokens int64 [batch * num_beams], beam_idx int64 [batch * num_beams]) -> fp32 logits [batch * num_beams, V]: row i continues
    from the cache of (old) flat row beam_idx[i] and consumes tokens[i].
    Returns int64 [batch, <= max_new_tokens] (new tokens only, like the greedy path)."""
    dev = first_logits.device
    B, nb, G = batch, num_beams, max_new_tokens
    V = first_logits.shape[-1]
    eos = None if not eos_token_ids else torch.tensor(list(eos_token_ids), device=dev, dtype=torch.int64)
    K = max(2, 1 + (0 if eos is None else eos.numel())) * nb
    fill = pad_token_id if pad_token_id is not None else (int(eos[0]) if eos is not None else -1)
    NEG = -1.0e9
    run_seq = torch.full((B, nb, G), fill, dtype=torch.int64, device=dev)
    fin_seq = run_seq.clone()
    run_score = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    run_score[:, 1:] = NEG                       # replicas: only beam 0 may seed the first step
    fin_score = torch.full((B, nb), NEG, dtype=torch.float32, device=dev)
    fin_len = torch.zeros((B, nb), dtype=torch.int64, device=dev)
    fin_done = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    improvable = torch.ones((B, 1), dtype=torch.bool, device=dev)   # can a running beam still beat the kept hypotheses?
    head = torch.arange(K, device=dev) < nb
    rows = torch.arange(B, device=dev)[:, None] * nb
    logits = first_logits
    cur = 0

    def take(t, idx):                            # [B, n, ...] gathered along dim 1 by idx [B, m]
        while idx.dim() < t.dim():
            idx = idx.unsqueeze(-1)
        return torch.take_along_dim(t, idx, dim=1)

    while True:
        lp = torch.log_softmax(logits.float(), dim=-1).reshape(B, nb, V) + run_score[:, :, None]
        top_lp, top_ix = torch.topk(lp.reshape(B, nb * V), K)
        src_beam = top_ix // V
        cand = take(run_seq, src_beam).clone()
        cand[:, :, cur] = top_ix % V
        # which continuations stop here: an EOS id, the token budget, a caller's criterion (judged on the ids so far, per candidate)
        stops = torch.zeros((B, K), dtype=torch.bool, device=dev)
        if eos is not None:
            stops |= (cand[:, :, cur, None] == eos[None, None, :]).any(-1)
        if cur + 1 >= G:
            stops[:] = True
        if stopping_criteria:
            flat = cand[:, :, :cur + 1].reshape(B * K, cur + 1)
            for crit in stopping_criteria:
                r = crit(flat, None)
                r = r if isinstance(r, torch.Tensor) else torch.full((B * K,), bool(r), device=dev)
                stops |= r.to(dev).bool().reshape(-1).expand(B * K).reshape(B, K)
        # the num_beams best continuations that go on
        live_lp = top_lp + stops.float() * NEG
        nxt = torch.topk(live_lp, nb)[1]
        new_run_seq, new_run_score, new_src = take(cand, nxt), take(live_lp, nxt), take(src_beam, nxt)
        # finished hypotheses: only a stop among the first num_beams candidates counts
        just = stops & head[None, :]
        hyp = top_lp / float((cur + 1) ** length_penalty)
        hyp = hyp + (fin_done.all(-1, keepdim=True) & (early_stopping is True)).float() * NEG
        hyp = hyp + (~improvable).float() * NEG + (~just).float() * NEG
        m_score = torch.cat((fin_score, hyp), 1)
        keep = torch.topk(m_score, nb)[1]
        fin_seq = take(torch.cat((fin_seq, cand), 1), keep)
        fin_score = take(m_score, keep)
        fin_len = take(torch.cat((fin_len, torch.full((B, K), cur + 1, dtype=torch.int64, device=dev)), 1), keep)
        fin_done = take(torch.cat((fin_done, just), 1), keep)
        run_seq, run_score = new_run_seq, new_run_score
        cur += 1
        # can the best running beam still beat the worst kept hypothesis?  (early_stopping False: judged at the current length)
        best_len = (G if (early_stopping == "never" and length_penalty > 0.0) else cur)
        best_run = run_score[:, :1] / float(best_len ** length_penalty)
        worst = torch.where(fin_done, fin_score.min(dim=1, keepdim=True)[0], torch.full_like(fin_score, NEG))
        improvable = improvable & (best_run > worst).any(-1, keepdim=True)
        go_on = bool(improvable.any()) and not bool(fin_done.all() and early_stopping is True) and not bool(stops.all())
        if not go_on:
            break
        logits = step(run_seq[:, :, cur - 1].reshape(-1), (new_src + rows).reshape(-1))
    n = int(fin_len[:, 0].max())
    return fin_seq[:, 0, :max(n, 1)].contiguous()
