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


def beam_search_v5(first_logits: torch.Tensor, step, batch: int, num_beams: int, max_new_tokens: int, eos_token_ids: Optional[List[int]],
                   pad_token_id: Optional[int], length_penalty: float = 1.0, early_stopping=False, stopping_criteria=None) -> torch.Tensor:
    """The VECTORISED beam search of the installed transformers (5.x) -- kept because tests/golden/beam_kat.npz could only be minted
    with the installed release (the reference's generate(num_beams=3) run here); the product path is `beam_generate` below, which
    follows the reference's PINNED release (4.37.2: BeamSearchScorer) and agrees with this one on the golden fixtures.

    `generate(num_beams > 1, do_sample=False)` as the reference's callers can ask for it (`--num_beams`: eval_spatial.py:234,
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


beam_search = beam_search_v5  # the name the golden-minting script and the CPU pin use


# ------------------------------------------------------------------------------------------------------------------------------
# transformers==4.37.2 (the reference's pin, pyproject.toml:17): BeamSearchScorer / BeamHypotheses, GenerationMixin.beam_search and
# GenerationMixin.beam_sample, restated for a decoder whose prompt went in as `inputs_embeds` (llava_llama.py:212): input_ids start
# EMPTY, decoder_prompt_len = 0, max_length = max_new_tokens.  num_beam_groups 1, one returned sequence per batch item.
# Differences from the vectorised 5.x form above that matter to a caller:
#   * a batch item is done when its WORST kept hypothesis is at least as good as the best of ALL 2 x num_beams candidates of the step
#     (EOS candidates included) at the current length -- 5.x compares against the best beam that keeps running;
#   * the returned row holds the hypothesis WITHOUT the token that ended it, followed by eos_token_id[0] when there is room --
#     whichever EOS id of a list fired (Llama-3 lists two);
#   * stopping criteria end the whole search (`if beam_scorer.is_done or stopping_criteria(input_ids, scores): break`), the beams
#     still running are then entered as hypotheses at their current length.
# ------------------------------------------------------------------------------------------------------------------------------
class _BeamHypotheses:
    """transformers 4.37.2 generation/beam_search.py: BeamHypotheses (n-best list of one batch item)"""

    def __init__(self, num_beams: int, length_penalty: float, early_stopping, max_length: Optional[int]):
        self.num_beams, self.length_penalty, self.early_stopping, self.max_length = num_beams, length_penalty, early_stopping, max_length
        self.beams: list = []  # (score, token list)
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: List[int], sum_logprobs: float, generated_len: int):
        score = sum_logprobs / (generated_len ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                order = sorted([(s_, i) for i, (s_, _) in enumerate(self.beams)])
                del self.beams[order[0][1]]
                self.worst_score = order[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self) < self.num_beams:
            return False
        if self.early_stopping is True:
            return True
        if self.early_stopping is False:
            return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty
        # "never"
        if self.length_penalty > 0.0:
            return self.worst_score >= best_sum_logprobs / self.max_length ** self.length_penalty
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


class BeamScorer437:
    """transformers 4.37.2 BeamSearchScorer.process / .finalize on host lists (a step hands over 2 x num_beams candidates per batch
    item: a few dozen numbers)."""

    def __init__(self, batch: int, num_beams: int, max_length: int, length_penalty: float = 1.0, early_stopping=False):
        self.batch, self.nb, self.max_length = batch, num_beams, max_length
        self.hyps = [_BeamHypotheses(num_beams, length_penalty, early_stopping, max_length) for _ in range(batch)]
        self.done = [False] * batch

    @property
    def is_done(self) -> bool:
        return all(self.done)

    def process(self, seqs: List[List[int]], next_scores, next_tokens, next_indices, pad_token_id, eos_token_ids):
        """seqs: the batch * num_beams running id lists; next_*: [batch][K] candidates sorted by score (descending).
        -> (beam_scores, beam_tokens, beam_idx) flat lists of batch * num_beams"""
        nb = self.nb
        cur_len = (len(seqs[0]) if seqs else 0) + 1
        out_s, out_t, out_i = [], [], []
        for b in range(self.batch):
            if self.done[b]:
                # a finished batch item keeps ITS OWN rows (HF pads with beam index 0 = row 0 of the whole batch; the caller here
                # permutes the KV cache by these indices, and a finished item of a ragged batch must not inherit row 0's shorter cache:
                # its logits would be read from unwritten positions)
                out_s += [0.0] * nb
                out_t += [pad_token_id if pad_token_id is not None else 0] * nb
                out_i += [b * nb + j for j in range(nb)]
                continue
            kept = 0
            for rank, (tok, sc, idx) in enumerate(zip(next_tokens[b], next_scores[b], next_indices[b])):
                row = b * nb + idx
                if eos_token_ids is not None and tok in eos_token_ids:
                    if rank >= nb:
                        continue
                    self.hyps[b].add(list(seqs[row]), sc, cur_len)
                else:
                    out_s.append(sc)
                    out_t.append(tok)
                    out_i.append(row)
                    kept += 1
                if kept == nb:
                    break
            if kept < nb:
                raise ValueError(f"At most {nb} tokens in {next_tokens[b]} can be equal to `eos_token_id: {eos_token_ids}`. "
                                 "Make sure they are defined correctly.")
            self.done[b] = self.done[b] or self.hyps[b].is_done(max(next_scores[b]), cur_len)
        return out_s, out_t, out_i

    def finalize(self, seqs: List[List[int]], beam_scores: List[float], pad_token_id, eos_token_ids) -> List[List[int]]:
        nb = self.nb
        for b in range(self.batch):
            if self.done[b]:
                continue
            for j in range(nb):
                row = b * nb + j
                self.hyps[b].add(list(seqs[row]), beam_scores[row], len(seqs[row]))
        best = []
        for b in range(self.batch):
            best.append(sorted(self.hyps[b].beams, key=lambda x: x[0]).pop()[1])
        lengths = [len(h) for h in best]
        sent_max_len = min(max(lengths) + 1, self.max_length)
        if min(lengths) != max(lengths) and pad_token_id is None:
            raise ValueError("`pad_token_id` has to be defined")
        rows = []
        for h in best:
            row = list(h[:sent_max_len])
            if len(h) < sent_max_len:
                row.append(eos_token_ids[0])  # "inserting only the first eos_token_id"
            fill = pad_token_id if pad_token_id is not None else (eos_token_ids[0] if eos_token_ids else 0)
            rows.append(row + [fill] * (sent_max_len - len(row)))
        return rows


def beam_sample_candidates(scores: torch.Tensor, n: int, generator: Optional[torch.Generator] = None):
    """`torch.multinomial(softmax(scores), n)` WITHOUT replacement over the flattened (beam, token) axis, then HF's re-sort by score:
    scores [B, num_beams * V] (filtered entries -inf) -> (cand_scores [B, n] descending, cand_index [B, n]).
    Drawn as the top n of scores + Gumbel noise -- the Plackett-Luce order statistic, i.e. exactly sequential draws without
    replacement from the softmax (Gumbel-top-k) -- so one top-k replaces n dependent multinomial draws.  A row with fewer than n
    finite entries raises like torch.multinomial does ("invalid multinomial distribution")."""
    if int(torch.isfinite(scores).sum(-1).min()) < n:
        raise RuntimeError("invalid multinomial distribution (with replacement=False, not enough non-negative category to sample)")
    u = torch.rand(scores.shape, device=scores.device, dtype=torch.float32, generator=generator).clamp_(1e-20, 1.0 - 1e-7)
    keys = scores.float() - torch.log(-torch.log(u))
    idx = torch.topk(keys, n, dim=-1)[1]
    sc = torch.gather(scores.float(), -1, idx)
    sc, order = torch.sort(sc, descending=True, dim=-1, stable=True)
    return sc, torch.gather(idx, -1, order)


def beam_generate(first_logits: torch.Tensor, step, batch: int, num_beams: int, max_new_tokens: int,
                  eos_token_ids: Optional[List[int]], pad_token_id: Optional[int], do_sample: bool = False, temperature=None,
                  top_k=None, top_p=None, generator: Optional[torch.Generator] = None, length_penalty: float = 1.0,
                  early_stopping=False, stopping_criteria=None, warp_before_beam_scores: bool = True) -> torch.Tensor:
    """`generate(num_beams > 1)` of the reference's callers -- `--num_beams` of eval_spatial.py:231-235, eval_region_cls.py:318-322,
    model_vqa.py:72-76, which pass `do_sample = temperature > 0` with `--temperature` defaulting to 0.2: the default flags plus
    `--num_beams 3` are BEAM-SAMPLE -- as transformers 4.37.2 runs them (GenerationMixin.beam_search / .beam_sample +
    BeamSearchScorer), over a decoder fed `inputs_embeds` (ids start empty; lengths count new tokens).

    Every step: scores = log_softmax(logits) (+ the running beam scores); beam search takes the max(2, 1 + #eos) * num_beams best
    (beam, token) continuations per batch item; beam-sample warps the token scores (temperature -> top-k -> top-p, per beam row) and
    THEN adds the running beam scores -- `logits_warper(logits_processor(log_softmax))  + beam_scores`, the order of the releases
    since the "temperature also scales the beam scores" fix of 2023 (the pinned 4.37.2 wheel cannot be inspected offline; the 5.15
    here, which minted tests/golden/beam_kat.npz, has this order; `warp_before_beam_scores=False` gives the older releases' warp of
    the summed scores) --, draws 2 * num_beams continuations without replacement from their softmax and sorts them by score.  BeamScorer437.process keeps
    the first num_beams that do not end in an EOS id and files the EOS ones that rank among the first num_beams as hypotheses.

    first_logits fp32 [batch * num_beams, V]; step(tokens int64 [batch * num_beams], beam_idx int64 [batch * num_beams]) -> logits of
    the next position for rows continued from (old) rows beam_idx.  Returns int64 [batch, <= max_new_tokens]: new tokens only."""
    dev = first_logits.device
    B, nb, G = batch, num_beams, max_new_tokens
    V = first_logits.shape[-1]
    eos = list(eos_token_ids) if eos_token_ids else None
    scorer = BeamScorer437(B, nb, G, length_penalty, early_stopping)
    beam_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9
    seqs: List[List[int]] = [[] for _ in range(B * nb)]
    flat_scores = beam_scores.reshape(-1).tolist()
    logits = first_logits
    K = 2 * nb if do_sample else max(2, 1 + (len(eos) if eos else 0)) * nb
    # `_get_logits_warper` (4.37.2): with beams every warper keeps at least one token per EOS id + 1 (2 for a single id / none)
    keep_min = (len(eos) + 1) if (eos and len(eos) > 1) else 2
    while True:
        lp = torch.log_softmax(logits.float(), dim=-1)
        if do_sample and warp_before_beam_scores:
            lp = warp_logits(lp, temperature, top_k, top_p, min_tokens_to_keep=keep_min)
        sc = lp + beam_scores.reshape(B * nb, 1)
        if do_sample:
            if not warp_before_beam_scores:
                sc = warp_logits(sc, temperature, top_k, top_p, min_tokens_to_keep=keep_min)
            cs, ci = beam_sample_candidates(sc.reshape(B, nb * V), K, generator)
        else:
            cs, ci = torch.topk(sc.reshape(B, nb * V), K, dim=1, largest=True, sorted=True)
        host = torch.stack((cs.double(), (ci // V).double(), (ci % V).double())).cpu()  # one copy per step: [3, B, K]
        n_scores, n_idx, n_tok = host[0].tolist(), [[int(v) for v in r] for r in host[1].tolist()], [[int(v) for v in r] for r in host[2].tolist()]
        flat_scores, toks, rows = scorer.process(seqs, n_scores, n_tok, n_idx, pad_token_id, eos)
        seqs = [seqs[r] + [t] for r, t in zip(rows, toks)]
        beam_scores = torch.tensor(flat_scores, dtype=torch.float32, device=dev).reshape(B, nb)
        stop = scorer.is_done or len(seqs[0]) >= G
        if not stop and stopping_criteria:
            ids = torch.tensor(seqs, dtype=torch.int64)
            for crit in stopping_criteria:
                r = crit(ids, None)
                stop = stop or (bool(r.all()) if isinstance(r, torch.Tensor) else bool(r))
        if stop:
            break
        logits = step(torch.tensor(toks, dtype=torch.int64, device=dev), torch.tensor(rows, dtype=torch.int64, device=dev))
    out = scorer.finalize(seqs, flat_scores, pad_token_id, eos)
    return torch.tensor(out, dtype=torch.int64, device=dev)
