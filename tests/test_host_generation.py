"""CPU: generation-config resolution and the sampling warpers, pinned to HF.

The reference generates through `self.llm.generate(...)` (llava_llama.py:212), i.e. HF `GenerationMixin` on
`llm.generation_config` (= <ckpt>/llm/generation_config.json) overwritten by the call's keywords.  The demo path samples
(gradio_web_server_multi.py:202-213, temperature 0.2): the filtered distribution of `generation.warp_logits` must equal HF's own
`TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper` chain on the same logits (transformers is installed here)."""
import json
import os

import pytest
import torch

from spatialrgpt_amd.generation import NOT_GIVEN, generation_config_from_files, resolve_generation, warp_logits


def _hf_chain(logits, temperature, top_k, top_p):
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

    s = logits.clone()
    ids = torch.zeros((logits.shape[0], 1), dtype=torch.long)
    if temperature is not None and temperature != 1.0:
        s = TemperatureLogitsWarper(temperature)(ids, s)
    if top_k is not None and top_k != 0:
        s = TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1)(ids, s)
    if top_p is not None and top_p < 1.0:
        s = TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)(ids, s)
    return s


@pytest.mark.parametrize("temperature,top_k,top_p", [(0.2, 50, None), (0.2, 50, 1.0), (0.7, None, 0.9), (0.6, 50, 0.9), (1.0, 5, 0.5),
                                                     (1.3, 0, 0.95), (0.2, 1, None), (2.0, 1000, 0.3), (0.01, 50, 0.999)])
def test_warped_distribution_equals_hf_warpers(temperature, top_k, top_p):
    g = torch.Generator().manual_seed(17)
    V = 517
    logits = torch.randn((6, V), generator=g) * 3.0
    logits[1] = logits[1].round()          # heavy ties
    logits[2, :] = 0.25                     # all equal
    logits[3, 7] = 40.0                     # one dominant token
    logits[4] = torch.linspace(-8, 8, V)    # sorted input
    got = warp_logits(logits, temperature, top_k, top_p)
    ref = _hf_chain(logits, temperature, top_k, top_p)
    assert torch.equal(torch.isinf(got), torch.isinf(ref)), "different kept-token sets"
    assert torch.equal(got, ref)
    assert torch.equal(got.softmax(-1), ref.softmax(-1))  # the distribution torch.multinomial draws from
    assert bool((got.softmax(-1).sum(-1) - 1).abs().max() < 1e-5)


def test_resolution_follows_hf_generate():
    # Llama-3-Instruct style generation_config.json: a LIST of EOS ids, sampling defaults stored in the file
    stored = {"bos_token_id": 128000, "eos_token_id": [128001, 128009], "do_sample": True, "temperature": 0.6, "top_p": 0.9}
    # eval_spatial.py:224-236: do_sample False, temperature 0, top_p None, num_beams 1, max_new_tokens 128 -- no eos / pad / criteria
    g = resolve_generation(stored, do_sample=False, temperature=0.0, top_p=None, num_beams=1, max_new_tokens=128,
                           top_k=NOT_GIVEN, eos_token_id=NOT_GIVEN, pad_token_id=NOT_GIVEN)
    assert g.eos_token_ids == [128001, 128009] and g.pad_token_id == 128001 and g.max_new_tokens == 128
    assert g.do_sample is False and g.top_p is None and g.top_k == 50
    # demo (gradio_web_server_multi.py:202-213): temperature 0.2 sampling, nothing else -> stored top_p, default top_k
    g = resolve_generation(stored, do_sample=True, temperature=0.2, max_new_tokens=512)
    assert g.do_sample and g.temperature == 0.2 and g.top_p == 0.9 and g.top_k == 50
    # nothing stored: 4.37.2 defaults (greedy, 20 new tokens, never stops on a token)
    g = resolve_generation({})
    assert (g.do_sample, g.temperature, g.top_k, g.top_p, g.max_new_tokens, g.eos_token_ids, g.pad_token_id) == \
        (False, 1.0, 50, 1.0, 20, None, None)
    # an explicit None overrides the stored value (generation_config.update(**kwargs)); an int EOS stays one id; explicit pad wins
    g = resolve_generation(stored, eos_token_id=None)
    assert g.eos_token_ids is None and g.pad_token_id is None
    g = resolve_generation({"eos_token_id": 2}, pad_token_id=0, max_length=7)
    assert g.eos_token_ids == [2] and g.pad_token_id == 0 and g.max_new_tokens == 7
    g = resolve_generation({"eos_token_id": 2, "pad_token_id": 5})
    assert g.pad_token_id == 5


def test_generation_config_file_beats_model_config(tmp_path):
    """`from_pretrained` semantics: <llm>/generation_config.json when present, else the generation fields of <llm>/config.json;
    `config_from_checkpoint` carries the result (builder.py)."""
    lc = {"eos_token_id": 2, "bos_token_id": 1, "hidden_size": 64}
    assert generation_config_from_files(lc, None) == {"eos_token_id": 2, "bos_token_id": 1}
    assert generation_config_from_files(lc, {"eos_token_id": [7, 9], "temperature": 0.6, "transformers_version": "4.37.2"}) == \
        {"eos_token_id": [7, 9], "temperature": 0.6}
    from tests.test_host_loader import _ckpt

    from spatialrgpt_amd.builder import config_from_checkpoint, load_tokenizer, read_checkpoint

    root, cfgd, w = _ckpt(tmp_path)
    cfg = config_from_checkpoint(root)
    assert cfg.eos_token_id == 2 and cfg.generation_config == {"eos_token_id": 2, "bos_token_id": 1}
    json.dump({"bos_token_id": 1, "eos_token_id": [2, 9], "do_sample": True, "temperature": 0.6, "top_p": 0.9},
              open(os.path.join(root, "llm", "generation_config.json"), "w"))
    cfg, sd = read_checkpoint(root)
    assert cfg.eos_token_id == [2, 9] and cfg.generation_config["top_p"] == 0.9 and cfg.pad_token_id is None
    # tokenizer kwargs of build_llm_and_tokenizer (language_model/builder.py:84-91): padding_side="right", and the loader's
    # model_max_length (None on the eval path) overrides tokenizer_config.json's 4096
    tok = load_tokenizer(root, cfg, sd)
    assert tok.padding_side == "right" and tok.model_max_length > 10 ** 9
    cfg2, sd2 = read_checkpoint(root)
    tok2 = load_tokenizer(root, cfg2, sd2, model_max_length=777)
    assert tok2.model_max_length == 777
    assert cfg.eos_token_id == [2, 9]  # the tokenizer's eos id never replaces the generation config's


def test_beam_search_reproduces_the_references_generate_num_beams_3():
    """tests/golden/beam_kat.npz = ids of the REFERENCE model's generate(num_beams=3) (oracle/make_golden.py beam) on the tiny fp32
    model of tiny_fp32.npz: no EOS, an EOS list the search meets (length-penalised hypotheses of different lengths), a batch of two.
    `generation.beam_search`, driven by the oracle's llama_forward on the CPU, returns the same ids -- and not the greedy ones."""
    import os

    import numpy as np

    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.generation import beam_search
    from tests.util import GOLD, load_tiny

    cfgd, dtype, w, inp, ref = load_tiny("tiny_fp32.npz")
    cfg = so.SrgptConfig(**{k: v for k, v in cfgd.items() if k in so.SrgptConfig.__dataclass_fields__})
    z = np.load(os.path.join(GOLD, "beam_kat.npz"))
    NB, G, PAD = int(z["num_beams"]), int(z["max_new_tokens"]), int(z["pad_token_id"])

    def run(ids, images, depths, masks, eos):
        emb, _, _, _ = so.prepare_inputs(w, cfg, ids, images, depths, masks)
        B, T, _ = emb.shape
        state = {"kv": so.KVCache(cfg.layers), "pos": T}
        first = so.llama_forward(w, cfg, emb.repeat_interleave(NB, dim=0), torch.arange(T)[None].expand(B * NB, -1), state["kv"],
                                 last_only=True)[:, -1]

        def step(tokens, beam_idx):
            kv = state["kv"]
            kv.k = [k.index_select(0, beam_idx) for k in kv.k]
            kv.v = [v.index_select(0, beam_idx) for v in kv.v]
            e = torch.nn.functional.embedding(tokens[:, None], w["llm.model.embed_tokens.weight"])
            lg = so.llama_forward(w, cfg, e, torch.full((B * NB, 1), state["pos"]), kv, last_only=True)[:, -1]
            state["pos"] += 1
            return lg

        return beam_search(first, step, B, NB, G, eos, PAD)

    one = (inp["input_ids"], inp["images"], inp["depths"], inp["masks"])
    out = run(*one, None)
    assert torch.equal(out, torch.from_numpy(z["noeos.ids"])) and not torch.equal(out, ref["new_ids"])
    eos = z["eos.eos"].tolist()
    assert torch.equal(run(*one, eos), torch.from_numpy(z["eos.ids"]))
    images = torch.from_numpy(z["batch2.images_q32"].astype(np.float32)) / 32
    depths = (torch.from_numpy(z["batch2.depths_q32"].astype(np.float32)) / 32).expand(-1, 3, -1, -1).contiguous()
    masks = [torch.from_numpy(m.astype(np.float32)) for m in z["batch2.masks_u8"]]
    assert torch.equal(run(torch.from_numpy(z["batch2.input_ids"]), images, depths, masks, eos), torch.from_numpy(z["batch2.ids"]))


# ------------------------------------------------------------------------------------------------ beams, transformers 4.37.2 semantics
def _oracle_beam_driver():
    import numpy as np

    from oracle import srgpt_oracle as so
    from tests.util import GOLD, load_tiny

    cfgd, dtype, w, inp, ref = load_tiny("tiny_fp32.npz")
    cfg = so.SrgptConfig(**{k: v for k, v in cfgd.items() if k in so.SrgptConfig.__dataclass_fields__})
    z = np.load(os.path.join(GOLD, "beam_kat.npz"))
    NB, G, PAD = int(z["num_beams"]), int(z["max_new_tokens"]), int(z["pad_token_id"])

    def run(fn, ids, images, depths, masks, eos, **kw):
        emb, _, _, _ = so.prepare_inputs(w, cfg, ids, images, depths, masks)
        B, T, _ = emb.shape
        state = {"kv": so.KVCache(cfg.layers), "pos": T}
        first = so.llama_forward(w, cfg, emb.repeat_interleave(NB, dim=0), torch.arange(T)[None].expand(B * NB, -1), state["kv"],
                                 last_only=True)[:, -1]

        def step(tokens, beam_idx):
            kv = state["kv"]
            kv.k = [k.index_select(0, beam_idx) for k in kv.k]
            kv.v = [v.index_select(0, beam_idx) for v in kv.v]
            e = torch.nn.functional.embedding(tokens[:, None], w["llm.model.embed_tokens.weight"])
            lg = so.llama_forward(w, cfg, e, torch.full((B * NB, 1), state["pos"]), kv, last_only=True)[:, -1]
            state["pos"] += 1
            return lg

        return fn(first, step, B, NB, G, eos, PAD, **kw)

    one = (inp["input_ids"], inp["images"], inp["depths"], inp["masks"])
    images = torch.from_numpy(z["batch2.images_q32"].astype(np.float32)) / 32
    depths = (torch.from_numpy(z["batch2.depths_q32"].astype(np.float32)) / 32).expand(-1, 3, -1, -1).contiguous()
    masks = [torch.from_numpy(m.astype(np.float32)) for m in z["batch2.masks_u8"]]
    two = (torch.from_numpy(z["batch2.input_ids"]), images, depths, masks)
    return run, one, two, z, (NB, G, PAD)


def test_beam_generate_4_37_2_reproduces_the_references_generate_num_beams_3():
    """The product path (generation.beam_generate: the reference's pinned BeamSearchScorer semantics) returns the ids of the
    reference model's own generate(num_beams=3) on all three golden fixtures -- the same fixtures the vectorised 5.x form is pinned
    to, so the two restatements agree wherever the reference itself was run."""
    from spatialrgpt_amd.generation import beam_generate

    run, one, two, z, _ = _oracle_beam_driver()
    eos = z["eos.eos"].tolist()
    assert torch.equal(run(beam_generate, *one, None), torch.from_numpy(z["noeos.ids"]))
    assert torch.equal(run(beam_generate, *one, eos), torch.from_numpy(z["eos.ids"]))
    assert torch.equal(run(beam_generate, *two, eos), torch.from_numpy(z["batch2.ids"]))


def test_beam_scorer_rules_of_the_pinned_release():
    """The places where transformers 4.37.2's BeamSearchScorer differs from the installed release, each on a hand-made
    step: (1) is_done compares the worst kept hypothesis with the best of ALL candidates of the step, EOS ones included;
    (2) finalize() writes eos_token_id[0] behind the hypothesis whichever id ended it; (3) an EOS candidate ranked below num_beams
    is skipped, not filed; (4) more EOS candidates than the 2 x num_beams window can absorb raise."""
    from spatialrgpt_amd.generation import BeamScorer437

    EOS = [7, 9]
    sc = BeamScorer437(batch=1, num_beams=2, max_length=10)
    seqs = [[3], [4]]
    # ranks 0, 1 end in EOS ids (filed: 2 hypotheses = num_beams), ranks 2, 3 continue
    s, t, i = sc.process(seqs, [[-1.0, -1.5, -2.0, -2.5]], [[9, 7, 5, 6]], [[0, 1, 0, 1]], pad_token_id=0, eos_token_ids=EOS)
    assert (t, i) == ([5, 6], [0, 1]) and s == [-2.0, -2.5]
    assert sorted(h[0] for h in sc.hyps[0].beams) == [-1.5 / 2, -1.0 / 2]          # sum_logprobs / generated_len (EOS counted)
    # (1): worst kept = -0.75; best of ALL candidates = -1.0 -> -1.0 / 2 = -0.5 > -0.75: NOT done (the best RUNNING beam, -2.0 / 2 =
    # -1.0, would have ended the search -- the 5.x rule)
    assert sc.done == [False]
    # next step: every candidate is worse than the kept hypotheses -> done
    seqs = [[3, 5], [4, 6]]
    sc.process(seqs, [[-4.0, -4.5, -5.0, -5.5]], [[5, 5, 6, 6]], [[0, 1, 0, 1]], pad_token_id=0, eos_token_ids=EOS)
    assert sc.done == [True]
    # (2): the best hypothesis is [3] (ended by id 9 = eos[1]); the returned row carries eos[0] = 7 behind it
    assert sc.finalize([[3, 5, 5], [4, 6, 5]], [-4.0, -4.5], pad_token_id=0, eos_token_ids=EOS) == [[3, 7]]
    # (3)
    sc = BeamScorer437(batch=1, num_beams=2, max_length=10)
    sc.process([[1], [2]], [[-1.0, -1.1, -1.2, -1.3]], [[5, 6, 7, 8]], [[0, 0, 1, 1]], pad_token_id=0, eos_token_ids=EOS)
    assert len(sc.hyps[0]) == 0
    # (4)
    sc = BeamScorer437(batch=1, num_beams=2, max_length=10)
    with pytest.raises(ValueError, match="eos_token_id"):
        sc.process([[1], [2]], [[-1.0, -1.1, -1.2, -1.3]], [[7, 9, 7, 5]], [[0, 0, 1, 1]], pad_token_id=0, eos_token_ids=EOS)
    # unfinished items: finalize files the running beams at their current length; rows of different lengths need a pad id
    sc = BeamScorer437(batch=2, num_beams=1, max_length=3)
    sc.hyps[0].add([4], -0.1, 2)
    sc.done[0] = True
    assert sc.finalize([[4, 4, 4], [5, 6, 8]], [-9.0, -3.0], pad_token_id=0, eos_token_ids=[7]) == [[4, 7, 0], [5, 6, 8]]


def test_beam_sample_is_the_warped_softmax_without_replacement():
    """beam-sample's draw: (a) the scores it draws from are HF's warper chain (min_tokens_to_keep = 2 with beams) on log_softmax +
    beam scores; (b) Gumbel-top-k == torch.multinomial(replacement=False): the inclusion frequencies of both agree (chi-square
    against the exact Plackett-Luce inclusion probabilities of a 6-way distribution, 3 draws); (c) candidates come back sorted by
    score; (d) a row with too few finite scores raises like torch.multinomial."""
    import itertools

    from spatialrgpt_amd.generation import beam_sample_candidates

    # (a)
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(3)
    lp = torch.log_softmax(torch.randn((6, 97), generator=g) * 2, -1) + torch.tensor([0.0, -1.0, -3.0, 0.0, -0.5, -2.0])[:, None]
    ids = torch.zeros((6, 1), dtype=torch.long)
    ref = TopPLogitsWarper(top_p=0.8, min_tokens_to_keep=2)(ids, TopKLogitsWarper(top_k=20, min_tokens_to_keep=2)(
        ids, TemperatureLogitsWarper(0.2)(ids, lp.clone())))
    assert torch.equal(warp_logits(lp, 0.2, 20, 0.8, min_tokens_to_keep=2), ref)
    # top_k = 1 with beams still keeps two entries per row
    assert int(torch.isfinite(warp_logits(lp, 0.2, 1, None, min_tokens_to_keep=2)).sum(-1).min()) == 2
    # (b) exact inclusion probabilities of sampling 3 of 6 without replacement
    p = torch.tensor([0.4, 0.25, 0.15, 0.1, 0.07, 0.03], dtype=torch.float64)
    incl = torch.zeros(6, dtype=torch.float64)
    for perm in itertools.permutations(range(6), 3):
        pr, rest = 1.0, 1.0
        for k in perm:
            pr *= float(p[k]) / rest
            rest -= float(p[k])
        for k in perm:
            incl[k] += pr
    N = 40000
    scores = p.log().float()[None].expand(N, -1).contiguous()
    gen = torch.Generator().manual_seed(11)
    sc, idx = beam_sample_candidates(scores, 3, gen)
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())                     # (c)
    assert torch.equal(sc, torch.gather(scores, 1, idx))
    counts = torch.bincount(idx.reshape(-1), minlength=6).double()
    # each category is included or not per trial: binomial(N, incl); z-score bound 4.5 (two-sided 7e-6 per category)
    zs = (counts - N * incl) / torch.sqrt(N * incl * (1 - incl))
    assert float(zs.abs().max()) < 4.5, zs
    mult = torch.multinomial(p.float()[None].expand(N, -1), 3, replacement=False, generator=torch.Generator().manual_seed(12))
    zs2 = (torch.bincount(mult.reshape(-1), minlength=6).double() - N * incl) / torch.sqrt(N * incl * (1 - incl))
    assert float(zs2.abs().max()) < 4.5, zs2                          # torch's own sampler against the same expectation
    # (d)
    bad = torch.full((2, 8), float("-inf"))
    bad[:, :2] = 0.0
    with pytest.raises(RuntimeError, match="invalid multinomial"):
        beam_sample_candidates(bad, 3)


def test_beam_sample_through_the_oracle_is_seeded_and_respects_eos():
    """`--num_beams 3` under the eval CLIs' DEFAULT flags (temperature 0.2 -> do_sample True; eval_spatial.py:231-235, :274) is HF
    beam-sample: runs end to end on the oracle's decoder, is reproducible from the generator's seed, differs between seeds at
    temperature 1, never returns more than max_new_tokens, and a row that stops on an EOS id ends with eos_token_id[0]."""
    from spatialrgpt_amd.generation import beam_generate

    run, one, two, z, (NB, G, PAD) = _oracle_beam_driver()
    eos = z["eos.eos"].tolist()

    def go(args, seed, **kw):
        return run(beam_generate, *args, eos, do_sample=True, generator=torch.Generator().manual_seed(seed), **kw)

    a = go(one, 5, temperature=0.2, top_k=50, top_p=None)
    assert torch.equal(a, go(one, 5, temperature=0.2, top_k=50, top_p=None))
    outs = {tuple(go(one, s_, temperature=1.0, top_k=50, top_p=0.95).reshape(-1).tolist()) for s_ in range(8)}
    assert len(outs) > 1
    for s_ in range(4):
        o = go(two, s_, temperature=0.7, top_k=20, top_p=0.9)
        assert o.shape[0] == 2 and 1 <= o.shape[1] <= G
        for row in o.tolist():
            hit = [j for j, t in enumerate(row) if t in eos]
            if hit:  # the token at the stop position is eos[0]; everything behind it is padding
                assert row[hit[0]] == eos[0] and all(t == PAD for t in row[hit[0] + 1:])
    # the older releases' warper order (warp the summed scores) is available and also runs
    go(one, 1, temperature=0.2, top_k=50, top_p=None, warp_before_beam_scores=False)
