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
