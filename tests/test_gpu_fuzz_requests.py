"""Seeded random REQUESTS through the whole path -- both tower passes, refinement, pooling, projector, splice, prefill, greedy
decode -- against the oracle on the CPU (fp32, tiny widths, the true 27 x 27 patch grid): batches of 1..4 prompts with a text-only
prompt among them, 0..4 regions per image, `masks[i] = None`, `depths=None`, ragged prompt lengths under LEFT padding (the one
ragged form in which HF's `logits[:, -1]` is every row's own last token: llava_arch.py:575-600, modeling_llama.py:1127-1133).
The fixtures pin single shapes; this sweeps the combinations.  SRGPT_FUZZ_CASES=<n> widens the sweep (default 10 cases)."""
import os

import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
IMAGE_TOKEN_INDEX = -200
CFG = dict(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=378, patch_size=14, hidden=64, inter=160, layers=2,
           heads=4, kv_heads=2, vocab=128, mask_token_id=120, depth_token_id=121, rope_theta=500000.0)
G = 5


def _request(seed):
    """-> dict(input_ids [B,P], attention_mask | None, images [Nimg,3,S,S], depths | None, masks list(len Nimg), side)"""
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    S = CFG["image_size"]
    B = ri(1, 4)
    have_depths = ri(0, 4) != 0
    rows, masks = [], []
    has_img = [int(ri(0, 4) != 0) for _ in range(B)]  # images per prompt
    if not any(has_img):
        has_img[ri(0, B - 1)] = 1
    for b in range(B):
        txt = lambda n: torch.randint(3, 118, (n,), generator=g).tolist()  # noqa: E731
        seq = [1] + txt(ri(1, 6))
        if has_img[b]:
            seq += [IMAGE_TOKEN_INDEX] + txt(ri(0, 4))
            k = ri(0, 4)
            none_entry = k == 0  # no regions: None (an EMPTY mask tensor raises in the reference's F.interpolate, base_extractor.py:53-58)
            m = torch.zeros((k, S, S))
            kt = k if ri(0, 3) else ri(0, k)  # sometimes fewer <mask> ids than region embeddings (the surplus is unused, llava_arch.py:470-485)
            for r in range(k):
                hh, ww = ri(S // 8, S // 2), ri(S // 8, S // 2)
                y0, x0 = ri(0, S - hh), ri(0, S - ww)
                m[r, y0:y0 + hh, x0:x0 + ww] = 1.0
                if r < kt:
                    seq += txt(ri(0, 3)) + [CFG["mask_token_id"]] + ([CFG["depth_token_id"]] if have_depths else [])
            masks.append(None if none_entry else m)
            if ri(0, 4) == 0:
                # a second image in the same prompt: its rows are spliced at its sentinel; the prompt's <mask> / <depth> ids all take
                # the FIRST image's region embeddings (mask_embeds[cur_image_idx] is read before the images are walked)
                seq += txt(ri(0, 2)) + [IMAGE_TOKEN_INDEX]
                k2 = ri(0, 2)
                m2 = torch.zeros((k2, S, S))
                m2[:, S // 4:S // 2, S // 3:S // 2] = 1.0
                masks.append(m2 if k2 else None)
                has_img[b] += 1
        seq += txt(ri(1, 5))
        rows.append(seq)
    P = max(len(r) for r in rows)
    ragged = any(len(r) != P for r in rows)
    ids = torch.zeros((B, P), dtype=torch.long)
    am = torch.zeros((B, P), dtype=torch.bool)
    for b, r in enumerate(rows):  # left padding
        ids[b, P - len(r):] = torch.tensor(r)
        am[b, P - len(r):] = True
    n_img = sum(has_img)
    images = (torch.randn((n_img, 3, S, S), generator=g).clamp_(-1, 1) * 32).round().div(32)
    depths = None
    if have_depths:
        depths = (torch.randn((n_img, 1, S, S), generator=g).clamp_(-1, 1) * 32).round().div(32).expand(-1, 3, -1, -1).contiguous()
    return dict(input_ids=ids, attention_mask=am if (ragged or ri(0, 1)) else None, images=images, depths=depths, masks=masks,
                text_only_rows=[b for b in range(B) if not has_img[b]])


@pytest.fixture(scope="module")
def pair():
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel

    ocfg = so.SrgptConfig(**CFG, padding_side="left")
    w = so.synth_weights(ocfg, seed=11, dtype=torch.float32, std=0.08)  # wider than N(0, 0.02): decisions with margins
    model = LlavaLlamaModel(SrgptConfig(**CFG, padding_side="left"), dict(w), device=DEV, dtype=torch.float32, rope_positions=512)
    return so, ocfg, w, model


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SRGPT_FUZZ_CASES", "10")))))
def test_random_request_equals_the_oracle(pair, seed):
    so, ocfg, w, model = pair
    r = _request(1000 + seed)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    ids_o, st = so.generate(w, ocfg, r["input_ids"], r["images"], r["depths"], r["masks"], r["attention_mask"], max_new_tokens=G,
                            eos_token_id=None, return_stages=True)
    # (1) the spliced embeddings and mask
    emb, am, _ = model.engine.prepare_inputs(dev(r["input_ids"]), dev(r["images"]), dev(r["depths"]), [dev(m) for m in r["masks"]],
                                             dev(r["attention_mask"]))
    ref = st["inputs_embeds"]
    assert emb.shape == ref.shape, (emb.shape, ref.shape)
    assert_close(emb, ref, 2e-4 * float(ref.abs().max()), 0, f"seed {seed}: inputs_embeds")
    # (2) greedy ids: equal, or first different where the oracle's own top-2 margin is inside the fp32 tolerance of the logits
    out = model.generate(dev(r["input_ids"]), images=dev(r["images"]), depths=dev(r["depths"]), masks=[dev(m) for m in r["masks"]],
                         attention_mask=dev(r["attention_mask"]), do_sample=False, max_new_tokens=G, eos_token_id=None).cpu()
    assert out.shape == ids_o.shape
    sl = st["step_logits"]  # [B, G, V]
    tol = 4e-4 * float(sl.abs().max())
    for b in range(out.shape[0]):
        for t in range(G):
            if int(out[b, t]) != int(ids_o[b, t]):
                top2 = sl[b, t].topk(2).values
                assert float(top2[0] - top2[1]) <= tol, f"seed {seed}: row {b} step {t}: {out[b].tolist()} vs {ids_o[b].tolist()}"
                break  # a flipped near-tie changes every later input of this row
    # (2b) an EOS id some row actually produces: rows finish at different steps, finished rows are padded (HF: pad_token_id defaults to
    #      the eos id), the loop ends when every row has finished -- only where (2) found the ids equal (same trajectories)
    if torch.equal(out, ids_o):
        eos = int(ids_o[seed % ids_o.shape[0], 1 + seed % (G - 1)])
        want = so.generate(w, ocfg, r["input_ids"], r["images"], r["depths"], r["masks"], r["attention_mask"], max_new_tokens=G,
                           eos_token_id=eos)
        got = model.generate(dev(r["input_ids"]), images=dev(r["images"]), depths=dev(r["depths"]), masks=[dev(m) for m in r["masks"]],
                             attention_mask=dev(r["attention_mask"]), do_sample=False, max_new_tokens=G, eos_token_id=eos).cpu()
        assert got.shape == want.shape and torch.equal(got, want), f"seed {seed}: eos {eos}: {got.tolist()} vs {want.tolist()}"
    # (3) forward(labels=...) (llava_llama.py:100-192): spliced labels equal, loss within 2e-5, logits at the valid positions
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, CFG["vocab"], r["input_ids"].shape, generator=g)
    labels[torch.rand(labels.shape, generator=g) < 0.3] = -100
    am_in = r["attention_mask"] if r["attention_mask"] is not None else torch.ones_like(r["input_ids"], dtype=torch.bool)  # forward() needs one (llava_llama.py:131)
    # a row's FIRST token carries no target: under left padding its shifted partner is the logits of a padding position, which
    # the reference's varlen attention leaves at lm_head(0) = 0 (as here) and the oracle's masked eager attention does not
    labels[torch.arange(labels.shape[0]), am_in.int().argmax(dim=1)] = -100
    image_features, mask_embeds, depth_embeds, _ = so.encode_visual(w, ocfg, r["images"], r["depths"], r["masks"])
    _, am_o, _, new_labels = so.splice(w, ocfg, r["input_ids"], am_in, image_features, mask_embeds, depth_embeds,
                                       have_depths=r["depths"] is not None, labels=labels)
    if r["attention_mask"] is None:
        # no mask in, none out (llava_arch.py:614-617) -- even when the spliced rows differ in length (a two-image prompt next to a
        # one-image prompt): steps (1) and (2) ran the LLM over the padding like the reference; forward() gets the all-ones mask
        _, st = so.generate(w, ocfg, r["input_ids"], r["images"], r["depths"], r["masks"], am_in, max_new_tokens=1, eos_token_id=None,
                            return_stages=True)
    loss_o = so.causal_lm_loss(st["prefill_logits"], new_labels)
    out_f = model(input_ids=dev(r["input_ids"]), images=dev(r["images"]), masks=[dev(m) for m in r["masks"]], depths=dev(r["depths"]),
                  attention_mask=dev(am_in), labels=dev(labels))
    if bool(torch.isnan(loss_o)):
        assert bool(torch.isnan(out_f.loss))
    else:
        assert abs(float(out_f.loss) - float(loss_o)) <= 2e-5 * max(1.0, abs(float(loss_o))), (float(out_f.loss), float(loss_o))
    keep = torch.ones(ref.shape[:2], dtype=torch.bool) if am_o is None else am_o.bool()
    lo = st["prefill_logits"]
    assert_close(out_f.logits.cpu()[keep], lo[keep], 4e-4 * float(lo.abs().max()), 0, f"seed {seed}: forward logits at the valid positions")


def test_an_empty_mask_tensor_raises_like_the_reference(pair):
    """`masks[i]` with zero regions is `None` in the reference's callers; an EMPTY [0, S, S] tensor makes its MaskPooling raise a
    RuntimeError (F.interpolate on an empty batch, base_extractor.py:53-58) -- here too, instead of a silent no-region image."""
    so, ocfg, w, model = pair
    r = _request(1006)  # one prompt, one image
    empty = [torch.zeros((0, CFG["image_size"], CFG["image_size"]))]
    with pytest.raises(RuntimeError):
        so.prepare_inputs(w, ocfg, r["input_ids"], r["images"], r["depths"], empty, r["attention_mask"])
    with pytest.raises(RuntimeError):
        model.engine.prepare_inputs(r["input_ids"].to(DEV), r["images"].to(DEV), r["depths"].to(DEV), [empty[0].to(DEV)], None)


@pytest.mark.parametrize("seed", list(range(max(2, int(os.environ.get("SRGPT_FUZZ_CASES", "10")) // 4))))
def test_random_true_width_request_equals_the_oracle_bf16(seed):
    """tests/test_gpu_edge_cases.py's true-width, truncated-depth bf16 comparison (stage tensors, prefill logits at every position,
    teacher-forced decode logits) on random request SHAPES: 1..3 prompts, 1..12 regions, prompts of 40..130 ids -- 235..325 spliced
    rows per prompt, across the row counts at which the prefill products change kernels (225 / 272 / 288 / 384 rows and their batch
    multiples) -- for the three LLM geometries and the fp8 weight format"""
    from tests import test_gpu_edge_cases as te
    g = torch.Generator().manual_seed(5000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    geom = ["vila15_8b", "llama2_7b", "sheared_3b", "vila15_8b-fp8", "clip_l14_336"][ri(0, 4)]
    regions = ri(1, 12)
    te.test_true_width_truncated_depth_bf16_vs_oracle(geom, batch=ri(1, 3), regions=regions, prompt_len=max(ri(40, 130), 3 * regions + 8),
                                                      seed=100 + seed)


@pytest.mark.parametrize("seed", list(range(max(2, int(os.environ.get("SRGPT_FUZZ_CASES", "10")) // 3))))
def test_ragged_prefill_random_lengths(seed):
    """srgpt_llm_prefill_ragged + batched decode == every row alone (tests/test_gpu_pipeline.py's check) for random batches of 1..6
    rows of 1..300 positions, fp32 and bf16"""
    from tests import test_gpu_pipeline as tp
    g = torch.Generator().manual_seed(6000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    lens = [[ri(1, 30), ri(20, 300), ri(200, 300)][ri(0, 2)] for _ in range(ri(1, 6))]
    tp.test_ragged_prefill_and_batched_decode_equal_single_rows([torch.float32, torch.bfloat16][ri(0, 1)], lens=lens)


def test_a_long_lived_model_answers_like_a_fresh_one():
    """State carried between requests (the pooled decode state and its cache capacity, the captured graphs, workspaces, the attention's
    arrival tickets, the sampler's counters) must not leak into the next answer: ONE model serves a random sequence of requests --
    batch 1..4, 0..4 regions, greedy / seeded sampling / 2 beams / a stopping criterion, 1..12 new tokens -- and every answer equals
    the answer of a model built fresh for that request alone (same weights).  tests/test_gpu_soak.py compares repeats of a request on
    the same model; this compares against a model with no history.  SRGPT_FUZZ_CASES scales the sequence length (default 30 calls)."""
    from oracle import srgpt_oracle as so  # weights only
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel

    ocfg = so.SrgptConfig(**CFG, padding_side="left")
    w = so.synth_weights(ocfg, seed=11, dtype=torch.float32, std=0.08)
    build = lambda: LlavaLlamaModel(SrgptConfig(**CFG, padding_side="left"), dict(w), device=DEV, dtype=torch.float32, rope_positions=512)  # noqa: E731
    veteran = build()
    g = torch.Generator().manual_seed(77)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    n_calls = 3 * int(os.environ.get("SRGPT_FUZZ_CASES", "10"))
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    for call in range(n_calls):
        r = _request(3000 + ri(0, 40))  # a small pool of requests: shapes recur with other shapes in between
        mode = ["greedy", "greedy", "sample", "beam", "criterion"][ri(0, 4)]
        n_new = ri(1, 12)
        kw = dict(do_sample=False)
        if mode == "sample":
            kw = dict(do_sample=True, temperature=0.8, top_k=20, top_p=0.9)
        elif mode == "beam":
            kw = dict(do_sample=False, num_beams=2)
        elif mode == "criterion":
            kw = dict(do_sample=False, stopping_criteria=[lambda ids_, s_: ids_.shape[1] >= 3])
        args = dict(images=dev(r["images"]), depths=dev(r["depths"]), masks=[dev(m) for m in r["masks"]],
                    attention_mask=dev(r["attention_mask"]), max_new_tokens=n_new, eos_token_id=None, **kw)
        torch.manual_seed(call)  # the sampler's Philox seed is drawn from torch's CPU generator
        a = veteran.generate(dev(r["input_ids"]), **args)
        torch.manual_seed(call)
        b = build().generate(dev(r["input_ids"]), **args)
        assert a.shape == b.shape and torch.equal(a, b), f"call {call} ({mode}, batch {r['input_ids'].shape[0]}, {n_new} new): {a.tolist()} vs fresh {b.tolist()}"


@pytest.mark.parametrize("fmt", ["native", "fp8"])
def test_a_long_lived_true_width_model_answers_like_a_fresh_one(fmt):
    """the same at TRUE width in bf16 / fp8 weights (2 LLM + 2 ViT layers, 16k vocabulary): the MFMA decode attention with its tickets,
    the batched products with their row-statistics tables and packed weights, the captured graphs of several batch sizes"""
    from oracle import srgpt_oracle as so  # weights / inputs only
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.model import LlavaLlamaModel

    kw = dict(vit_layers=2, layers=2, vocab=16386, mask_token_id=16384, depth_token_id=16385)
    ocfg = so.SrgptConfig(**kw)
    w = so.synth_weights(ocfg, seed=11, dtype=torch.bfloat16)
    build = lambda: LlavaLlamaModel(SrgptConfig(**kw), dict(w), device=DEV, dtype=torch.bfloat16, rope_positions=1024,  # noqa: E731
                                    llm_weight_format=fmt)
    veteran = build()
    g = torch.Generator().manual_seed(78)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    for call in range(int(os.environ.get("SRGPT_FUZZ_CASES", "10"))):
        B, K = [1, 1, 2, 3, 4, 8][ri(0, 5)], ri(1, 8)
        ids, images, depths, masks = so.synth_inputs(ocfg, batch=B, regions=K, prompt_len=max(ri(40, 100), 3 * K + 8), seed=ri(0, 5),
                                                     dtype=torch.bfloat16)
        mode = ["greedy", "greedy", "sample", "beam"][ri(0, 3)]
        kwg = dict(do_sample=False)
        if mode == "sample":
            kwg = dict(do_sample=True, temperature=0.8, top_k=20, top_p=0.9)
        elif mode == "beam":
            kwg = dict(do_sample=False, num_beams=2)
        args = dict(images=images.to(DEV), depths=depths.to(DEV), masks=[m.to(DEV) for m in masks], max_new_tokens=ri(1, 10),
                    eos_token_id=None, **kwg)
        torch.manual_seed(call)
        a = veteran.generate(ids.to(DEV), **args)
        torch.manual_seed(call)
        fresh = build()
        b = fresh.generate(ids.to(DEV), **args)
        del fresh
        assert a.shape == b.shape and torch.equal(a, b), f"call {call} ({mode}, batch {B}): {a.tolist()} vs fresh {b.tolist()}"
