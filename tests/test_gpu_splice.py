"""GPU: the device-side splice (csrc/splice.hip: srgpt_splice_plan + srgpt_splice_gather) against the host loop it replaced.

The edge cases are pinned to the REFERENCE by tests/golden/splice_kat.npz (row-source maps minted from llava_arch.py:333-650,
`test_splice_equals_the_reference_row_source_maps`).  `splice_reference` below IS rounds 1-4's implementation of the same index
arithmetic -- a per-token Python loop, itself checked against that fixture on CPU (tests/test_host_logic.py) -- kept for the
randomised shapes the fixture does not hold (2100-position prompts, 8 prompts).  Integer / byte work: every comparison is exact."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
IGNORE_INDEX, IMAGE_TOKEN_INDEX = -100, -200


def splice_reference(cfg, vocab, embed, ids_cpu, am_cpu, image_features, mask_embeds, depth_embeds, have_depths, labels=None):
    """host loop -> (out [B,T,H] (CPU), attention mask bool [B,T], lens, labels [B,T])"""
    B, P = ids_cpu.shape
    am_cpu = torch.ones_like(ids_cpu, dtype=torch.bool) if am_cpu is None else am_cpu.bool()
    nimg_feat = image_features.shape[1]
    cur_image_idx = 0
    mask_off, n_me = [], 0
    if mask_embeds is not None:
        for e in mask_embeds:
            mask_off.append(n_me)
            n_me += 0 if e is None else e.shape[0]
    seqs, labs = [], []
    for b in range(B):
        cur = ids_cpu[b][am_cpu[b]].tolist()
        cur_lab = [IGNORE_INDEX] * len(cur) if labels is None else labels[b][am_cpu[b]].tolist()
        n_images = sum(1 for t in cur if t == IMAGE_TOKEN_INDEX)
        if n_images == 0:
            seqs.append([("t", t) for t in cur])
            labs.append(cur_lab)
            continue
        seq, lab = [], []
        first_img = cur_image_idx
        nm = nd = 0
        n_mask_tok = sum(1 for t in cur if t == cfg.mask_token_id)
        n_depth_tok = sum(1 for t in cur if t == cfg.depth_token_id)
        me = mask_embeds[first_img] if (cfg.enable_region and mask_embeds is not None) else None
        de = depth_embeds[first_img] if (cfg.enable_region and cfg.enable_depth and have_depths and depth_embeds is not None) else None
        if me is not None and n_mask_tok > me.shape[0]:
            raise RuntimeError("shape mismatch")
        if de is not None and n_depth_tok > de.shape[0]:
            raise RuntimeError("shape mismatch")
        for t, tl in zip(cur, cur_lab):
            if t == IMAGE_TOKEN_INDEX:
                seq.extend(("i", cur_image_idx, r) for r in range(nimg_feat))
                lab.extend([IGNORE_INDEX] * nimg_feat)
                cur_image_idx += 1
                continue
            lab.append(tl)
            if me is not None and t == cfg.mask_token_id:
                seq.append(("m", mask_off[first_img] + nm))
                nm += 1
            elif de is not None and t == cfg.depth_token_id:
                seq.append(("d", mask_off[first_img] + nd))
                nd += 1
            else:
                seq.append(("t", t))
        seqs.append(seq)
        labs.append(lab)
    mx = cfg.tokenizer_model_max_length
    if mx is not None:
        seqs = [s[:mx] for s in seqs]
        labs = [l[:mx] for l in labs]
    lens = [len(s) for s in seqs]
    T = max(lens)
    left = cfg.padding_side == "left"
    H = embed.shape[1]
    out = torch.zeros((B, T, H), dtype=embed.dtype)
    am = torch.zeros((B, T), dtype=torch.bool)
    new_labels = torch.full((B, T), IGNORE_INDEX, dtype=torch.int64)
    feats = image_features.reshape(-1, H)
    me_all = torch.cat([e for e in mask_embeds if e is not None], 0) if mask_embeds and any(e is not None for e in mask_embeds) else None
    de_all = torch.cat([e for e in depth_embeds if e is not None], 0) if depth_embeds and any(e is not None for e in depth_embeds) else None
    for b, seq in enumerate(seqs):
        off = T - len(seq) if left else 0
        for j, item in enumerate(seq):
            if item[0] == "t":
                assert 0 <= item[1] < vocab
                out[b, off + j] = embed[item[1]]
            elif item[0] == "i":
                out[b, off + j] = feats[item[1] * nimg_feat + item[2]]
            elif item[0] == "m":
                out[b, off + j] = me_all[item[1]]
            else:
                out[b, off + j] = de_all[item[1]]
        am[b, off:off + len(seq)] = True
        if labs[b]:
            new_labels[b, off:off + len(seq)] = torch.tensor(labs[b], dtype=torch.int64)
    return out, am, lens, new_labels


@pytest.fixture(scope="module")
def eng():
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine
    from spatialrgpt_amd.weights import synth_state_dict

    cfg = SrgptConfig(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=42, patch_size=14, hidden=64, inter=128,
                      layers=1, heads=4, kv_heads=2, vocab=300, mask_token_id=298, depth_token_id=299)
    return SrgptEngine(cfg, synth_state_dict(cfg, seed=2, dtype=torch.float32, device=DEV), device=DEV, dtype=torch.float32,
                       rope_positions=64)


def _case(eng, seed, B, P, n_img_per, masks_per_img, side="right", mx=None, with_am=False, with_labels=False, have_depths=True,
          none_masks=()):
    """random prompts: prompt b owns n_img_per[b] images; <mask>/<depth> ids sprinkled in; some positions masked out"""
    cfg = eng.cfg
    g = torch.Generator().manual_seed(seed)
    nf, H = 5, cfg.hidden
    n_images = sum(n_img_per)
    ids = torch.randint(3, 290, (B, P), generator=g)
    for b in range(B):
        pos = torch.randperm(P, generator=g).tolist()
        for _ in range(n_img_per[b]):
            ids[b, pos.pop()] = IMAGE_TOKEN_INDEX
        for _ in range(int(torch.randint(0, masks_per_img + 1, (1,), generator=g))):
            ids[b, pos.pop()] = cfg.mask_token_id
        for _ in range(int(torch.randint(0, masks_per_img + 1, (1,), generator=g))):
            ids[b, pos.pop()] = cfg.depth_token_id
    am = (torch.rand((B, P), generator=g) > 0.15) if with_am else None
    if am is not None:
        # the reference raises when a prompt's <mask> ids outnumber its embeddings; masked-out sentinels change which image a prompt owns
        am = am | (ids == IMAGE_TOKEN_INDEX)
    labels = torch.randint(0, 290, (B, P), generator=g) if with_labels else None
    feats = torch.randn((n_images, nf, H), generator=g)
    me = [None if i in none_masks else torch.randn((masks_per_img, H), generator=g) for i in range(n_images)]
    de = [None if i in none_masks else torch.randn((masks_per_img, H), generator=g) for i in range(n_images)] if have_depths else None
    cfg.padding_side, cfg.tokenizer_model_max_length = side, mx
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = eng.splice(ids.to(DEV), None if am is None else am.to(DEV), feats.to(DEV), [None if e is None else e.to(DEV) for e in me],
                             None if de is None else [None if e is None else e.to(DEV) for e in de], have_depths,
                             labels=None if labels is None else labels.to(DEV))
        ref = splice_reference(cfg, eng.w.vocab, eng.w.embed.cpu(), ids, am, feats, me, de, have_depths, labels)
    finally:
        cfg.padding_side, cfg.tokenizer_model_max_length = "right", None
    assert got[2] == ref[2], (got[2], ref[2])
    assert torch.equal(got[0].cpu(), ref[0])
    if am is not None:
        assert got[1].dtype == am.dtype and torch.equal(got[1].cpu(), ref[1])
    else:
        assert got[1] is None
    if labels is not None:
        assert torch.equal(got[3].cpu(), ref[3])


@pytest.mark.parametrize("side", ["right", "left"])
def test_splice_matches_the_host_loop(eng, side):
    _case(eng, 1, 1, 40, [1], 3, side)                                                    # the benchmark's shape: one prompt, one image
    _case(eng, 2, 4, 64, [1, 1, 1, 1], 4, side, with_am=True, with_labels=True)           # ragged through the attention mask
    _case(eng, 3, 3, 50, [2, 0, 1], 3, side, with_labels=True)                            # a text-only prompt between image prompts
    _case(eng, 4, 3, 50, [0, 3, 0], 2, side, with_am=True)                                # several images in one prompt, none before it
    _case(eng, 5, 2, 33, [1, 1], 3, side, have_depths=False, with_labels=True)            # depths=None: only <mask> ids are replaced
    _case(eng, 6, 3, 45, [1, 1, 1], 3, side, none_masks=(1,), with_am=True)               # masks[1] is None: its ids stay text
    _case(eng, 7, 2, 70, [1, 2], 3, side, mx=31, with_labels=True, with_am=True)          # cut at tokenizer_model_max_length
    _case(eng, 8, 5, 2100, [1, 0, 2, 1, 1], 6, side, with_am=True, with_labels=True)      # > 1024 positions: several per thread
    _case(eng, 9, 8, 64, [1] * 8, 8, side)                                                # configs[4]'s shape


def test_splice_equals_the_reference_row_source_maps():
    """The device splice against tests/golden/splice_kat.npz: the REFERENCE's row-source maps (llava_arch.py:333-650 run on
    index-coded rows by oracle/make_golden.py `splice`) for several images per prompt, text-only rows, None mask entries, surplus /
    missing region embeddings, depths=None, left padding, truncation, attention masks with holes, labels.  Exact."""
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine
    from spatialrgpt_amd.weights import synth_state_dict
    from tests.util import splice_kat_cases, splice_kat_tables

    engine, n = None, 0
    for c in splice_kat_cases():
        if engine is None:
            cfg = SrgptConfig(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=42, patch_size=14, hidden=64,
                              inter=128, layers=1, heads=4, kv_heads=2, vocab=c["vocab"], mask_token_id=c["mask_token_id"],
                              depth_token_id=c["depth_token_id"])
            engine = SrgptEngine(cfg, synth_state_dict(cfg, seed=3, dtype=torch.float32, device=DEV), device=DEV,
                                 dtype=torch.float32, rope_positions=64)
        cfg = engine.cfg
        embed, feats, me, de, expect = splice_kat_tables(c, cfg.hidden, embed=engine.w.embed.cpu(), seed=n)
        n += 1
        dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
        cfg.padding_side, cfg.tokenizer_model_max_length = c["padding_side"], c["max_length"]
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                args = (dev(c["input_ids"]), dev(c["attention_mask"]), dev(feats), [dev(e) for e in me],
                        None if de is None else [dev(e) for e in de], c["have_depths"])
                if c["raises"]:
                    with pytest.raises(RuntimeError, match="shape mismatch"):
                        engine.splice(*args, labels=dev(c["labels"]))
                    continue
                got = engine.splice(*args, labels=dev(c["labels"]))
        finally:
            cfg.padding_side, cfg.tokenizer_model_max_length = "right", None
        assert torch.equal(got[0].cpu(), expect()), c["name"]
        if c["attention_mask_out"] is None:
            assert got[1] is None, c["name"]
        else:
            assert got[1].dtype == c["attention_mask"].dtype and torch.equal(got[1].cpu().bool(), c["attention_mask_out"]), c["name"]
        assert got[2] == [int((c["src_kind"][b] >= 0).sum()) for b in range(c["src_kind"].shape[0])], c["name"]
        if c["labels"] is not None:
            assert torch.equal(got[3].cpu(), c["new_labels"]), c["name"]
    assert n >= 14


def test_splice_errors_match_the_reference(eng):
    cfg = eng.cfg
    H = cfg.hidden
    feats = torch.randn((1, 5, H), device=DEV)
    ids = torch.tensor([[5, IMAGE_TOKEN_INDEX, cfg.mask_token_id, cfg.mask_token_id, 7]], device=DEV)
    with pytest.raises(RuntimeError, match="shape mismatch"):   # two <mask> ids, one embedding (boolean-index assignment error upstream)
        eng.splice(ids, None, feats, [torch.randn((1, H), device=DEV)], None, False)
    with pytest.raises(IndexError):                             # an id outside the embedding table
        eng.splice(torch.tensor([[5, IMAGE_TOKEN_INDEX, 300]], device=DEV), None, feats, [None], None, False)
    with pytest.raises(IndexError):                             # two sentinels, one image
        eng.splice(torch.tensor([[IMAGE_TOKEN_INDEX, 4, IMAGE_TOKEN_INDEX]], device=DEV), None, feats, [None], None, False)
    # <mask> ids without embeddings: a printed complaint, the ids embed as text (llava_arch.py:470-505)
    out, _, lens = eng.splice(torch.tensor([[5, IMAGE_TOKEN_INDEX, cfg.mask_token_id]], device=DEV), None, feats, [None], None, False)
    assert lens == [7] and torch.equal(out[0, 6], eng.w.embed[cfg.mask_token_id])
