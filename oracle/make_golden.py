"""TEST INFRASTRUCTURE ONLY.  Mints tests/golden/*.npz from the REAL reference run on CPU.

Run in the build container (needs /root/reference):   python oracle/make_golden.py

What it does
  1. builds the reference `LlavaLlamaModel` (tiny, seeded random weights) through the reference's own
     builders (oracle/ref_harness.py), fp32, eager attention;
  2. runs the reference hot path -- `prepare_inputs_labels_for_multimodal` (llava_arch.py:333) with
     forward hooks on every stage boundary of SURVEY 8a, `llm(inputs_embeds=...)` for prefill logits /
     hidden states, and `model.generate(...)` (llava_llama.py:194) greedy;
  3. asserts that the self-contained restatement oracle/srgpt_oracle.py reproduces every stage and the
     generated ids from the same weights and inputs  (this is what PINS the oracle);
  4. writes weights (canonical checkpoint key names), inputs, stage outputs and ids to
     tests/golden/tiny_fp32.npz, a bf16 run of the same model to tests/golden/tiny_bf16.npz, and
     model-free MaskPooling / DownSampleBlock / LayerNorm2d / deconv known-answer vectors
     (reference modules loaded standalone) to tests/golden/region_kat.npz.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness as rh  # noqa: E402
from oracle import srgpt_oracle as so  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

TINY_LLM = dict(vocab_size=128, hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=500000.0,
                tie_word_embeddings=False, bos_token_id=1, eos_token_id=2, attention_bias=False)
TINY_VIT = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=3, num_attention_heads=4, image_size=378,
                patch_size=14, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh")


TINY_CLIP = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=336,
                 patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu", projection_dim=32)


def canonical_state_dict(model) -> dict:
    """reference in-memory keys -> checkpoint (transformers 4.37.2) key names."""
    out = {}
    for k, v in model.state_dict().items():
        if k.startswith("vision_tower.vision_tower.") and not k.startswith("vision_tower.vision_tower.vision_model."):
            k = k.replace("vision_tower.vision_tower.", "vision_tower.vision_tower.vision_model.", 1)
        if ".head." in k or "post_layernorm" in k:
            continue  # pooling head / post-LN are computed and discarded by the reference
        out[k] = v.detach().clone()
    return out


def cfg_from(model, tok) -> so.SrgptConfig:
    lc, vc = model.llm.config, model.get_vision_tower().vision_tower.config
    return so.SrgptConfig(
        vit_hidden=vc.hidden_size, vit_inter=vc.intermediate_size, vit_layers=vc.num_hidden_layers,
        vit_heads=vc.num_attention_heads, image_size=vc.image_size, patch_size=vc.patch_size,
        vit_eps=vc.layer_norm_eps, select_layer=-2,
        select_feature=model.get_vision_tower().select_feature,
        tower="clip" if "clip" in type(model.get_vision_tower().vision_tower).__name__.lower() else "siglip",
        hidden=lc.hidden_size, inter=lc.intermediate_size, layers=lc.num_hidden_layers,
        heads=lc.num_attention_heads, kv_heads=lc.num_key_value_heads, vocab=model.llm.get_input_embeddings().weight.shape[0],
        rms_eps=lc.rms_norm_eps, rope_theta=float(lc.rope_parameters["rope_theta"]) if hasattr(lc, "rope_parameters") else float(lc.rope_theta),
        mask_token_id=model.get_vision_tower().config.llm_mask_token_id,
        depth_token_id=model.get_vision_tower().config.llm_depth_token_id,
        enable_region=True, enable_depth=True,
        tokenizer_model_max_length=getattr(lc, "tokenizer_model_max_length", None),
        padding_side=getattr(lc, "tokenizer_padding_side", "right"), eos_token_id=None,
    )


def run_reference(model, ids, images, depths, masks, max_new_tokens):
    """Run the reference path with hooks; returns dict of stage tensors + generated ids."""
    st = {}
    hooks = []
    vt = model.get_vision_tower()
    tower_calls = []
    hooks.append(vt.register_forward_hook(lambda m, i, o: tower_calls.append(o.detach().clone())))
    rex = model.get_region_extractor()
    orig_fr = rex.feature_refinement

    def fr(x):
        h, l = orig_fr(x)
        st["hres"], st["lres"] = h.detach().clone(), l.detach().clone()
        return h, l

    rex.feature_refinement = fr
    hooks.append(rex.register_forward_hook(lambda m, i, o: st.update(
        mask_embeds=torch.stack([e.detach() for e in o[0]]), depth_embeds=torch.stack([e.detach() for e in o[1]]))))
    hooks.append(model.get_mm_projector().register_forward_hook(lambda m, i, o: st.update(image_features=o.detach().clone())))
    with torch.no_grad():
        (_, pos, am, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(
            ids, None, None, None, None, images, masks, depths)
        st["tower_features"], st["depth_features"] = tower_calls[0], tower_calls[1]
        st["inputs_embeds"] = embeds.detach().clone()
        out = model.llm(inputs_embeds=embeds.to(model.dtype), output_hidden_states=True, use_cache=False)
        st["prefill_logits"] = out.logits.float().detach().clone()
        st["hidden_states"] = torch.stack([h.detach() for h in out.hidden_states])  # last one is post-final-norm
    for h in hooks:
        h.remove()
    rex.feature_refinement = orig_fr
    with torch.no_grad():
        gen = model.generate(input_ids=ids, images=images, depths=depths, masks=masks, do_sample=False,
                             max_new_tokens=max_new_tokens, use_cache=True, eos_token_id=None, pad_token_id=0,
                             min_new_tokens=max_new_tokens)
        gen2 = model.generate(input_ids=ids, images=images, depths=depths, masks=masks, do_sample=False,
                              max_new_tokens=max_new_tokens, use_cache=True, eos_token_id=None, pad_token_id=0,
                              min_new_tokens=max_new_tokens)
    assert torch.equal(gen, gen2), "reference generate() is not deterministic"
    st["new_ids"] = gen
    return st


def check(name, a, b, atol, rtol=0.0):
    a, b = a.float(), b.float()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    ok = err <= atol + rtol * ref
    print(f"  {name:18s} max|d|={err:.3e} (ref max {ref:.3e}) {'OK' if ok else 'MISMATCH'}")
    assert ok, name


def tensor_np(t):
    t = t.detach()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)  # raw bf16 bits
    return t.numpy()


def mint_model_case(dtype: torch.dtype, fname: str, max_new_tokens=12, tower="siglip"):
    print(f"== {fname}")
    with tempfile.TemporaryDirectory() as td:
        if tower == "clip":  # CLIP-L/336 style: 577 tokens, select_feature "patch" -> 576 = 24^2
            model, tok = rh.build_tiny_reference_model(td, llm=TINY_LLM, vit=TINY_CLIP, dtype="torch.float32", seed=0,
                                                       tower="clip", select_feature="patch")
        else:
            model, tok = rh.build_tiny_reference_model(td, llm=TINY_LLM, vit=TINY_VIT, dtype="torch.float32", seed=0)
    # make norm gains / biases non-trivial so that parity checks see them
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or "layernorm" in n or "layer_norm" in n or n.endswith("module.1.weight") \
                    or n == "mm_projector.layers.1.weight":
                if n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    model = model.to(dtype)
    cfg = cfg_from(model, tok)
    w = canonical_state_dict(model)
    ids, images, depths, masks = so.synth_inputs(cfg, batch=1, regions=2, prompt_len=15, seed=1, dtype=dtype)
    ref = run_reference(model, ids, images, depths, masks, max_new_tokens)
    # ---- pin the restatement against the reference
    new_ids, st = so.generate(w, cfg, ids, images, depths, masks, max_new_tokens=max_new_tokens, return_stages=True,
                              model_dtype=dtype)
    tol = 1e-5 if dtype == torch.float32 else 0.0
    rt = 1e-5 if dtype == torch.float32 else 2e-2
    for k in ("tower_features", "depth_features", "hres", "lres", "image_features", "inputs_embeds"):
        check(k, st[k], ref[k], tol, rt)
    check("mask_embeds", torch.stack(st["mask_embeds"]), ref["mask_embeds"], tol, rt)
    check("depth_embeds", torch.stack(st["depth_embeds"]), ref["depth_embeds"], tol, rt)
    check("prefill_logits", st["prefill_logits"], ref["prefill_logits"], 1e-4 if dtype == torch.float32 else 0.0, rt)
    if dtype == torch.float32:
        assert torch.equal(new_ids, ref["new_ids"]), (new_ids, ref["new_ids"])
        print("  greedy ids identical:", new_ids.tolist())
    else:
        agree = (new_ids == ref["new_ids"]).float().mean().item()
        print(f"  greedy ids agreement (bf16, informational): {agree:.2f}")
    assert ref["inputs_embeds"].shape[1] == 15 - 1 + 196
    # ---- write
    blob = {"cfg_json": np.frombuffer(json.dumps(cfg.to_dict()).encode(), dtype=np.uint8),
            "dtype": np.frombuffer(str(dtype).encode(), dtype=np.uint8),
            "in.input_ids": ids.numpy(),
            "in.images_q32": (images.float() * 32).round().to(torch.int8).numpy(),   # pixel = q / 32
            "in.depths_q32": (depths.float()[:, :1] * 32).round().to(torch.int8).numpy(),  # 1 channel, x3
            "in.masks_u8": torch.stack(masks).float().to(torch.uint8).numpy()}
    for k, v in w.items():
        blob["w." + k] = tensor_np(v)
    for k, v in ref.items():
        if k == "hres":  # 11664 x C: keep every 7th row (+ the last) to keep the fixture small
            idx = torch.cat([torch.arange(0, v.shape[1], 7), torch.tensor([v.shape[1] - 1])])
            blob["ref.hres_rows_idx"] = idx.numpy()
            blob["ref.hres_rows"] = tensor_np(v[:, idx].contiguous())
            continue
        blob["ref." + k] = tensor_np(v)
    # per-step teacher-forced logits from the restatement (== reference ids in fp32)
    blob["ref.step_logits"] = tensor_np(st["step_logits"])
    np.savez_compressed(os.path.join(GOLD, fname), **blob)
    print("  wrote", fname, f"{os.path.getsize(os.path.join(GOLD, fname)) / 1e6:.2f} MB")


def _load_standalone(relpath, name):
    import importlib.util

    rh.install_shims()
    spec = importlib.util.spec_from_file_location(name, os.path.join(rh.REFERENCE_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def mint_region_kat():
    """Model-free known-answer vectors straight from the reference modules (SURVEY 8c iii)."""
    print("== region_kat.npz")
    be = _load_standalone("llava/model/region_extractor/base_extractor.py", "_ref_base_extractor")
    bp = _load_standalone("llava/model/multimodal_projector/base_projector.py", "_ref_base_projector")
    g = torch.Generator().manual_seed(7)
    blob = {}
    mp = be.MaskPooling()
    C = 16

    def case(tag, L, S, masks, dtype=torch.float32):
        # features = int8 / 64: exact in fp32 and bf16, stored as int8
        feat_q = torch.randint(-128, 128, (1, L, C), generator=g, dtype=torch.int16).to(torch.int8)
        feat = (feat_q.float() / 64).to(dtype)
        with torch.no_grad():
            out = mp(feat, [masks.to(dtype)], return_list=True)[0]
        mine = so.mask_pooling(feat, [masks.to(dtype)])[0]
        check("pool/" + tag, mine, out, 1e-6 if dtype == torch.float32 else 0.0, 0 if dtype == torch.float32 else 1e-2)
        blob[f"pool.{tag}.feat_q64"] = feat_q[0].numpy()
        blob[f"pool.{tag}.masks_q16"] = (masks * 16).round().to(torch.uint8).numpy()  # mask = q / 16
        blob[f"pool.{tag}.out"] = tensor_np(out)

    S = 384
    boxes = torch.zeros((5, S, S))
    boxes[0, 10:200, 30:90] = 1
    boxes[1, 300:384, 0:384] = 1
    boxes[2, 100:101, 200:201] = 1  # single pixel
    boxes[3] = 1  # all ones -> mean of features
    # boxes[4] stays empty -> zeros (denorm = 1e-8)
    case("rgb108", 108 * 108, S, boxes)
    case("depth27", 27 * 27, S, boxes)
    soft = torch.randint(0, 17, (3, 336, 336), generator=g).float() / 16
    case("soft336_to_108", 108 * 108, 336, soft)
    case("soft336_to_96", 96 * 96, 336, soft)
    case("rgb108_bf16", 108 * 108, S, boxes, torch.bfloat16)
    small = (torch.rand((4, 56, 56), generator=g) > 0.5).float()
    case("up56_to_108", 108 * 108, 56, small)  # upsampling branch of the bilinear map

    # DownSampleBlock ordering
    x = torch.randn((2, 729, 8), generator=g)
    with torch.no_grad():
        y = bp.DownSampleBlock()(x)
    n, L, c = x.shape
    mine = so.flat_square(x.reshape(n, 27, 27, c)).reshape(n, -1, 4 * c)
    assert torch.equal(mine, y)
    blob["s2d.in"], blob["s2d.out"] = x.numpy(), y.numpy()

    # feature refinement module (deconv -> LN2d -> GELU -> deconv -> GELU) + adaptive pool, 27->108 and 24->96
    C = 16
    for tag, hs in (("27", 27), ("24", 24)):
        torch.manual_seed(11)
        frm = be.get_feature_refinement_module(C)
        with torch.no_grad():
            frm[1].weight.copy_(1 + 0.1 * torch.randn(C, generator=g))
            frm[1].bias.copy_(0.1 * torch.randn(C, generator=g))
        feats = torch.randn((1, hs * hs, C), generator=g)
        with torch.no_grad():
            h = frm(feats.reshape(1, hs, hs, C).permute(0, 3, 1, 2))
            l = torch.nn.AdaptiveAvgPool2d(27)(h)
        w = {so.RE + "feature_refinement_module." + k: v for k, v in frm.state_dict().items()}
        mh, ml = so.feature_refinement(w, feats)
        check("refine/hres" + tag, mh, h.flatten(2).transpose(1, 2), 1e-5)
        check("refine/lres" + tag, ml, l.flatten(2).transpose(1, 2), 1e-5)
        blob[f"refine{tag}.in"] = feats.numpy()
        for k, v in frm.state_dict().items():
            blob[f"refine{tag}.w.{k}"] = v.numpy()
        hf = h.flatten(2).transpose(1, 2).contiguous()
        idx = torch.cat([torch.arange(0, hf.shape[1], 5), torch.tensor([hf.shape[1] - 1])])
        blob[f"refine{tag}.hres_rows_idx"] = idx.numpy()
        blob[f"refine{tag}.hres_rows"] = hf[:, idx].contiguous().numpy()
        blob[f"refine{tag}.lres"] = l.flatten(2).transpose(1, 2).contiguous().numpy()
    np.savez_compressed(os.path.join(GOLD, "region_kat.npz"), **blob)
    print("  wrote region_kat.npz", f"{os.path.getsize(os.path.join(GOLD, 'region_kat.npz')) / 1e6:.2f} MB")


def mint_tokenizer_kat():
    """tokenizer_image_token (llava/mm_utils.py:545-570) against a BOS-adding fake tokenizer."""
    print("== tokenizer_kat.json")
    rh.install_shims()
    from llava.mm_utils import tokenizer_image_token

    class FakeTok:
        bos_token_id = 1

        def __call__(self, text):
            class R:
                pass

            r = R()
            r.input_ids = [1] + [3 + (sum(map(ord, wd)) % 90) for wd in text.split()]
            return r

    cases = ["aa bbb <image>\ncccc d", "<image>\ncccc <image> d", "no image here", "<image>", "x <image> y <image> z <image>"]
    out = []
    for c in cases:
        out.append({"prompt": c, "ids": tokenizer_image_token(c, FakeTok()),
                    "ids_lstrip": tokenizer_image_token(c, FakeTok(), lstrip=True)})
    with open(os.path.join(GOLD, "tokenizer_kat.json"), "w") as f:
        json.dump(out, f, indent=1)


def mint_labels_kat():
    """Teacher-forced forward with labels (llava_llama.py:100-192 -> LlamaForCausalLM loss): the reference's spliced labels,
    attention mask / position ids and its loss for a ragged batch of two samples, same tiny fp32 model as tiny_fp32.npz
    (identical seeds -> identical weights; the fixture stores only inputs and reference outputs)."""
    print("== labels_kat.npz")
    dtype = torch.float32
    with tempfile.TemporaryDirectory() as td:
        model, tok = rh.build_tiny_reference_model(td, llm=TINY_LLM, vit=TINY_VIT, dtype="torch.float32", seed=0)
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or "layernorm" in n or "layer_norm" in n or n.endswith("module.1.weight") \
                    or n == "mm_projector.layers.1.weight":
                if n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    cfg = cfg_from(model, tok)
    w = canonical_state_dict(model)
    gold = np.load(os.path.join(GOLD, "tiny_fp32.npz"))
    for k, v in w.items():  # the fixture reuses tiny_fp32.npz's weights: they must be the same model
        assert np.array_equal(gold["w." + k], tensor_np(v)), k
    ids, images, depths, masks = so.synth_inputs(cfg, batch=1, regions=2, prompt_len=15, seed=1, dtype=dtype)
    # ragged batch of 2 (right padded): sample 1 drops the last 4 ids; labels supervise the second half of each prompt
    P = ids.shape[1]
    ids_b = torch.cat([ids, ids], 0)
    am = torch.ones_like(ids_b)
    am[1, P - 4:] = 0
    ids_b[1, P - 4:] = 0
    labels = ids_b.clone()
    labels[:, :6] = so.IGNORE_INDEX
    labels[ids_b == so.IMAGE_TOKEN_INDEX] = so.IGNORE_INDEX
    labels[am == 0] = so.IGNORE_INDEX
    im2, dp2, mk2 = torch.cat([images, images], 0), torch.cat([depths, depths], 0), [masks[0], masks[0]]
    with torch.no_grad():
        (_, pos, am_out, _, embeds, new_labels) = model.prepare_inputs_labels_for_multimodal(
            ids_b, None, am, None, labels, im2, mk2, dp2)
        out = model(input_ids=ids_b, images=im2, masks=mk2, depths=dp2, attention_mask=am, labels=labels)
    loss_ref = float(out.loss)
    # the restatement
    image_features, mask_embeds, depth_embeds, _ = so.encode_visual(w, cfg, im2, dp2, mk2)
    o_emb, o_am, o_pid, o_lab = so.splice(w, cfg, ids_b, am, image_features, mask_embeds, depth_embeds, have_depths=True,
                                          labels=labels)
    assert torch.equal(o_lab, new_labels), "spliced labels differ from the reference"
    assert torch.equal(o_am, am_out) and pos is None  # the reference returns position_ids only when the caller passed them
    check("inputs_embeds", o_emb, embeds, 1e-5, 1e-5)
    kv = so.KVCache(cfg.layers)
    logits = so.llama_forward(w, cfg, o_emb, o_pid, kv, key_padding_mask=o_am.bool())
    loss_o = float(so.causal_lm_loss(logits, o_lab))
    print(f"  loss reference {loss_ref:.6f}  oracle {loss_o:.6f}")
    assert abs(loss_o - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref))
    valid = o_am.bool()  # padded query rows are unspecified (nobody reads them): compare the valid positions only
    check("logits", logits[valid], out.logits.float()[valid], 1e-4, 1e-5)
    np.savez_compressed(os.path.join(GOLD, "labels_kat.npz"), input_ids=ids_b.numpy(), attention_mask=am.numpy(),
                        labels=labels.numpy(), new_labels=new_labels.numpy(),
                        attention_mask_out=am_out.numpy(), loss=np.float64(loss_ref),
                        logits_valid_last=torch.stack([out.logits.float()[b, int(o_am[b].sum()) - 1] for b in range(2)]).numpy())
    print("  wrote labels_kat.npz")


def mint_splice_kat():
    """The token-stream splice's edge cases, pinned to the reference's OWN `prepare_inputs_labels_for_multimodal`
    (llava_arch.py:333-650) as an exact ROW-SOURCE MAP: forward hooks replace what the projector / region extractor return and
    the embedding table by index-coded rows (value = 1 + row number, per source kind), so every row of the reference's
    `inputs_embeds` names the row it was copied from -- integer work, compared exactly.  Cases: several `-200` per prompt
    (:503-539), `num_images == 0` rows (:456-463), `masks[i] is None` with `<mask>` ids present (:476-477), more embeddings than
    `<mask>` ids (:480-485), fewer (the boolean-index shape error), `depths=None` (:406-407), left padding (:575-600), truncation
    (:541-547), an attention mask with interior holes (:439-446), list / 5-D `images` (:387-396), labels (:430-431, :513-533)."""
    print("== splice_kat.npz")
    with tempfile.TemporaryDirectory() as td:
        model, tok = rh.build_tiny_reference_model(td, llm=TINY_LLM, vit=TINY_VIT, dtype="torch.float32", seed=0)
    cfg = cfg_from(model, tok)
    MASK, DEPTH, IMG = cfg.mask_token_id, cfg.depth_token_id, so.IMAGE_TOKEN_INDEX
    V = model.llm.model.embed_tokens.weight.shape[0]
    H = cfg.hidden
    NF = 196
    S = cfg.image_size
    KIND_TEXT, KIND_IMAGE, KIND_MASK, KIND_DEPTH = 0, 1, 2, 3

    def coded(n):  # [n, H] rows whose every element is 1 + row number (exact in fp32)
        return (torch.arange(n, dtype=torch.float32) + 1.0)[:, None].expand(n, H).contiguous()

    with torch.no_grad():
        model.llm.model.embed_tokens.weight.copy_(coded(V))
    state = {}
    hooks = [
        model.get_mm_projector().register_forward_hook(
            lambda m, i, o: (coded(o.shape[0] * o.shape[1]) + 1e5 * KIND_IMAGE).reshape(o.shape)),
    ]

    def rex_hook(m, i, o):
        me, de = o
        off, me2, de2 = 0, [], (None if de is None else [])
        for k, e in enumerate(me):
            n = 0 if e is None else e.shape[0]
            me2.append(None if e is None else coded(off + n)[off:] + 1e5 * KIND_MASK)
            if de is not None:
                de2.append(None if de[k] is None else coded(off + n)[off:] + 1e5 * KIND_DEPTH)
            off += n
        return me2, de2

    hooks.append(model.get_region_extractor().register_forward_hook(rex_hook))
    g = torch.Generator().manual_seed(7)

    def text(n):
        return torch.randint(3, min(MASK, DEPTH, V - 8), (n,), generator=g).tolist()

    def img(n):
        return (torch.randn((n, 3, S, S), generator=g).clamp_(-1, 1) * 32).round().div(32)

    def boxes(k):
        m = torch.zeros((k, S, S))
        for r in range(k):
            m[r, 10 * r:10 * r + 60, 20:120] = 1.0
        return m

    def pad(rows, fill=0):
        P = max(len(r) for r in rows)
        ids = torch.full((len(rows), P), fill, dtype=torch.long)
        am = torch.zeros((len(rows), P), dtype=torch.long)
        for b, r in enumerate(rows):
            ids[b, :len(r)] = torch.tensor(r)
            am[b, :len(r)] = 1
        return ids, am

    R = lambda k: sum(([MASK, DEPTH] + text(1) for _ in range(k)), [])  # noqa: E731
    cases = []

    def add(name, rows, n_masks, *, am="pad", labels=False, depths=True, side="right", mx=None, images_as="tensor", holes=None):
        ids, amask = pad(rows)
        n_img = int((ids == IMG).sum())
        if holes:
            for (b, j) in holes:
                amask[b, j] = 0
        cases.append(dict(name=name, ids=ids, am=None if am is None else amask, labels=labels, depths=depths, side=side, mx=mx,
                          n_masks=n_masks, n_img=n_img, images_as=images_as))

    add("one_image", [[1] + text(3) + [IMG] + text(2) + R(2) + text(2)], [2], am=None)
    add("two_images_in_one_prompt", [[1] + text(2) + [IMG] + R(1) + [IMG] + R(2) + text(1)], [3, 1], am=None, labels=True)
    add("text_only_between_image_prompts", [[1, IMG] + R(2) + text(2), [1] + text(6), [1] + text(1) + [IMG] + R(1)], [2, 1], labels=True)
    add("left_padding", [[1, IMG] + R(2) + text(4), [1] + text(2) + [IMG] + R(1)], [2, 1], side="left", labels=True)
    add("truncation", [[1] + text(2) + [IMG] + R(2) + text(3), [1, IMG] + text(2)], [2, 1], mx=150, labels=True)
    add("truncation_left", [[1] + text(2) + [IMG] + R(2) + text(3), [1, IMG] + text(2)], [2, None], mx=150, side="left")
    add("mask_entry_is_None", [[1, IMG] + R(2) + text(1), [1, IMG] + R(1) + text(2)], [2, None])
    add("more_embeddings_than_mask_ids", [[1, IMG] + R(1) + text(2)], [3], am=None)
    add("fewer_embeddings_than_mask_ids", [[1, IMG] + R(3) + text(2)], [2], am=None)
    add("depths_None", [[1, IMG] + R(2) + text(2), [1] + text(1) + [IMG] + R(1)], [2, 1], depths=False, labels=True)
    add("attention_mask_interior_holes", [[1] + text(2) + [IMG] + R(2) + text(3), [1, IMG] + R(1) + text(5)], [2, 1],
        holes=[(0, 1), (0, 6), (1, 4)], labels=True)
    add("hole_over_a_mask_id", [[1, IMG] + R(2) + text(3)], [2], holes=[(0, 2)])
    add("images_as_list", [[1, IMG] + R(1) + text(2), [1, IMG] + R(1)], [1, 1], images_as="list")
    add("images_5d", [[1, IMG, IMG] + R(1) + text(2), [1] + text(1) + [IMG, IMG] + R(2)], [1, None, 2, 1], images_as="5d", labels=True)

    out = {"mask_token_id": np.int64(MASK), "depth_token_id": np.int64(DEPTH), "vocab": np.int64(V), "image_tokens": np.int64(NF),
           "case_names": np.array([c["name"] for c in cases])}
    side0 = getattr(model.llm.config, "tokenizer_padding_side", "right")
    mx0 = getattr(model.llm.config, "tokenizer_model_max_length", None)
    for ci, c in enumerate(cases):
        ids, am = c["ids"], c["am"]
        n_img = c["n_img"]
        images = img(n_img)
        depths = img(n_img) if c["depths"] else None
        masks = [None if k is None else boxes(k) for k in c["n_masks"]]
        labels = None
        if c["labels"]:
            labels = torch.randint(0, V, ids.shape, generator=g)
            labels[ids == IMG] = so.IGNORE_INDEX
        im_in, dp_in = images, depths
        if c["images_as"] == "list":
            im_in = [images[i:i + 1] for i in range(n_img)]
            dp_in = None if depths is None else [depths[i:i + 1] for i in range(n_img)]
        elif c["images_as"] == "5d":
            im_in = images.reshape(ids.shape[0], -1, 3, S, S)
            dp_in = None if depths is None else depths.reshape(ids.shape[0], -1, 3, S, S)
        model.llm.config.tokenizer_padding_side = c["side"]
        model.llm.config.tokenizer_model_max_length = c["mx"] if c["mx"] is not None else mx0
        pre = f"c{ci}."
        out[pre + "input_ids"] = ids.numpy()
        out[pre + "has_attention_mask"] = np.int64(am is not None)
        if am is not None:
            out[pre + "attention_mask"] = am.numpy()
        if labels is not None:
            out[pre + "labels"] = labels.numpy()
        out[pre + "n_masks"] = np.array([-1 if k is None else k for k in c["n_masks"]], dtype=np.int64)
        out[pre + "have_depths"] = np.int64(c["depths"])
        out[pre + "padding_side_left"] = np.int64(c["side"] == "left")
        out[pre + "max_length"] = np.int64(-1 if c["mx"] is None else c["mx"])
        # the oracle's restatement on the same coded rows
        feats_c = (coded(n_img * NF) + 1e5 * KIND_IMAGE).reshape(n_img, NF, H)
        off, me_c, de_c = 0, [], ([] if c["depths"] else None)
        for k in c["n_masks"]:
            n = 0 if k is None else k
            me_c.append(None if k is None else coded(off + n)[off:] + 1e5 * KIND_MASK)
            if de_c is not None:
                de_c.append(None if k is None else coded(off + n)[off:] + 1e5 * KIND_DEPTH)
            off += n
        wv = {so.LM + "model.embed_tokens.weight": coded(V)}
        cfg.padding_side = c["side"]
        cfg.tokenizer_model_max_length = c["mx"] if c["mx"] is not None else mx0
        err = None
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                (_, pos, am_out, _, embeds, new_labels) = model.prepare_inputs_labels_for_multimodal(
                    ids, None, am, None, labels, im_in, masks, dp_in)
            except (RuntimeError, IndexError) as e:
                err = type(e).__name__
            o_err = None
            try:
                o = so.splice(wv, cfg, ids, am, feats_c, me_c, de_c, have_depths=bool(c["depths"]), labels=labels)
            except (RuntimeError, IndexError) as e:
                o_err = type(e).__name__
        assert (err is None) == (o_err is None), (c["name"], err, o_err)
        out[pre + "raises"] = np.int64(err is not None)
        if err is not None:
            print(f"  {c['name']}: the reference raises {err} (oracle: {o_err})")
            continue
        code = embeds[..., 0].double()
        assert torch.equal(embeds, embeds[..., :1].expand_as(embeds)), "coded rows must stay constant along H"
        kind = torch.floor(code / 1e5).long()
        index = (code - 1e5 * kind).round().long() - 1  # -1: a padding row (zeros)
        kind[index < 0] = -1
        out[pre + "src_kind"] = kind.numpy().astype(np.int8)
        out[pre + "src_index"] = index.numpy().astype(np.int32)
        out[pre + "has_attention_mask_out"] = np.int64(am_out is not None)
        if am_out is not None:
            out[pre + "attention_mask_out"] = am_out.numpy().astype(np.int8)
        if labels is not None:
            out[pre + "new_labels"] = new_labels.numpy()
        assert torch.equal(o[0], embeds), c["name"]
        assert (o[1] is None) == (am_out is None) and (am_out is None or torch.equal(o[1].bool(), am_out.bool())), c["name"]
        if labels is not None:
            assert torch.equal(o[3], new_labels), c["name"]
        print(f"  {c['name']}: T = {embeds.shape[1]}, sources text/image/mask/depth/pad = "
              + "/".join(str(int((kind == k).sum())) for k in (0, 1, 2, 3, -1)) + "  oracle == reference")
    model.llm.config.tokenizer_padding_side, model.llm.config.tokenizer_model_max_length = side0, mx0
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(GOLD, "splice_kat.npz"), **out)
    print("  wrote splice_kat.npz")


def mint_posembed_kat():
    """vision_resolution elevation: run the reference's OWN `VisionTower._maybe_resize_pos_embeds`
    (llava/model/multimodal_encoder/vision_encoder.py:36-113, interpolate_mode "linear") on stand-in objects that carry exactly
    the attributes the method touches (the transformers-5 SiglipVisionModel has no `.vision_model`), and store
    old/new position tables for an up-sizing (27^2 -> 32^2 tokens) and a down-sizing (27^2 -> 24^2 = the 336-px case)."""
    from types import SimpleNamespace

    rh.install_shims()
    from llava.model.multimodal_encoder.vision_encoder import VisionTower

    blob = {}
    g = torch.Generator().manual_seed(11)
    for tag, old_res, new_res, patch in (("up", 378, 448, 14), ("down", 384, 336, 14)):
        n_old = (old_res // patch) ** 2
        emb = torch.nn.Embedding(n_old, 48)
        emb.weight.data = torch.randn((n_old, 48), generator=g)
        old = emb.weight.data.clone()
        embeddings = SimpleNamespace(patch_size=patch, position_embedding=emb, image_size=old_res, num_patches=n_old,
                                     num_positions=n_old, position_ids=None)
        model = SimpleNamespace(config=SimpleNamespace(image_size=old_res), vision_model=SimpleNamespace(embeddings=embeddings))
        proc = SimpleNamespace(size={"height": old_res, "width": old_res})
        tower = VisionTower.__new__(VisionTower)
        VisionTower._maybe_resize_pos_embeds(tower, model, proc, resolution=new_res, interpolate_mode="linear")
        new = embeddings.position_embedding.weight.data
        assert new.shape == ((new_res // patch) ** 2, 48) and model.config.image_size == new_res
        assert proc.size == {"height": new_res, "width": new_res}
        blob[f"{tag}_old"], blob[f"{tag}_new"] = old.numpy(), new.numpy()
        print(f"  posembed {tag}: {n_old} -> {new.shape[0]} rows")
    np.savez_compressed(os.path.join(GOLD, "posembed_kat.npz"), **blob)
    print("  wrote posembed_kat.npz")


def mint_checkpoint():
    """tests/golden/ckpt_tiny/: a checkpoint directory written by the REFERENCE's own writer -- `LlavaLlamaModel.save_pretrained`
    (llava_arch.py:181-250: tokenizer + HF sub-model directories llm/ vision_tower/ mm_projector/ region_extractor/ and the
    top-level config.json with the nested sub-configs) -- for the loader tests, plus the ids the reference's `generate()` produces
    on it (tests/golden/ckpt_tiny_kat.npz).  The model is the tiny fp32 model of tiny_fp32.npz (same seeds, same perturbed gains)."""
    import shutil

    out = os.path.join(GOLD, "ckpt_tiny")
    shutil.rmtree(out, ignore_errors=True)
    with tempfile.TemporaryDirectory() as td:
        model, tok = rh.build_tiny_reference_model(td, llm=TINY_LLM, vit=TINY_VIT, dtype="torch.float32", seed=0)
        g = torch.Generator().manual_seed(123)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith("norm.weight") or "layernorm" in n or "layer_norm" in n or n.endswith("module.1.weight") \
                        or n == "mm_projector.layers.1.weight":
                    if n.endswith("weight"):
                        p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                    else:
                        p.copy_(0.1 * torch.randn(p.shape, generator=g))
                elif n.endswith(".bias"):
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
        # a Llama-3 style generation config next to the weights: HF reads it back into `llm.generation_config`
        model.llm.generation_config.eos_token_id = [2, 9]
        model.llm.generation_config.pad_token_id = None
        model.save_pretrained(out)  # <-- the reference's writer
        cfg = cfg_from(model, tok)
        ids, images, depths, masks = so.synth_inputs(cfg, batch=1, regions=2, prompt_len=15, seed=1)
        with torch.no_grad():
            gen = model.generate(input_ids=ids, images=images, depths=depths, masks=masks, do_sample=False, max_new_tokens=12,
                                 use_cache=True, eos_token_id=None, pad_token_id=0, min_new_tokens=12)
        prompt = "<image>\nhow far is region <mask> <depth> from region <mask> <depth> ?"
        sys.path.insert(0, rh.REFERENCE_ROOT)
        from llava.mm_utils import tokenizer_image_token

        pids = tokenizer_image_token(prompt, tok, return_tensors="pt")
    files = sorted(os.path.relpath(os.path.join(r, f), out) for r, _, fs in os.walk(out) for f in fs)
    total = sum(os.path.getsize(os.path.join(out, f)) for f in files)
    print(f"  reference save_pretrained wrote {len(files)} files, {total / 1e6:.2f} MB:", files)
    np.savez_compressed(os.path.join(GOLD, "ckpt_tiny_kat.npz"), input_ids=ids.numpy(),
                        images_q32=(images * 32).round().to(torch.int8).numpy(),
                        depths_q32=(depths[:, :1] * 32).round().to(torch.int8).numpy(),
                        masks_u8=torch.stack(masks).to(torch.uint8).numpy(), new_ids=gen.numpy(),
                        prompt=np.frombuffer(prompt.encode(), dtype=np.uint8), prompt_ids=pids.numpy(),
                        mask_token_id=cfg.mask_token_id, depth_token_id=cfg.depth_token_id)
    print("  wrote ckpt_tiny/ and ckpt_tiny_kat.npz; reference ids:", gen.tolist())


def mint_vendored_llama_kat():
    """The reference's VENDORED Llama file (llava/train/transformers_replace/models/llama/modeling_llama.py: the decoder the
    reference actually runs -- LlamaFlashAttention2 hard-wired at :611-619, LlamaLinearScalingRotaryEmbedding :133-140) executed
    on CPU with oracle/flash_attn_cpu.py standing in for the flash-attn CUDA extension, under `rope_scaling = {linear, 3.0}` as
    context_length_extension writes it (language_model/builder.py:31-38; 3.0 is not a power of two, so "divide the positions" and
    "divide inv_freq" round differently and the fixture tells them apart).  Cases:
      single : one 40-token prompt, fp32 -- all-position logits, then 10 greedy steps over the returned cache (positions 40..49
               of a model whose max_position_embeddings is 32: beyond the original context, the reason the scaling exists)
      ragged : a right-padded batch (lengths 40 / 23) with the attention mask, position ids and seqlens_in_batch the LLaVA wrapper
               passes (llava_llama.py:150-176) -- the varlen / unpad branch of _flash_attention_forward
      bf16   : the single case in bfloat16
    Asserts that oracle.llama_forward reproduces all of them (fp32: 2e-6 of the logit range, ids equal; bf16: 3e-2) and writes
    tests/golden/vendored_llama_kat.npz."""
    m = rh.install_vendored_llama()
    from transformers import LlamaConfig

    geo = dict(hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
               vocab_size=128, max_position_embeddings=32, rms_norm_eps=1e-5)
    factor, theta = 3.0, 10000.0
    lc = LlamaConfig(**geo)
    for k, v in dict(rope_theta=theta, rope_scaling={"type": "linear", "factor": factor}, attention_bias=False,
                     attention_dropout=0.0, pretraining_tp=1, use_cache=True, output_attentions=False,
                     output_hidden_states=False).items():
        setattr(lc, k, v)  # transformers 5 moved / dropped these LlamaConfig fields; the vendored file reads the 4.37.2 names
    torch.manual_seed(11)
    lm = m.LlamaForCausalLM(lc).eval()
    assert type(lm.model.layers[0].self_attn).__name__ == "LlamaFlashAttention2"
    assert type(lm.model.layers[0].self_attn.rotary_emb).__name__ == "LlamaLinearScalingRotaryEmbedding"
    with torch.no_grad():  # spread the logits (default init gives a near-flat distribution: argmax would sit on rounding noise)
        lm.lm_head.weight.mul_(8.0)
        lm.model.embed_tokens.weight.mul_(20.0)
    sd = {"llm." + k: v.detach().clone() for k, v in lm.state_dict().items() if "rotary_emb" not in k}
    cfg = so.SrgptConfig(hidden=64, inter=160, layers=2, heads=4, kv_heads=2, vocab=128, rms_eps=1e-5, rope_theta=theta,
                         rope_factor=factor, mask_token_id=126, depth_token_id=127)
    g = torch.Generator().manual_seed(5)
    T, G = 40, 10
    ids = torch.randint(3, 120, (1, T), generator=g)
    out = {"geo_json": np.frombuffer(json.dumps(dict(geo, rope_theta=theta, rope_factor=factor)).encode(), dtype=np.uint8)}
    for k, v in sd.items():
        out["w." + k] = v.numpy()

    def run_single(model, dtype):
        with torch.no_grad():
            o = model(input_ids=ids, use_cache=True)
            logits, past = o.logits, o.past_key_values
            new, steps = [], []
            nxt = logits[:, -1].argmax(-1)
            for t in range(G):
                new.append(nxt)
                o = model(input_ids=nxt[:, None], past_key_values=past, use_cache=True,
                          position_ids=torch.tensor([[T + t]]))
                past = o.past_key_values
                steps.append(o.logits[:, -1])
                nxt = o.logits[:, -1].argmax(-1)
        return logits, torch.stack(new, 1), torch.stack(steps, 1)

    def oracle_single(w, dtype):
        kv = so.KVCache(cfg.layers)
        emb = torch.nn.functional.embedding(ids, w["llm.model.embed_tokens.weight"])
        logits = so.llama_forward(w, cfg, emb, torch.arange(T)[None], kv)
        new, steps = [], []
        nxt = logits[:, -1].argmax(-1)
        for t in range(G):
            new.append(nxt)
            e = torch.nn.functional.embedding(nxt[:, None], w["llm.model.embed_tokens.weight"])
            lg = so.llama_forward(w, cfg, e, torch.tensor([[T + t]]), kv, last_only=True)[:, -1]
            steps.append(lg)
            nxt = lg.argmax(-1)
        return logits, torch.stack(new, 1), torch.stack(steps, 1)

    ref_l, ref_ids, ref_s = run_single(lm, torch.float32)
    got_l, got_ids, got_s = oracle_single(sd, torch.float32)
    rng = float(ref_l.abs().max())
    top2 = ref_s.topk(2, -1).values
    print(f"  vendored llama fp32: logit range {rng:.2f}, min top-1/top-2 margin over the {G} steps {float((top2[..., 0] - top2[..., 1]).min()):.3f}")
    check("vendored.single.prefill_logits", got_l, ref_l, 2e-6 * rng)
    check("vendored.single.step_logits", got_s, ref_s, 2e-6 * rng)
    assert torch.equal(got_ids, ref_ids), (got_ids, ref_ids)
    out.update({"single.ids": ids.numpy(), "single.prefill_logits": ref_l.numpy(), "single.new_ids": ref_ids.numpy(),
                "single.step_logits": ref_s.numpy()})

    # the rotary tables of the vendored class itself (fp32 and bf16), positions 0 .. 63: pins weights.rope_tables bit for bit
    rot = lm.model.layers[0].self_attn.rotary_emb
    pp = torch.arange(64)[None]
    c32, s32 = rot(torch.zeros(1, dtype=torch.float32), pp)
    c16, s16 = rot(torch.zeros(1, dtype=torch.bfloat16), pp)
    oc, osn = so.rope_cos_sin(cfg, pp, torch.float32)
    assert torch.equal(oc, c32) and torch.equal(osn, s32)
    half = cfg.head_dim // 2
    out.update({"rope.cos_f32": c32[0, :, :half].numpy(), "rope.sin_f32": s32[0, :, :half].numpy(),
                "rope.cos_bf16": c16[0, :, :half].float().numpy(), "rope.sin_bf16": s16[0, :, :half].float().numpy()})

    # ragged right-padded batch through the unpad / varlen branch, called the way LlavaLlamaModel.forward calls the LLM
    lens = [40, 23]
    ids2 = torch.randint(3, 120, (2, T), generator=g)
    am = torch.zeros((2, T), dtype=torch.long)
    pos2 = torch.zeros((2, T), dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, :n] = 1
        pos2[b, :n] = torch.arange(n)
    with torch.no_grad():
        o2 = lm(input_ids=ids2, attention_mask=am, position_ids=pos2, seqlens_in_batch=am.sum(-1).int(), use_cache=False)
    emb2 = torch.nn.functional.embedding(ids2, sd["llm.model.embed_tokens.weight"])
    got2 = so.llama_forward(sd, cfg, emb2, pos2, so.KVCache(cfg.layers), key_padding_mask=am.bool())
    for b, n in enumerate(lens):
        check(f"vendored.ragged.row{b}", got2[b, :n], o2.logits[b, :n], 2e-6 * rng)
    out.update({"ragged.ids": ids2.numpy(), "ragged.lens": np.array(lens), "ragged.logits": o2.logits.numpy()})

    lm16 = m.LlamaForCausalLM(lc).eval()
    lm16.load_state_dict(lm.state_dict())
    lm16 = lm16.to(torch.bfloat16)
    sd16 = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    b_l, b_ids, b_s = run_single(lm16, torch.bfloat16)
    o_l, o_ids, o_s = oracle_single(sd16, torch.bfloat16)
    check("vendored.bf16.prefill_logits", o_l, b_l, 3e-2 * rng)
    check("vendored.bf16.step_logits", o_s, b_s, 3e-2 * rng)
    out.update({"bf16.prefill_logits": b_l.float().numpy(), "bf16.new_ids": b_ids.numpy(), "bf16.step_logits": b_s.float().numpy()})
    np.savez_compressed(os.path.join(GOLD, "vendored_llama_kat.npz"), **out)
    print("  wrote vendored_llama_kat.npz; vendored greedy ids:", ref_ids.tolist(), " bf16:", b_ids.tolist())


def mint_beam_kat():
    """`generate(num_beams=3)` of the REFERENCE model (tiny fp32, the weights of tiny_fp32.npz) -- the `--num_beams` flag of
    eval_spatial.py:234 / eval_region_cls.py:321 / model_vqa.py:75 -- for three settings: no EOS (every hypothesis runs to the
    budget), an EOS id that the search meets (hypotheses of different lengths compete under the length penalty), and a batch of two
    different prompts.  Asserts that spatialrgpt_amd.generation.beam_search, driven by the oracle's llama_forward, returns the same
    ids, and writes tests/golden/beam_kat.npz (inputs are tiny_fp32.npz's; only the settings and the reference ids are stored)."""
    print("== beam_kat.npz")
    from spatialrgpt_amd.generation import beam_search

    dtype = torch.float32
    with tempfile.TemporaryDirectory() as td:
        model, tok = rh.build_tiny_reference_model(td, llm=TINY_LLM, vit=TINY_VIT, dtype="torch.float32", seed=0)
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():  # the same post-init edits as mint_model_case / mint_labels_kat: identical weights
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or "layernorm" in n or "layer_norm" in n or n.endswith("module.1.weight") \
                    or n == "mm_projector.layers.1.weight":
                if n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    cfg = cfg_from(model, tok)
    w = canonical_state_dict(model)
    gold = np.load(os.path.join(GOLD, "tiny_fp32.npz"))
    for k, v in w.items():
        assert np.array_equal(gold["w." + k], tensor_np(v)), k
    ids, images, depths, masks = so.synth_inputs(cfg, batch=1, regions=2, prompt_len=15, seed=1, dtype=dtype)
    ids2, images2, depths2, masks2 = so.synth_inputs(cfg, batch=2, regions=2, prompt_len=15, seed=5, dtype=dtype)
    NB, G = 3, 12
    # a NON-ZERO pad id: the installed transformers pads finished rows with `pad_token_id or eos_token_id[0]`, so pad id 0 is
    # replaced by the first EOS id there; the reference's pinned 4.37.2 (BeamSearchScorer.finalize) pads with pad_token_id itself,
    # which is what spatialrgpt_amd.generation.beam_search does -- with a non-zero pad id the two releases agree
    PAD = 7

    def ref(ids_, im, dp, mk, **kw):
        with torch.no_grad():
            return model.generate(input_ids=ids_, images=im, depths=dp, masks=mk, do_sample=False, num_beams=NB, max_new_tokens=G,
                                  use_cache=True, **kw)

    def mine(ids_, im, dp, mk, eos, pad):
        emb, am, pid, _ = so.prepare_inputs(w, cfg, ids_, im, dp, mk)
        B, T, _ = emb.shape
        emb = emb.repeat_interleave(NB, dim=0)
        state = {"kv": so.KVCache(cfg.layers), "pos": T}
        first = so.llama_forward(w, cfg, emb, torch.arange(T)[None].expand(B * NB, -1), state["kv"], last_only=True)[:, -1]

        def step(tokens, beam_idx):
            kv = state["kv"]
            kv.k = [k.index_select(0, beam_idx) for k in kv.k]
            kv.v = [v.index_select(0, beam_idx) for v in kv.v]
            e = torch.nn.functional.embedding(tokens[:, None], w["llm.model.embed_tokens.weight"])
            lg = so.llama_forward(w, cfg, e, torch.full((B * NB, 1), state["pos"]), kv, last_only=True)[:, -1]
            state["pos"] += 1
            return lg

        return beam_search(first, step, B, NB, G, eos, pad)

    out = {}
    a = ref(ids, images, depths, masks, eos_token_id=None, pad_token_id=PAD)
    b = mine(ids, images, depths, masks, None, PAD)
    assert torch.equal(a, b), (a, b)
    greedy = gold["ref.new_ids"]
    print("  no EOS:", a.tolist(), " (greedy:", greedy.tolist(), ")")
    out["noeos.ids"] = a.numpy()
    # an EOS id the search meets: a token the no-EOS best hypothesis emits mid-way, plus one it never emits (a LIST, like Llama-3)
    eos = [int(a[0, 5]), 119]
    a2 = ref(ids, images, depths, masks, eos_token_id=eos, pad_token_id=PAD)
    b2 = mine(ids, images, depths, masks, eos, PAD)
    assert torch.equal(a2, b2), (a2, b2)
    print("  EOS", eos, ":", a2.tolist())
    out["eos.ids"], out["eos.eos"] = a2.numpy(), np.array(eos)
    a3 = ref(ids2, images2, depths2, masks2, eos_token_id=eos, pad_token_id=PAD)
    b3 = mine(ids2, images2, depths2, masks2, eos, PAD)
    assert torch.equal(a3, b3), (a3, b3)
    print("  batch of 2, EOS", eos, ":", a3.tolist())
    out["batch2.ids"], out["batch2.input_ids"] = a3.numpy(), ids2.numpy()
    out["batch2.images_q32"] = (images2 * 32).round().to(torch.int8).numpy()
    out["batch2.depths_q32"] = (depths2[:, :1] * 32).round().to(torch.int8).numpy()
    out["batch2.masks_u8"] = torch.stack(masks2).to(torch.uint8).numpy()
    out["num_beams"], out["max_new_tokens"], out["pad_token_id"] = np.int64(NB), np.int64(G), np.int64(PAD)
    np.savez_compressed(os.path.join(GOLD, "beam_kat.npz"), **out)
    print("  wrote beam_kat.npz")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "posembed":
        mint_posembed_kat()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ckpt":  # only the reference-written checkpoint directory
        mint_checkpoint()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "beam":  # only the beam-search fixture
        mint_beam_kat()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vendored":  # only the vendored-Llama fixture (rope scaling, flash-attn call sites)
        mint_vendored_llama_kat()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "splice":  # only the splice edge-case fixture (row-source maps from the reference)
        mint_splice_kat()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "labels":  # only the labels / loss fixture (reuses tiny_fp32.npz's weights)
        mint_labels_kat()
        sys.exit(0)
    mint_region_kat()
    mint_tokenizer_kat()
    mint_model_case(torch.float32, "tiny_fp32.npz")
    mint_model_case(torch.bfloat16, "tiny_bf16.npz")
    mint_model_case(torch.float32, "tiny_clip_fp32.npz", tower="clip")
    mint_labels_kat()
    mint_splice_kat()
    mint_posembed_kat()
    mint_checkpoint()
    mint_vendored_llama_kat()
    mint_beam_kat()
    print("golden vectors written to", GOLD)
