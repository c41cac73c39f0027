"""Shared helpers for the test-suite: golden-fixture loading and comparison."""
import json
import math
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _t(a, dtype):
    """npz array -> torch tensor (uint16 arrays hold raw bf16 bits)."""
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t


def load_tiny(name):
    """-> (cfg_dict, dtype, weights{name: tensor}, inputs dict, ref dict)"""
    z = np.load(os.path.join(GOLD, name))
    cfg = json.loads(bytes(z["cfg_json"]).decode())
    dtype = {"torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16}[bytes(z["dtype"]).decode()]
    w, ref = {}, {}
    for k in z.files:
        if k.startswith("w."):
            w[k[2:]] = _t(z[k], dtype)
        elif k.startswith("ref."):
            ref[k[4:]] = _t(z[k], dtype)
    images = (torch.from_numpy(z["in.images_q32"].astype(np.float32)) / 32).to(dtype)
    d1 = torch.from_numpy(z["in.depths_q32"].astype(np.float32)) / 32
    depths = d1.expand(-1, 3, -1, -1).contiguous().to(dtype)
    masks = [torch.from_numpy(z["in.masks_u8"][i].astype(np.float32)).to(dtype) for i in range(z["in.masks_u8"].shape[0])]
    # masks stacked as [n_img=1 stacked...]: stored shape [1, M, S, S]
    inputs = dict(input_ids=torch.from_numpy(z["in.input_ids"]), images=images, depths=depths, masks=masks)
    return cfg, dtype, w, inputs, ref


def load_kat():
    return np.load(os.path.join(GOLD, "region_kat.npz"))


def max_err(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def _log_measured(what, err, ref, atol):
    """SRGPT_PARITY_LOG=<file>: append the measured error of every comparison (so tolerances are set from data)."""
    path = os.environ.get("SRGPT_PARITY_LOG")
    if not path or not err.numel():
        return
    scale = float(ref.abs().max()) + 1e-30
    with open(path, "a") as f:
        f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "what": what, "max_err_over_ref_max": float(err.max()) / scale,
                            "atol_over_ref_max": atol / scale, "dtype_note": str(ref.dtype)}) + "\n")


def assert_close(a, b, atol, rtol=0.0, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    _log_measured(what, err, b, atol)
    bad = err > tol
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance, "
                                 f"max err {err.max().item():.4e} (ref max {b.abs().max().item():.4e})")


def teacher_forced_decode_logits(eng, st, teacher_ids):
    """Per-step logits of the DECODE path under teacher forcing: `st` is the state a prefill just returned (st.logits =
    logits of the last prompt position = step 0); teacher_ids [B, G] are the ids the checker (oracle / reference golden)
    generated.  Step t+1 feeds teacher_ids[:, t] through `eng.step` (GEMV kernels + decode attention over the appended cache),
    whatever the engine's own argmax was, so EVERY step is compared -- an in-margin flip never hides the rest.
    Returns fp32 [B, G, V]."""
    out = [st.logits.clone()]
    G = teacher_ids.shape[1]
    for t in range(G - 1):
        out.append(eng.step(st, teacher_ids[:, t:t + 1].to(eng.device)))
    return torch.stack(out, dim=1)


def logit_parity_report(got, ref, tol_rel, what=""):
    """got/ref fp32 [..., V].  Returns a dict: max|d| and rms as fractions of the reference logit range, argmax agreement, and
    the number of argmax disagreements whose reference top-1/top-2 margin exceeds 2*tol (those are failures)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    rng = float(ref.abs().max())
    d = got - ref
    flat_g, flat_r = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
    top2 = flat_r.topk(2, dim=-1).values
    margin = top2[:, 0] - top2[:, 1]
    agree = flat_g.argmax(-1) == flat_r.argmax(-1)
    bad = (~agree) & (margin > 2 * tol_rel * rng)
    return {"what": what, "rows": int(flat_r.shape[0]), "logit_range": rng, "max_abs_over_range": float(d.abs().max()) / rng,
            "rms_over_range": float(d.pow(2).mean().sqrt()) / rng, "argmax_agree": int(agree.sum()),
            "argmax_disagree_in_margin": int((~agree).sum() - bad.sum()), "argmax_disagree_out_of_margin": int(bad.sum()),
            "tol_rel": tol_rel}


def make_peaked(sd, cfg, seed=7, embed_std=0.5, resid_gain=1.0):
    """PEAKED-MARGIN variant of a synthetic state dict, in place (test infrastructure; SURVEY 7 "hard parts", VERDICT r2 #1b).

    Random N(0, 0.02) weights give logits whose top-1 / top-2 margin is inside the bf16 noise floor at most steps, so free-running
    greedy ids of two correct bf16 implementations diverge.  This keeps every layer random but gives the model a decision that
    survives bf16 arithmetic by a wide, ASSERTABLE margin:
      * residual-branch output projections (o_proj, down_proj) are scaled by resid_gain / sqrt(2 * layers) -- the GPT-2 /
        Megatron "scaled init": the 2 * layers random branch outputs then add up to an O(1) perturbation instead of swamping
        the token embedding;
      * embed_tokens ~ N(0, embed_std);
      * lm_head row perm[v] = the unit vector of embed_tokens row v (perm = a seeded permutation of the text ids), other rows 0:
        the logit of perm[last token] is the embedding's share of the final hidden state (~16 at the defaults), every other logit
        a random projection (~N(0, 1)) that DOES depend on all layers, the cache and the positions.
    So the greedy continuation walks the permutation, and the margin (asserted >= 10 x the measured bf16 noise floor by the tests)
    is what the layers' arithmetic has to preserve.  cfg: any object with hidden / layers / vocab / mask_token_id / depth_token_id."""
    dev = sd["llm.model.embed_tokens.weight"].device
    dt = sd["llm.model.embed_tokens.weight"].dtype
    g = torch.Generator(device="cpu").manual_seed(seed)
    V = sd["llm.model.embed_tokens.weight"].shape[0]
    hi = min(cfg.mask_token_id, cfg.depth_token_id, V)
    scale = resid_gain / math.sqrt(2.0 * cfg.layers)
    for i in range(cfg.layers):
        for n in ("self_attn.o_proj.weight", "mlp.down_proj.weight"):
            k = f"llm.model.layers.{i}.{n}"
            sd[k] = (sd[k].float() * scale).to(dt)
    emb = sd["llm.model.embed_tokens.weight"].float()
    emb = (emb / emb.std() * embed_std).to(dt)  # keeps the seeded pattern, sets the scale
    sd["llm.model.embed_tokens.weight"] = emb
    perm = torch.arange(V)
    text = torch.arange(3, hi)
    perm[3:hi] = text[torch.randperm(hi - 3, generator=g)]
    unit = torch.nn.functional.normalize(emb[3:hi].float(), dim=1)
    head = torch.zeros_like(emb, dtype=torch.float32)
    head[perm[3:hi].to(dev)] = unit
    sd["llm.lm_head.weight"] = head.to(dt)
    return perm


def make_heavy_tailed(w, cfg, seed=13, gains=(0.1, 50.0), n_massive=6, massive=300.0, hot_rows=3, hot_row_gain=30.0, attn_gain=4.0):
    """OUTLIER-STATISTICS variant of a synthetic weight dict (oracle key names), in place -- VERDICT r4 missing #3.  Trained SigLIP /
    Llama checkpoints (llava/model/builder.py:141-159; a8cheng/SpatialRGPT-VILA1.5-8B) are not N(0, 0.02): norm gains spread over
    orders of magnitude, a handful of "massive activation" channels of the residual stream carry values hundreds of times the rest,
    some weight rows are far larger than their neighbours, attention logits reach tens.  None can be downloaded here; this plants the
    same statistics into the seeded random weights:
      * every LayerNorm / RMSNorm / LayerNorm2d gain: log-uniform in [gains] per channel;
      * n_massive channels of embed_tokens and of the ViT position embedding multiplied by `massive`;
      * hot_rows random rows of every projection matrix (LLM q/k/v/o/gate/up/down, ViT q/k/v/out/fc1/fc2) multiplied by hot_row_gain;
      * the q and k rows of the first head (LLM and ViT) multiplied by attn_gain (logits x attn_gain^2).
    Returns a dict describing what was planted (for the report)."""
    g = torch.Generator().manual_seed(seed)
    lo, hi = math.log(gains[0]), math.log(gains[1])
    planted = {"gains": [], "massive_llm": None, "massive_vit": None, "hot_rows": 0}
    norm_tails = ("layernorm.weight", "layer_norm1.weight", "layer_norm2.weight", "model.norm.weight", "post_layernorm.weight",
                  "feature_refinement_module.1.weight", "mm_projector.layers.1.weight")
    proj_tails = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "o_proj.weight", "out_proj.weight", "gate_proj.weight",
                  "up_proj.weight", "down_proj.weight", "fc1.weight", "fc2.weight")
    for k in list(w):
        t = w[k]
        if k.endswith(norm_tails):
            w[k] = torch.exp(torch.rand(t.shape, generator=g) * (hi - lo) + lo).to(t.dtype)
            planted["gains"].append(k)
        elif k.endswith(proj_tails):
            rows = torch.randperm(t.shape[0], generator=g)[:hot_rows]
            tt = t.clone()
            tt[rows] = (tt[rows].float() * hot_row_gain).to(t.dtype)
            w[k] = tt
            planted["hot_rows"] += hot_rows
    e = w["llm.model.embed_tokens.weight"]
    dims = torch.randperm(e.shape[1], generator=g)[:n_massive]
    e = e.clone()
    e[:, dims] = (e[:, dims].float() * massive).to(e.dtype)
    w["llm.model.embed_tokens.weight"] = e
    planted["massive_llm"] = dims.tolist()
    pk = "vision_tower.vision_tower.vision_model.embeddings.position_embedding.weight"
    pe = w[pk].clone()
    vd = torch.randperm(pe.shape[1], generator=g)[:max(4, n_massive - 2)]
    pe[:, vd] = (pe[:, vd].float() * massive).to(pe.dtype)
    w[pk] = pe
    planted["massive_vit"] = vd.tolist()
    hd = cfg.hidden // cfg.heads
    for i in range(cfg.layers):
        for n in ("q_proj", "k_proj"):
            k = f"llm.model.layers.{i}.self_attn.{n}.weight"
            t = w[k].clone()
            t[:hd] = (t[:hd].float() * attn_gain).to(t.dtype)
            w[k] = t
    vhd = cfg.vit_hidden // cfg.vit_heads
    for k in list(w):
        if "vision_model.encoder.layers" in k and (k.endswith("self_attn.q_proj.weight") or k.endswith("self_attn.k_proj.weight")):
            t = w[k].clone()
            t[:vhd] = (t[:vhd].float() * attn_gain).to(t.dtype)
            w[k] = t
    return planted


# ---- tests/golden/splice_kat.npz: the reference's own row-source maps of the token-stream splice's edge cases ----
KIND_PAD, KIND_TEXT, KIND_IMAGE, KIND_MASK, KIND_DEPTH = -1, 0, 1, 2, 3


def splice_kat_cases():
    """Yield the cases of tests/golden/splice_kat.npz (oracle/make_golden.py `splice`: llava_arch.py:333-650 run on index-coded
    rows) as dicts: inputs (ids, attention mask | None, labels | None, per-image region counts with None entries, have_depths,
    padding side, max length) and the reference's outputs (src_kind / src_index [B, T]: which row of which table every output row
    is -- KIND_* above, index -1 = padding zeros --, attention mask, labels) or raises=True."""
    z = np.load(os.path.join(GOLD, "splice_kat.npz"))
    head = dict(mask_token_id=int(z["mask_token_id"]), depth_token_id=int(z["depth_token_id"]), vocab=int(z["vocab"]),
                image_tokens=int(z["image_tokens"]))
    for ci, name in enumerate(z["case_names"].tolist()):
        g = lambda k: z[f"c{ci}.{k}"]  # noqa: E731
        has = lambda k: f"c{ci}.{k}" in z.files  # noqa: E731
        ids = torch.from_numpy(g("input_ids"))
        c = dict(head, name=name, input_ids=ids,
                 attention_mask=torch.from_numpy(g("attention_mask")) if int(g("has_attention_mask")) else None,
                 labels=torch.from_numpy(g("labels")) if has("labels") else None,
                 n_masks=[None if k < 0 else int(k) for k in g("n_masks").tolist()], have_depths=bool(int(g("have_depths"))),
                 padding_side="left" if int(g("padding_side_left")) else "right",
                 max_length=None if int(g("max_length")) < 0 else int(g("max_length")), raises=bool(int(g("raises"))))
        c["n_images"] = len(c["n_masks"])
        if not c["raises"]:
            c["src_kind"] = torch.from_numpy(g("src_kind").astype(np.int64))
            c["src_index"] = torch.from_numpy(g("src_index").astype(np.int64))
            c["attention_mask_out"] = torch.from_numpy(g("attention_mask_out")).bool() if int(g("has_attention_mask_out")) else None
            c["new_labels"] = torch.from_numpy(g("new_labels")) if has("new_labels") else None
        yield c


def splice_kat_tables(c, hidden, embed=None, seed=0, dtype=torch.float32):
    """Source tables for a splice_kat case: (embed [vocab, H], image_features [n_images, nf, H], mask_embeds list, depth_embeds list |
    None) with random rows, and `expect(kind, index) -> [B, T, H]` assembling the rows the reference's map names."""
    g = torch.Generator().manual_seed(seed)
    nf = c["image_tokens"]
    if embed is None:
        embed = torch.randn((c["vocab"], hidden), generator=g).to(dtype)
    feats = torch.randn((c["n_images"], nf, hidden), generator=g).to(dtype)
    me = [None if k is None else torch.randn((k, hidden), generator=g).to(dtype) for k in c["n_masks"]]
    de = [None if k is None else torch.randn((k, hidden), generator=g).to(dtype) for k in c["n_masks"]] if c["have_depths"] else None

    def expect():
        kind, index = c["src_kind"], c["src_index"]
        out = torch.zeros(tuple(kind.shape) + (hidden,), dtype=dtype)
        tabs = {KIND_TEXT: embed.cpu(), KIND_IMAGE: feats.reshape(-1, hidden)}
        if any(e is not None for e in me):
            tabs[KIND_MASK] = torch.cat([e for e in me if e is not None], 0)
        if de is not None and any(e is not None for e in de):
            tabs[KIND_DEPTH] = torch.cat([e for e in de if e is not None], 0)
        for k, tab in tabs.items():
            sel = kind == k
            out[sel] = tab[index[sel]]
        return out
    return embed, feats, me, de, expect
