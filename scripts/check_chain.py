"""The persistent GEMV chain (SRGPT_DECODE_CHAIN=1, tuning build) against the per-op composition, bit for bit, at the three LLM widths
(+ the error / barrier words afterwards).   SRGPT_DECODE_CHAIN=1 python scripts/check_chain.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
from spatialrgpt_amd import ops
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.engine import SrgptEngine
from spatialrgpt_amd.weights import synth_state_dict

DEV = "cuda"
ok_all = True
for geom in ("gqa_4096", "mha_4096", "mha_2560"):
    kw = dict(vit_hidden=64, vit_inter=176, vit_layers=2, vit_heads=4, image_size=42, patch_size=14, layers=3, vocab=4098,
              mask_token_id=4096, depth_token_id=4097)
    if geom == "gqa_4096":
        kw.update(hidden=4096, inter=14336, heads=32, kv_heads=8)
    elif geom == "mha_4096":
        kw.update(hidden=4096, inter=11008, heads=32, kv_heads=32, rope_theta=10000.0)
    else:
        kw.update(hidden=2560, inter=6912, heads=20, kv_heads=20, rope_theta=10000.0)
    cfg = SrgptConfig(**kw)
    dt = torch.bfloat16
    eng = SrgptEngine(cfg, synth_state_dict(cfg, seed=5, dtype=dt, device=DEV), device=DEV, dtype=dt, rope_positions=512)
    w = eng.w
    T0, G = 120, 10
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn((1, T0, cfg.hidden), device=DEV, generator=g) * 0.5).to(dt)
    st, _, _ = eng.prefill(x, max_new=G + 2)
    Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
    toks = torch.randint(3, 4096, (G,), device=DEV, generator=g)
    kc, vc = st.kcache.clone(), st.vcache.clone()
    worst = 0.0
    for t in range(G):
        tok = toks[t:t + 1].reshape(1, 1)
        got = eng.step(st, tok)
        h = ops.embed_rows(w.embed, tok.reshape(-1))
        pos = torch.tensor([T0 + t], device=DEV, dtype=torch.int32)
        for i in range(cfg.layers):
            qkv = ops.gemv(h, w.llm_t["wqkv"][i], norm_w=w.llm_t["attn_norm"][i], eps=cfg.rms_eps)
            a = ops.decode_attention(qkv, kc[i], vc[i], pos, w.rope_cos, w.rope_sin, Hq, Hkv, D)
            h = ops.gemv(a, w.llm_t["wo"][i], residual=h)
            act = ops.gemv(h, w.llm_t["wgu"][i], norm_w=w.llm_t["mlp_norm"][i], eps=cfg.rms_eps, swiglu=True)
            h = ops.gemv(act, w.llm_t["wdown"][i], residual=h)
        ref = ops.gemv(h, w.lm_head, norm_w=w.final_norm, eps=cfg.rms_eps, out_f32=True)
        worst = max(worst, float((got - ref).abs().max()))
    rc = _lib.load().srgpt_llm_decode_sync_state(C.byref(w.llm), C.byref(st.c), ops._stream())
    print(f"{geom}: chain={os.environ.get('SRGPT_DECODE_CHAIN', '0')} max |step - composition| over {G} steps = {worst:.3e}  (bit-identical: {worst == 0.0}); "
          f"sync state rc {rc} {_lib.last_error() if rc else ''}", flush=True)
    ok_all &= worst == 0.0 and rc == 0
    del eng
    torch.cuda.empty_cache()
print("CHAIN_CHECK", "OK" if ok_all else "FAILED")
