"""decode ms/token of the whole graph-captured step for several (weights, batch) configurations in one process.
  SRGPT_LIB=<path of a libsrgpt_hip*.so build> python scripts/ubench_decode_step.py fp8:8 fp8:4 bf16:4 bf16:1:2048
(weights:batch[:T] -- T = cached prompt positions, default 259)
Knobs of the tuning builds (SRGPT_SKINNY_W8_MODE, SRGPT_DECODE_PREFETCH_ROUNDS, ...) are read from the environment once per
process by the library itself; this script only reports them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatialrgpt_amd import _lib
if os.environ.get("SRGPT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SRGPT_LIB"])
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.engine import SrgptEngine
from spatialrgpt_amd.weights import synth_state_dict

cfg = SrgptConfig.vila15_8b()
G, T = 128, 259
knobs = {k: v for k, v in os.environ.items() if k.startswith("SRGPT_") or k.startswith("UBENCH_")}
wanted = [(a.split(":") + [str(T)])[:3] for a in sys.argv[1:]] or [["bf16", "1", str(T)]]
rope = max(1024, max(int(t) for _, _, t in wanted) + G + 128)
for fmt in sorted({w for w, _, _ in wanted}):
    sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
    eng = SrgptEngine(cfg, sd, device="cuda", dtype=torch.bfloat16, rope_positions=rope, consume_state_dict=True,
                      llm_weight_format="fp8" if fmt == "fp8" else "native", decode_layout=os.environ.get("UBENCH_DECODE_LAYOUT", "packed"))
    del sd
    for w, b, t_ in wanted:
        if w != fmt:
            continue
        B, T = int(b), int(t_)
        x = torch.randn((B, T, cfg.hidden), device="cuda").to(torch.bfloat16)
        best = 1e9
        for rep in range(3):
            st, _, _ = eng.prefill(x, max_new=G)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); eng.greedy_decode(st, G); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / G)
        wb = eng.w.llm_weight_bytes()
        print(f"{os.path.basename(_lib.LIB_PATH)} {knobs} | {fmt} batch {B}{'' if T == 259 else f' T {T}'}: {best:.4f} ms/step = {B / best * 1e3:.0f} tok/s decode-only, "
              f"{wb / best / 1e9:.2f} TB/s of weights = {wb / best / 1e9 / 8:.3f} of 8 TB/s"
              + ("" if T == 259 else f"; + KV {B * 2 * cfg.layers * cfg.kv_heads * cfg.head_dim * 2 * (T + G // 2) / 1e6:.0f} MB/step"), flush=True)
    del eng
    torch.cuda.empty_cache()
