"""Soak / race screen: many generate() calls of mixed batch sizes, lengths AND decoding modes on one engine; every repeat of a request
must return bit-identical ids (any race in the graph replay, the DMA-staged GEMM or the wave-private LDS stages shows up here).
Round 4: the cases rotate through greedy, device-side sampling (seeded), a stopping criterion (the run-ahead loop discards a step that
is already in flight) and beam search -- the pooled state switches between its greedy and its sampling graph, beams re-size it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spatialrgpt_amd import _lib
if os.environ.get("SRGPT_LIB"):  # a tuning build: the SRGPT_* knobs select kernel variants
    _lib.LIB_PATH = os.path.abspath(os.environ["SRGPT_LIB"])
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.model import LlavaLlamaModel
from spatialrgpt_amd.weights import synth_state_dict

dev = "cuda"
fmt = sys.argv[1] if len(sys.argv) > 1 else "native"
cfg = SrgptConfig.vila15_8b()
cfg = SrgptConfig(**{**cfg.to_dict(), "layers": 8, "vit_layers": 7})  # true widths, reduced depth: more calls per minute
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev)
model = LlavaLlamaModel(cfg, sd, device=dev, dtype=torch.bfloat16, rope_positions=2048, consume_state_dict=True, llm_weight_format=fmt)
g = torch.Generator().manual_seed(0)
cases = []
for i in range(8):
    B = [1, 2, 3, 4, 8, 16, 5, 1][i]
    P = [64, 40, 64, 100, 64, 32, 200, 512][i]
    G = [32, 16, 24, 8, 16, 8, 12, 40][i]
    reqs = [bench.synth_request(cfg, 4 + (i % 3), P, 10 * i + b, dev, torch.bfloat16) for b in range(B)]
    cases.append((torch.cat([r[0] for r in reqs], 0), torch.cat([r[1] for r in reqs], 0), torch.cat([r[2] for r in reqs], 0), [r[3][0] for r in reqs], G))
ref = {}
nbad = 0
t0 = time.time()
n = 0
order = torch.randint(0, len(cases), (int(sys.argv[2]) if len(sys.argv) > 2 else 60,), generator=g).tolist()
for k, ci in enumerate(order):
    ids, im, dp, mk, G = cases[ci]
    mode = ["greedy", "sample", "greedy", "criterion", "sample", "greedy", "beam", "greedy"][ci]
    if mode == "sample":
        torch.manual_seed(1000 + ci)  # the sampler's Philox seed is drawn from torch's CPU generator
        kw = dict(do_sample=True, temperature=0.9, top_k=40, top_p=0.95)
    elif mode == "criterion":
        kw = dict(do_sample=False, stopping_criteria=[lambda ids_, s_: ids_.shape[1] >= 5])
    elif mode == "beam":
        kw = dict(do_sample=False, num_beams=2)
    else:
        kw = dict(do_sample=False)
    out = model.generate(ids, images=im, depths=dp, masks=mk, max_new_tokens=G, eos_token_id=None, **kw).cpu()
    assert out.shape[1] == (5 if mode == "criterion" else G), (mode, out.shape)
    n += 1
    if ci in ref:
        if not torch.equal(out, ref[ci]):
            bad = (out != ref[ci]).nonzero()
            print(f"call {k}: case {ci} (batch {ids.shape[0]}) differs from its first run: first at (row, step) {bad[0].tolist()}, "
                  f"{int((out != ref[ci]).sum())} of {out.numel()} ids")
            nbad += 1
            if nbad >= 3:
                raise SystemExit("soak FAILED")
    else:
        ref[ci] = out
if nbad:
    raise SystemExit("soak FAILED")
print(f"soak ok: {n} calls, {len(ref)} distinct cases, weights={fmt}, {time.time() - t0:.1f} s")
