"""Wall-clock breakdown of one generate() call (host + GPU, synchronised between phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.model import LlavaLlamaModel
from spatialrgpt_amd.weights import synth_state_dict

dev = "cuda"
cfg = SrgptConfig.vila15_8b()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev)
model = LlavaLlamaModel(cfg, sd, device=dev, dtype=torch.bfloat16, rope_positions=1024, consume_state_dict=True)
req = bench.synth_request(cfg, 8, 64, 1, dev, torch.bfloat16)
G = 128
for _ in range(2):
    model.generate(req[0], images=req[1], depths=req[2], masks=req[3], do_sample=False, max_new_tokens=G, eos_token_id=None)
torch.cuda.synchronize()
eng = model.engine
def t():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = t()
    feats = eng.encode_visual(req[1], req[2], req[3])
    t1 = t()
    emb, am, lens = eng.splice(req[0], None, *feats, have_depths=True)
    t2 = t()
    st, _, _ = eng.prefill(emb, max_new=G)
    t3 = t()
    out = eng.greedy_decode(st, G)
    t4 = t()
    ta = t()
    model.generate(req[0], images=req[1], depths=req[2], masks=req[3], do_sample=False, max_new_tokens=G, eos_token_id=None)
    tb = t()
    print(f"encode_visual {1e3*(t1-t0):.2f}  splice {1e3*(t2-t1):.2f}  prefill {1e3*(t3-t2):.2f}  decode {1e3*(t4-t3):.2f}  sum {1e3*(t4-t0):.2f} | generate() {1e3*(tb-ta):.2f} ms")
