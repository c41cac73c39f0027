import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.model import LlavaLlamaModel
from spatialrgpt_amd.weights import synth_state_dict
dev="cuda"; cfg=SrgptConfig.vila15_8b()
sd=synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev)
model=LlavaLlamaModel(cfg, sd, device=dev, dtype=torch.bfloat16, rope_positions=1024, consume_state_dict=True)
req=bench.synth_request(cfg, 8, 64, 1, dev, torch.bfloat16)
def run(**kw):
    for _ in range(2):
        torch.cuda.synchronize(); t=time.perf_counter()
        out=model.generate(req[0], images=req[1], depths=req[2], masks=req[3], max_new_tokens=128, eos_token_id=None, **kw)
        torch.cuda.synchronize(); dt=time.perf_counter()-t
    return dt, out.shape
print("greedy", run(do_sample=False))
print("sample T=0.2", run(do_sample=True, temperature=0.2))
print("sample T=0.2 top_p=0.9", run(do_sample=True, temperature=0.2, top_p=0.9))
