"""phase stamps (s_memtime) of block 0 of the persistent GEMV chain after a short decode run -- tuning build, SRGPT_DECODE_CHAIN=1.
   SRGPT_DECODE_CHAIN=1 python scripts/ubench_chain_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.engine import SrgptEngine
from spatialrgpt_amd.weights import synth_state_dict
cfg = SrgptConfig.vila15_8b()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
eng = SrgptEngine(cfg, sd, device="cuda", dtype=torch.bfloat16, rope_positions=1024, consume_state_dict=True)
x = torch.randn((1, 259, cfg.hidden), device="cuda").to(torch.bfloat16)
st, _, _ = eng.prefill(x, max_new=64)
eng.greedy_decode(st, 40)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 32)()
assert lib.srgpt_chain_debug_stamps(buf, 32) == 0
v = list(buf)
names = ["o_proj", "gate/up", "down", "qkv"]
print("block 0 of the last chain launch (last layer has 3 phases; stamps of phase 3 are from the layer before), cycles since entry")
for p in range(4):
    s = [v[1 + 4 * p + i] - v[0] for i in range(4)]
    print(f"  {names[p]:8s} vector staged t={s[0]:8d}  rows done t={s[1]:8d} (+{s[1] - s[0]})  at barrier t={s[2]:8d}  released t={s[3]:8d} (+{s[3] - s[2]})   loader issued its last granule t={v[20 + p] - v[0]:8d}")
