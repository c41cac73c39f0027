"""phase stamps (s_memtime, shader cycles) inside the decode attention kernel after a short decode run -- tuning build only.
   python scripts/experiments/ubench_decode_stamps.py [batch] [native|fp8|fp8_w8a8]
   (phase names: the MFMA kernel's -- decode_mfma_kernel, every bf16 / head_dim-128 geometry)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = os.path.abspath("spatialrgpt_amd/libsrgpt_hip_tuning.so")
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.engine import SrgptEngine
from spatialrgpt_amd.weights import synth_state_dict
cfg = SrgptConfig.vila15_8b()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fmt = sys.argv[2] if len(sys.argv) > 2 else "native"
eng = SrgptEngine(cfg, sd, device="cuda", dtype=torch.bfloat16, rope_positions=1024, consume_state_dict=True, llm_weight_format=fmt)
x = torch.randn((B, 259, cfg.hidden), device="cuda").to(torch.bfloat16)
st, _, _ = eng.prefill(x, max_new=128)
eng.greedy_decode(st, 100)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 32)()
assert lib.srgpt_debug_stamps(buf, 32) == 0
v = list(buf)
names = ["entry", "pos + K/V rows requested", "q/k/v + rope loads consumed", "barrier", "K/V wait + scores + softmax + P V", "per-wave results to LDS", "barrier",
         "waves merged + partials published", "stores drained + barrier", "ticket drawn"]
print(f"batch {B}, {fmt}: block 0 (kv head 0, split 0), cycles since entry")
for i in range(1, 10):
    print(f"  {names[i]:32s} +{v[i] - v[i-1]:7d}   (t = {v[i] - v[0]})")
print("merging block of (sequence 0, kv head 0):")
print(f"  partials + statistics loaded, weights  +{v[17] - v[16]:7d}")
print(f"  merge + store                          +{v[18] - v[17]:7d}")
