"""decode step time vs the decode-attention split count (tuning build: SRGPT_DECODE_MIN_SPLITS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libsrgpt_hip.so", "libsrgpt_hip_tuning.so")
from spatialrgpt_amd.config import SrgptConfig
from spatialrgpt_amd.engine import SrgptEngine
from spatialrgpt_amd.weights import synth_state_dict
cfg = SrgptConfig.vila15_8b()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
eng = SrgptEngine(cfg, sd, device="cuda", dtype=torch.bfloat16, rope_positions=1024, consume_state_dict=True)
G = 128
x = torch.randn((1, 259, cfg.hidden), device="cuda").to(torch.bfloat16)
for rep in range(3):
    st, _, _ = eng.prefill(x, max_new=G)  # fresh positions every repetition (the cache holds T + G positions)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); eng.greedy_decode(st, G); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / G
print(f"min_splits={os.environ.get('SRGPT_DECODE_MIN_SPLITS', '16')}: {ms:.4f} ms/token")
