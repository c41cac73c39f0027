"""Multi-GPU readiness on a ONE-GPU box: does a rank keep its request rate while 7 sibling processes run the same
Python launch loop on the same host?  The siblings cannot have GPUs here, so they spin what a rank's host side spins -- ctypes calls
into the C-ABI library in a tight loop (the per-launch host work without the device) -- unpinned, or pinned to the cpu shares
spatialrgpt_amd.dist.plan_rank_affinity gives local ranks 1..7 of 8 while the measured rank takes rank 0's share.
   python scripts/round6/sibling_load.py [steps]     -> one bench line per case (value, ms_per_step, prefix_ms_per_call)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIB = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
from spatialrgpt_amd.dist import plan_rank_affinity
pin = sys.argv[1]
if pin != "-":
    r, w = (int(v) for v in pin.split("/"))
    os.sched_setaffinity(0, plan_rank_affinity(r, w, sorted(os.sched_getaffinity(0))))
lib = ctypes.CDLL(os.path.join(%r, "spatialrgpt_amd", "libsrgpt_hip.so"))
f = lib.srgpt_abi_version
n = 0
while True:
    for _ in range(1000):
        f()
    n += 1
""" % (ROOT, ROOT)


def bench(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1", "--no-cpu-baseline"] + extra,
                         capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        return {"error": out.stderr[-400:]}
    d = json.loads(line[-1])
    return {k: d.get(k) for k in ("value", "ms_per_step", "prefix_ms_per_call")} | {"affinity": d["config"]["dist"].get("affinity_rank0")}


steps = sys.argv[1] if len(sys.argv) > 1 else "3"
cases = [("alone, unpinned", [], None), ("alone, pinned as local rank 0 of 8", ["--pin-as", "0/8"], None),
         ("7 busy siblings, nobody pinned", [], "-"), ("7 busy siblings, everyone on its own cpu share", ["--pin-as", "0/8"], "pin")]
print(f"host cpus visible: {len(os.sched_getaffinity(0))}")
for name, extra, sib in cases:
    procs = []
    if sib is not None:
        for r in range(1, 8):
            procs.append(subprocess.Popen([sys.executable, "-c", SIB, "-" if sib == "-" else f"{r}/8"]))
        time.sleep(1.0)
    try:
        res = bench(extra)
    finally:
        for p in procs:
            p.kill()
        for p in procs:
            p.wait()
    print(f"{name}: {json.dumps(res)}", flush=True)
