#!/bin/bash
# One bench line per BASELINE.json config that fits one GPU (the 8-GPU configs are run at their per-GPU share).
# Usage: bash scripts/run_configs.sh [tag]  -> gpurun_out/<tag>_configs.jsonl (+ a table on stdout -> profiles/<tag>_configs.txt)
TAG=${1:-r05}
OUT=$PWD/gpurun_out/${TAG}_configs.jsonl
: > $OUT
run() { echo "# $1" >> $OUT; shift; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $OUT; }
run "configs[1] VILA1.5-8B, 8 regions, bs=1 (headline)"
run "configs[2] VILA1.5-8B bs=32 over 8 GPUs -> 4 requests per GPU (one GPU's share)" --batch 4
run "configs[3] llama2_7b geometry, 16 regions, 512-id prompt" --model llama2_7b --regions 16 --prompt-len 512
run "configs[4] VILA1.5-8B fp8 weights on the fp8 matrix pipe (W8A8 prefill, W8A16 decode), bs=64 over 8 GPUs -> 8 requests per GPU (one GPU's share)" --weights fp8_w8a8 --batch 8
run "configs[4] variant: W8A16 throughout (bf16 matrix pipe), same shape" --weights fp8 --batch 8
run "extra: VILA1.5-8B fp8 LLM weights, bs=1" --weights fp8
run "extra: sheared_3b geometry, bs=1" --model sheared_3b
run "extra: CLIP-L/14-336 tower (the only true 336-px tower) in front of the 8B LLM, bs=1" --model vila15_8b_clip336
echo "# N=2 on ONE device (both ranks on GPU 0, gloo exchange): exercises the self-launched multi-rank leg" >> $OUT
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 2>/dev/null | tail -1 >> $OUT
python - $OUT <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip()); continue
    d=json.loads(l); print("   ", d["value"], d["unit"], "| ms/step", d["ms_per_step"], "| gate/up GB/s", d["roofline"]["achieved"] if d["roofline"] else None, "| frac", d["roofline"]["frac"] if d["roofline"] else None, "| dtype", d["dtype"][:40])
PY
