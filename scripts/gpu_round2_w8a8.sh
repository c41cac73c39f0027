#!/bin/bash
# W8A8 prefill (fp8 matrix pipe): parity tests, GEMM table, configs[4] line with and without it (same box)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 120 python scripts/probe_f8_mfma_accumulation.py > gpurun_out/r02_f8_accumulation_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r02_f8_accumulation_probe.txt | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_fp8_mfma.py -q -s > gpurun_out/r02_w8a8_tests.txt 2>&1; echo "tests rc=$?"; grep -n "gemm_w8a8 \|+ bias\|W8A8 vs\|passed\|failed\|Error\|FAILED" gpurun_out/r02_w8a8_tests.txt | cut -c1-220
timeout 200 python scripts/ubench_gemm_w8a8.py > gpurun_out/r02_w8a8_gemm.txt 2>&1; echo "ubench rc=$?"; cat gpurun_out/r02_w8a8_gemm.txt | cut -c1-220
for wt in fp8_w8a8 fp8; do
  timeout 300 python bench.py --weights $wt --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r02_w8a8_bench_$wt.err | tail -1 > gpurun_out/r02_w8a8_bench_$wt.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r02_w8a8_bench_$wt.json"))
    print("$wt", d["value"], d["unit"], "ms/step", d["ms_per_step"], {k: v for k, v in d.items() if "prefill" in k or "prefix" in k or "decode_ms" in k})
except Exception as e:
    print("$wt bench failed", e); print(open("gpurun_out/r02_w8a8_bench_$wt.err").read()[-1500:])
PY
done
