#!/bin/bash
# round 2, GPU call t: bench.py contract tests + the bench lines of the shipped build
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_bench.py -q -x 2>&1 ) | tail -5
timeout 600 python bench.py 2>$OUT/r02t_bench_default.err | tail -1 > $OUT/r02t_bench_default.json
tail -3 $OUT/r02t_bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02t_bench_default.json")); r=d["roofline"]; c=d["cpu_baseline"]
print("default:", d["value"], d["unit"], "| ms/step", d["ms_per_step"], "| roofline", r["kernel"][:40], r["achieved"], r["frac"], "| step frac", r["decode_frac_whole_step"], "| cpu", c and c["value"], c and c["cores"])
PY
for args in "--batch 4" "--weights fp8 --batch 8"; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', '|', d['value'], 'tok/s | ms/step', d['ms_per_step'], '|', r['kernel'][:60], '| GB/s', r['achieved'], 'frac', r['frac'], '| decode ms/step', r['decode_ms_per_step'], 'rows', r['decode_rows'], '| whole-step frac', r['decode_frac_whole_step'])"; done | tee $OUT/r02t_bench_batch.txt
