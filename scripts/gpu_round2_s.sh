#!/bin/bash
# round 2, GPU call s: final check of the shipped build -- whole GPU suite, smoke(), default bench line, batch lines
OUT=$PWD/gpurun_out
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 ) > $OUT/r02s_tests.log 2>&1
tail -4 $OUT/r02s_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/r02s_bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02s_bench_default.json")); r=d["roofline"]; c=d["cpu_baseline"]
print("default:", d["value"], d["unit"], "| ms/step", d["ms_per_step"], "| roofline", r["kernel"][:40], r["achieved"], r["frac"], "| step frac", r["decode_frac_whole_step"], "| cpu", c and c["value"], c and c["cores"])
PY
for args in "--batch 4" "--weights fp8 --batch 8"; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', '|', d['value'], 'tok/s | ms/step', d['ms_per_step'], '|', r['kernel'][:60], '| GB/s', r['achieved'], 'frac', r['frac'], '| decode ms/step', r['decode_ms_per_step'], 'rows', r['decode_rows'], '| whole-step frac', r['decode_frac_whole_step'])"; done | tee $OUT/r02s_bench_batch.txt
