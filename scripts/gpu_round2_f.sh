#!/bin/bash
OUT=$PWD/gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 ) > $OUT/r02i_tests.log 2>&1
tail -6 $OUT/r02i_tests.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02i_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02i_bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["decode_ms_per_token"], d["roofline"]["frac"], d["roofline"]["decode_frac_whole_step"])
PY
for b in 4 8; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | tail -1 | cut -c1-200; done | tee $OUT/r02i_bench_batch.txt
timeout 200 python scripts/soak.py 2>&1 | tail -3
