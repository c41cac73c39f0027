#!/bin/bash
# round 6: per-layer teacher-forced outlier parity (writes gpurun_out/r06_outlier_per_layer.json) + the splice fixture tests
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT; rm -f $OUT/r06_outlier_per_layer.json
( timeout 2400 python -m pytest tests/test_gpu_outlier_parity.py -q -k "alone_on_the_oracles_input" 2>&1 | tail -40 ) > $OUT/r06_t9_perlayer.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_splice.py -x -q 2>&1 | tail -5 ) > $OUT/r06_t9_splice.log 2>&1
tail -30 $OUT/r06_t9_perlayer.log | cut -c1-600; cat $OUT/r06_t9_splice.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_outlier_per_layer.json"))
for k, v in d.items():
    print(k, " ".join(f"[{n}: rms {m["rms_ratio_to_floor"]:.2f} row {m["worst_row_rms_ratio_to_floor"]:.2f} max {m["max_ratio_to_floor"]:.2f}]" for n, m in v.items()))
PY
