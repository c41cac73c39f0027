#!/bin/bash
# round 6, final tree: round profile (kernel trace, PMC passes, batched / fp8 kernel tables), one bench line per config, bench tests
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -4 ) > $OUT/r06_final_quicktests.log 2>&1
bash scripts/profile_round.sh r06_final > $OUT/r06_final_profile.log 2>&1
bash scripts/run_configs.sh r06 > $OUT/r06_configs.txt 2>&1
cat $OUT/r06_final_quicktests.log; cat $OUT/r06_configs.txt; head -3 $OUT/r06_final_prefix.txt; cat $OUT/r06_final_bench_default.json | cut -c1-600
