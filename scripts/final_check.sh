#!/bin/bash
# Last GPU call of a round: whole suite + smoke + default bench (gpu_suite.sh), then the kernel trace of the same build
# (kernel table + request prefix) -- the PMC passes of profile_round.sh are not repeated.   gpurun -- 'bash scripts/final_check.sh r04'
R=${1:-r05}; OUT=$PWD/gpurun_out; mkdir -p $OUT
bash scripts/gpu_suite.sh ${R}_suite
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_kt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- $CMD > /tmp/kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $DB 40 > $OUT/${R}_final_kernel_stats.txt
python $GRAFT_REPO_ROOT/scripts/prof_prefix.py $DB > $OUT/${R}_final_prefix.txt
cp $OUT/${R}_suite_bench.json $OUT/${R}_final_bench_default.json
head -4 $OUT/${R}_final_prefix.txt
