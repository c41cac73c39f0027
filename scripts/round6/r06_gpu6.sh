#!/bin/bash
# round 6: kernel trace of configs[4] (fp8, 8 sequences per GPU) with the packed decode layout
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_c
rocprofv3 --kernel-trace -d /tmp/prof_c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset config4 > /tmp/c.log 2>&1
tail -2 /tmp/c.log
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_c -name "*.db" | head -1) 16 > $OUT/r06_mid_kernel_stats_fp8a8b8.txt
cut -c1-200 $OUT/r06_mid_kernel_stats_fp8a8b8.txt | head -24
