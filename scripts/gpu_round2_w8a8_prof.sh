#!/bin/bash
# kernel table of the W8A8 configs[4] line (rocprofv3 --kernel-trace; no counters in this run)
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_c
timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --weights fp8_w8a8 --batch 8 > /tmp/c.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_c -name "*.db" | head -1) 16 > $OUT/r02_final_kernel_stats_fp8a8b8.txt
head -14 $OUT/r02_final_kernel_stats_fp8a8b8.txt | cut -c1-150
