#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --max-new-tokens 4 > /tmp/x.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 60 | cut -c1-170 | grep -i "region\|calls"
python $GRAFT_REPO_ROOT/scripts/prof_prefix.py $(find /tmp/prof_x -name "*.db" | head -1) | cut -c1-170 > $OUT/r03k_prefix.txt; head -4 $OUT/r03k_prefix.txt; grep -i region $OUT/r03k_prefix.txt
