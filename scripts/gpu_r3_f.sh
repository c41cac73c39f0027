#!/bin/bash
# round 3, GPU call f: free-running parity (peaked weights, embed std 0.5), region pooling + decode attention kernel tests, prefetch-batch A/B
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "region or rope_append or composition" 2>&1 ) | grep -v amdgpu.ids | tail -6 > $OUT/r03f_kernel_tests.txt; cat $OUT/r03f_kernel_tests.txt
( timeout 2400 python -m pytest tests/test_gpu_freerun_parity.py -q -s --durations=6 2>&1 ) | grep -v "amdgpu.ids" | grep -E "FREERUN|passed|failed|Error|assert|^[0-9.]+s " | cut -c1-1500 > $OUT/r03f_freerun.txt; cat $OUT/r03f_freerun.txt
scripts/ab_decode_step.sh r03f_step.txt "bf16:1" \
  "SRGPT_DECODE_PREFETCH_BATCH=1" "SRGPT_DECODE_PREFETCH_BATCH=2" "SRGPT_DECODE_PREFETCH_BATCH=4" "SRGPT_DECODE_PREFETCH_BATCH=8" \
  "SRGPT_DECODE_PREFETCH_BATCH=2 SRGPT_DECODE_PREFETCH_ROUNDS=1" "SRGPT_DECODE_PREFETCH_ROUNDS=0"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --max-new-tokens 8 > /tmp/x.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 60 | cut -c1-170 > $OUT/r03f_kernels.txt
grep -i "region\|decode_\|calls" $OUT/r03f_kernels.txt
