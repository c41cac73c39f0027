#!/bin/bash
# round 3, GPU call e: MFMA decode attention + two-launch region pooling: suite (minus the long free-running parity), A/B, stamps, trace
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_freerun_parity.py 2>&1 ) | grep -v amdgpu.ids | tail -25 > $OUT/r03e_tests.txt; cat $OUT/r03e_tests.txt
for v in "SRGPT_DECODE_MFMA=1" "SRGPT_DECODE_MFMA=0"; do
  echo "## $v"; env $v timeout 300 python scripts/ubench_decode_stamps.py 1 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -14
done > $OUT/r03e_stamps.txt; cat $OUT/r03e_stamps.txt
scripts/ab_decode_step.sh r03e_step.txt "bf16:1 bf16:4" \
  "SRGPT_DECODE_MFMA=0" \
  "SRGPT_DECODE_MFMA=1" \
  "SRGPT_DECODE_MFMA=1 SRGPT_DECODE_MIN_SPLITS=16" \
  "SRGPT_DECODE_MFMA=1 SRGPT_DECODE_MIN_SPLITS=4"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --max-new-tokens 8 > /tmp/x.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 60 | cut -c1-170 > $OUT/r03e_kernels.txt
grep -i "region\|decode_\|calls" $OUT/r03e_kernels.txt
