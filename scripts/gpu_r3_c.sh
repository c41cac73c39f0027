#!/bin/bash
# round 3, GPU call c: fused attention + o_proj with the LDS-shared attention vector: bit-exactness, stamps, A/B, kernel trace of both forms
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused" 2>&1 ) | tail -5 > $OUT/r03c_tests.txt; cat $OUT/r03c_tests.txt
for v in "SRGPT_FUSE_SLEEP=10 SRGPT_DECODE_PF_GATEUP_ROUNDS=0"; do
  echo "## $v"; env $v timeout 300 python scripts/ubench_decode_stamps.py 1 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -24
done > $OUT/r03c_stamps.txt; cat $OUT/r03c_stamps.txt
scripts/ab_decode_step.sh r03c_step.txt "bf16:1" \
  "SRGPT_DECODE_FUSE_OPROJ=0" \
  "SRGPT_FUSE_SLEEP=5 SRGPT_DECODE_PF_GATEUP_ROUNDS=0" \
  "SRGPT_FUSE_SLEEP=10 SRGPT_DECODE_PF_GATEUP_ROUNDS=0" \
  "SRGPT_FUSE_SLEEP=10 SRGPT_DECODE_PF_GATEUP_ROUNDS=1 SRGPT_FUSE_PF_WHEN=0"
cd /tmp && export TMPDIR=/tmp
for v in "SRGPT_DECODE_FUSE_OPROJ=0" "SRGPT_FUSE_SLEEP=10 SRGPT_DECODE_PF_GATEUP_ROUNDS=0"; do
  rm -rf /tmp/prof_x
  env $v SRGPT_LIB=$GRAFT_REPO_ROOT/spatialrgpt_amd/libsrgpt_hip_tuning.so rocprofv3 --kernel-trace -d /tmp/prof_x -o run -- python $GRAFT_REPO_ROOT/scripts/ubench_decode_step.py bf16:1 > /tmp/x.log 2>&1
  echo "## $v"; python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 12
done > $OUT/r03c_kernels.txt; cat $OUT/r03c_kernels.txt
