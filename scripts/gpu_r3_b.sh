#!/bin/bash
# round 3, GPU call b: fused attention + o_proj -- bit-exactness vs the per-op composition, phase stamps, knob A/B of the decode step
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_loader.py -x -q -k "fused or generation_config" 2>&1 ) | tail -15 > $OUT/r03b_tests.txt; cat $OUT/r03b_tests.txt
for v in "SRGPT_FUSE_SLEEP=0" "SRGPT_FUSE_SLEEP=5" "SRGPT_FUSE_SLEEP=5 SRGPT_DECODE_PF_GATEUP_ROUNDS=0"; do
  echo "## $v"; env $v timeout 300 python scripts/ubench_decode_stamps.py 1 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -24
done > $OUT/r03b_stamps.txt; cat $OUT/r03b_stamps.txt
scripts/ab_decode_step.sh r03b_step.txt "bf16:1" \
  "SRGPT_DECODE_FUSE_OPROJ=0" \
  "SRGPT_FUSE_SLEEP=0 SRGPT_DECODE_PF_GATEUP_ROUNDS=0" \
  "SRGPT_FUSE_SLEEP=5 SRGPT_DECODE_PF_GATEUP_ROUNDS=0" \
  "SRGPT_FUSE_SLEEP=10 SRGPT_DECODE_PF_GATEUP_ROUNDS=0" \
  "SRGPT_FUSE_SLEEP=5 SRGPT_DECODE_PF_GATEUP_ROUNDS=1 SRGPT_FUSE_PF_WHEN=1" \
  "SRGPT_FUSE_SLEEP=5 SRGPT_DECODE_PF_GATEUP_ROUNDS=1 SRGPT_FUSE_PF_WHEN=0" \
  "SRGPT_FUSE_SLEEP=5 SRGPT_DECODE_PF_GATEUP_ROUNDS=0 SRGPT_FUSE_POLL_SLEEP=16"
