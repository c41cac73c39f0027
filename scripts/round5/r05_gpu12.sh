#!/bin/bash
# round 5, GPU call 12: ring depth of the fp8 SHORT launches (8-wave blocks: q/k/v -- 2 slices x 2 sub-units per wave): depth 3 / 4 puts
# all of a wave's weight stages in flight at kernel entry.  Variants of skinny.hip (tuning build), per product and per step.
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
{
for rep in 1 2; do for v in tuning s3 s4; do f=$L/libsrgpt_hip_tuning_$v.so; [ $v = tuning ] && f=$L/libsrgpt_hip_tuning.so
  for b in 8 4; do echo "== $v batch $b fp8 rep $rep"; scripts/ubench_decode_mv $f $b fp8 2>&1 | grep -v amdgpu.ids | tail -6; done; done; done
} > $OUT/r05_short_depth.txt 2>&1
bash scripts/ab_libs_decode_step.sh r05_short_depth_step.txt "fp8:8 fp8:4 fp8:2" $L/libsrgpt_hip_tuning.so $L/libsrgpt_hip_tuning_s3.so $L/libsrgpt_hip_tuning_s4.so > /dev/null 2>&1
( SRGPT_LIB=$L/libsrgpt_hip_tuning_s4.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv_w8 or rowss or batched_decode" 2>&1 | tail -3 ) > $OUT/r05_t12.log 2>&1
cat $OUT/r05_short_depth.txt | grep -E "==|qkv|sum"; cat $OUT/r05_short_depth_step.txt
