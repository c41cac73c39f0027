#!/bin/bash
# round 6: leaner RMSNorm staging in the batched product (1 / rms in registers, element pairs on v_pk_mul_f32) -- prev = the library before
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2 3; do
  run prev:packed $L/libsrgpt_hip_tuning_prev.so "$P" 8 fp8 pub
  run new:packed  $L/libsrgpt_hip_tuning.so "$P" 8 fp8 pub
done
for rep in 1 2; do
  run prev:rowmajor $L/libsrgpt_hip_tuning_prev.so "X=1" 4 bf16 pub
  run new:rowmajor  $L/libsrgpt_hip_tuning.so "X=1" 4 bf16 pub
  run prev:rowmajor $L/libsrgpt_hip_tuning_prev.so "X=1" 8 bf16 pub
  run new:rowmajor  $L/libsrgpt_hip_tuning.so "X=1" 8 bf16 pub
done
} > $OUT/r06_skinny_norm_staging.txt 2>&1
python3 scripts/round6/parse_mv.py $OUT/r06_skinny_norm_staging.txt
bash scripts/ab_libs_decode_step.sh r06_norm_staging_step.txt "fp8:8 bf16:4 bf16:8" $L/libsrgpt_hip_tuning_prev.so $L/libsrgpt_hip_tuning.so 2>&1 | tail -14
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv or rowss or batched_decode or skinny or pack" 2>&1 | tail -3 )
