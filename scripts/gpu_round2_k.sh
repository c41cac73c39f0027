#!/bin/bash
# round 2, GPU call k: parity subset on the product build, then same-box A/B of the skinny decode kernel (HEAD's vs the new one)
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=spatialrgpt_amd
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemv or fp8" 2>&1 ) > $OUT/r02k_tests.log 2>&1
tail -4 $OUT/r02k_tests.log
{
for rep in 1 2; do
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so; do
  for cfg in "8 fp8 1" "8 fp8 2" "4 fp8 1" "4 bf16 1" "8 bf16 1" "16 bf16 1"; do
      set -- $cfg
      [ $lib = libsrgpt_hip_tuning_old.so ] && [ $3 = 2 ] && continue
      echo "== $lib batch $1 $2 mode$3 rep$rep"
      SRGPT_SKINNY_W8_MODE=$3 timeout 120 scripts/ubench_decode_mv $L/$lib $1 $2
  done
done
done
} > $OUT/r02k_mv.txt 2>&1
{
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so; do
SRGPT_LIB=$L/$lib SRGPT_SKINNY_W8_MODE=1 SRGPT_DECODE_PREFETCH_ROUNDS=0 timeout 300 python scripts/ubench_decode_step.py bf16:4 bf16:8
done
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so; do
SRGPT_LIB=$L/$lib SRGPT_SKINNY_W8_MODE=1 SRGPT_DECODE_PREFETCH_ROUNDS=0 timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/r02k_step.txt
cat $OUT/r02k_step.txt
