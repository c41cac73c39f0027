#!/bin/bash
# round 2, GPU call i: same-box A/B of the skinny decode kernel -- HEAD's (libsrgpt_hip_tuning_old.so) vs the counted-wait /
# batched-fragment version (libsrgpt_hip_tuning.so), per product and per decode step
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=spatialrgpt_amd
{
for rep in 1 2; do
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so; do
  for cfg in "8 fp8" "4 bf16" "8 bf16" "16 bf16"; do
      set -- $cfg
      echo "== $lib batch $1 $2 rep$rep"
      SRGPT_SKINNY_W8_MODE=1 timeout 120 scripts/ubench_decode_mv $L/$lib $1 $2
  done
done
done
} > $OUT/r02i_mv.txt 2>&1
{
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so; do
SRGPT_LIB=$L/$lib SRGPT_SKINNY_W8_MODE=1 SRGPT_DECODE_PREFETCH_ROUNDS=0 timeout 300 python scripts/ubench_decode_step.py bf16:4 bf16:8
done
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so; do
SRGPT_LIB=$L/$lib SRGPT_SKINNY_W8_MODE=1 SRGPT_DECODE_PREFETCH_ROUNDS=0 timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/r02i_step.txt
cat $OUT/r02i_step.txt
