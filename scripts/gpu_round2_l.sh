#!/bin/bash
# round 2, GPU call l: balanced column partition of the skinny kernel -- parity subset, then same-box A/B vs HEAD's kernel
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=spatialrgpt_amd
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemv or fp8" 2>&1 ) > $OUT/r02l_tests.log 2>&1
tail -4 $OUT/r02l_tests.log
{
for rep in 1 2; do
for v in old new8 new4; do
  lib=libsrgpt_hip_tuning.so; [ $v = old ] && lib=libsrgpt_hip_tuning_old.so
  w=8; [ $v = new4 ] && w=4
  for cfg in "8 fp8" "4 fp8" "4 bf16" "8 bf16" "16 bf16"; do
      set -- $cfg
      echo "== $v batch $1 $2 rep$rep"
      SRGPT_SKINNY_WAVES=$w timeout 120 scripts/ubench_decode_mv $L/$lib $1 $2
  done
done
done
} > $OUT/r02l_mv.txt 2>&1
{
for v in old new8 new4 old new8; do
  lib=libsrgpt_hip_tuning.so; [ $v = old ] && lib=libsrgpt_hip_tuning_old.so
  w=8; [ $v = new4 ] && w=4
  echo "-- $v"
  SRGPT_LIB=$L/$lib SRGPT_SKINNY_WAVES=$w timeout 300 python scripts/ubench_decode_step.py bf16:4 bf16:8
done
for v in old new8 new4; do
  lib=libsrgpt_hip_tuning.so; [ $v = old ] && lib=libsrgpt_hip_tuning_old.so
  w=8; [ $v = new4 ] && w=4
  echo "-- $v"
  SRGPT_LIB=$L/$lib SRGPT_SKINNY_WAVES=$w timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/r02l_step.txt
cat $OUT/r02l_step.txt
