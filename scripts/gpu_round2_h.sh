#!/bin/bash
# round 2, GPU call h: the fp8 skinny kernel's 512-k form (parity subset on the product build), per-product timings of the
# build variants (ring depth 3, fragment batches of 2 / 8 k steps), decode-step A/B of the W8 form and of the o_proj L2 prefetch at 3+ rows
OUT=$PWD/gpurun_out
mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemv" 2>&1 ) > $OUT/r02h_tests.log 2>&1
tail -4 $OUT/r02h_tests.log
L=spatialrgpt_amd
{
for lib in libsrgpt_hip_tuning.so libsrgpt_hip_tuning_d3.so libsrgpt_hip_tuning_fs2.so libsrgpt_hip_tuning_fs8.so; do
  for cfg in "8 fp8" "4 fp8" "4 bf16" "8 bf16"; do
    for mode in 2 1; do
      set -- $cfg
      [ "$2" = bf16 ] && [ $mode = 1 ] && continue
      [ $lib != libsrgpt_hip_tuning.so ] && [ $mode = 1 ] && continue
      echo "== $lib batch $1 $2 w8_mode=$mode"
      SRGPT_SKINNY_W8_MODE=$mode timeout 120 scripts/ubench_decode_mv $L/$lib $1 $2
    done
  done
done
} > $OUT/r02h_mv.txt 2>&1
tail -30 $OUT/r02h_mv.txt
{
SRGPT_LIB=$L/libsrgpt_hip_tuning.so SRGPT_SKINNY_W8_MODE=1 SRGPT_DECODE_PREFETCH_ROUNDS=2 timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
SRGPT_LIB=$L/libsrgpt_hip_tuning.so SRGPT_SKINNY_W8_MODE=2 SRGPT_DECODE_PREFETCH_ROUNDS=2 timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
SRGPT_LIB=$L/libsrgpt_hip_tuning.so SRGPT_SKINNY_W8_MODE=2 SRGPT_DECODE_PREFETCH_ROUNDS=0 timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
SRGPT_LIB=$L/libsrgpt_hip_tuning.so SRGPT_DECODE_PREFETCH_ROUNDS=2 timeout 300 python scripts/ubench_decode_step.py bf16:4 bf16:8 bf16:1
SRGPT_LIB=$L/libsrgpt_hip_tuning.so SRGPT_DECODE_PREFETCH_ROUNDS=0 timeout 300 python scripts/ubench_decode_step.py bf16:4 bf16:8
} 2>&1 | grep -v Warning > $OUT/r02h_step.txt
cat $OUT/r02h_step.txt
