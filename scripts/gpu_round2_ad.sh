#!/bin/bash
# round 2, GPU call ad: skinny kernel, first weight stage before the RMSNorm statistics are reduced -- parity subset, decode step A/B
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemv or fp8" 2>&1 ) | tail -2
{
for lib in libsrgpt_hip_tuning_nopre.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_nopre.so libsrgpt_hip_tuning.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_decode_step.py bf16:4 bf16:8
done
for lib in libsrgpt_hip_tuning_nopre.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_nopre.so libsrgpt_hip_tuning.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | sed -E "s/\{[^}]*\} \| //" > $OUT/r02ad_step.txt
cat $OUT/r02ad_step.txt
