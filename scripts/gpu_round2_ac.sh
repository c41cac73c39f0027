#!/bin/bash
# round 2, GPU call ac: pipelined GEMVs (bf16 LDS / register variants, fp8) -- whole GPU suite, decode step vs the non-pipelined bf16 library
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 ) | tail -3
{
for lib in libsrgpt_hip_tuning_nopipe.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_nopipe.so libsrgpt_hip_tuning.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_decode_step.py bf16:1 bf16:2
done
SRGPT_LIB=spatialrgpt_amd/libsrgpt_hip_tuning.so timeout 300 python scripts/ubench_decode_step.py fp8:1 fp8:2 fp8:8
} 2>&1 | grep -v "Warning\|amdgpu.ids" | sed -E "s/\{[^}]*\} \| //" > $OUT/r02ac_step.txt
cat $OUT/r02ac_step.txt
