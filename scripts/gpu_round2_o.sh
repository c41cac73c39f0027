#!/bin/bash
# round 2, GPU call o: decode attention -- K/V rows fetched up front (PF = 2 / 3 / 4 key iterations) x split count, decode step time
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=spatialrgpt_amd
{
for lib in libsrgpt_hip_tuning.so libsrgpt_hip_tuning_pf3.so libsrgpt_hip_tuning_pf4.so; do
for sp in 4 8 16; do
  SRGPT_LIB=$L/$lib SRGPT_DECODE_MIN_SPLITS=$sp timeout 300 python scripts/ubench_decode_step.py bf16:1 bf16:4 bf16:8
done
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | sed -E "s/\{[^}]*MIN_SPLITS': '([0-9]+)'[^}]*\} \| /splits=\1 /" > $OUT/r02o_pf.txt
cat $OUT/r02o_pf.txt
