#!/bin/bash
# round 2, GPU call n: MX MFMA probe; decode-attention split-count sweep at 8 and 4 sequences (tuning build)
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=spatialrgpt_amd
scripts/ubench_mfma_mx_probe > $OUT/r02n_mx_probe.txt 2>&1
cat $OUT/r02n_mx_probe.txt
{
for sp in 0 2 4 8 16; do
  SRGPT_LIB=$L/libsrgpt_hip_tuning.so SRGPT_DECODE_MIN_SPLITS=$sp timeout 300 python scripts/ubench_decode_step.py fp8:8 fp8:4
done
for sp in 0 2 4 8; do
  SRGPT_LIB=$L/libsrgpt_hip_tuning.so SRGPT_DECODE_MIN_SPLITS=$sp timeout 300 python scripts/ubench_decode_step.py bf16:4
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | sed -E "s/\{[^}]*MIN_SPLITS': '([0-9]+)'[^}]*\} \| /splits=\1 /" > $OUT/r02n_splits.txt
cat $OUT/r02n_splits.txt
