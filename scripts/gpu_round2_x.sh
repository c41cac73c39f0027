#!/bin/bash
# round 2, GPU call x: decode attention with DPP / permlane reductions -- parity subset, phase stamps, decode step vs HEAD's attn (old lib)
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py -q -x -k "decode or attention or greedy or graph or fp8 or oracle" 2>&1 ) | tail -4
for b in 1 8; do timeout 300 python scripts/ubench_decode_stamps.py $b 2>&1 | grep -v Warn | tail -14; done > $OUT/r02x_stamps.txt
cat $OUT/r02x_stamps.txt
{
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_old.so libsrgpt_hip_tuning.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_decode_step.py bf16:1 bf16:4 bf16:8
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | sed -E "s/\{[^}]*\} \| //" > $OUT/r02x_step.txt
cat $OUT/r02x_step.txt
