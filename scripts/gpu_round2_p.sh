#!/bin/bash
# round 2, GPU call p: flash attention with the next K/V tile prefetched -- parity subset + same-box A/B (old = tuning_old's attn)
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or attn or flash" 2>&1 ) | tail -3 > $OUT/r02p_tests.log
cat $OUT/r02p_tests.log
{
for rep in 1 2; do
for lib in libsrgpt_hip_tuning_old.so libsrgpt_hip.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_attention.py
done
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/r02p_attn.txt
cat $OUT/r02p_attn.txt
