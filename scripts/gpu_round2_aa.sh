#!/bin/bash
# round 2, GPU call aa: wave_sum / wave_max on the VALU (DPP + permlane) everywhere -- whole GPU suite, then decode step A/B vs the
# ds_bpermute butterfly (libsrgpt_hip_tuning_bperm.so: same sources, -DSRGPT_WAVE_BPERMUTE) and the raw products
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 ) | tail -3
{
for lib in libsrgpt_hip_tuning_bperm.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_bperm.so libsrgpt_hip_tuning.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_decode_step.py bf16:1 bf16:2 bf16:4
done
for lib in libsrgpt_hip_tuning_bperm.so libsrgpt_hip_tuning.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_decode_step.py fp8:1 fp8:8
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | sed -E "s/\{[^}]*\} \| //" > $OUT/r02aa_step.txt
cat $OUT/r02aa_step.txt
for lib in libsrgpt_hip_tuning_bperm.so libsrgpt_hip_tuning.so; do echo "== $lib"; scripts/ubench_decode_mv spatialrgpt_amd/$lib 1 bf16; done > $OUT/r02aa_mv.txt 2>&1
cat $OUT/r02aa_mv.txt
