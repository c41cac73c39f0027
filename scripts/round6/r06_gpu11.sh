#!/bin/bash
# round 6: gate/up as ONE 8-wave block per CU that walks K once (8 sub-units per pass) vs two 4-wave blocks per CU; TIMING ONLY (packed knob)
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2; do
  for b in 8 4; do
    run packed $L/libsrgpt_hip_tuning.so "$P" $b fp8 pub
    run packed+w8swiglu:2pass $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_W8_SWIGLU=1" $b fp8 pub
    run m8:w8swiglu $L/libsrgpt_hip_tuning_m8.so "$P SRGPT_SKINNY_W8_SWIGLU=1" $b fp8 pub
    run m8a:w8swiglu $L/libsrgpt_hip_tuning_m8a.so "$P SRGPT_SKINNY_W8_SWIGLU=1" $b fp8 pub
    run m8:w8swiglu:gr8 $L/libsrgpt_hip_tuning_m8.so "$P SRGPT_SKINNY_W8_SWIGLU=1 SRGPT_SKINNY_PACKED_TN=8" $b fp8 pub
  done
  run rowmajor $L/libsrgpt_hip_tuning.so "X=1" 4 bf16 pub
  run m8:w8swiglu:rowmajor $L/libsrgpt_hip_tuning_m8.so "SRGPT_SKINNY_W8_SWIGLU=1" 4 bf16 pub
done
} > $OUT/r06_skinny_gateup_8wave.txt 2>&1
python3 scripts/round6/parse_mv.py $OUT/r06_skinny_gateup_8wave.txt
