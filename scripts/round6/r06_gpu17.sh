#!/bin/bash
# round 6: does specialising the batched kernel to ONE sub-unit count (no pass switch, 140 - 180 instead of 212 - 244 VGPRs, a quarter of the
# code) change anything?  fixN is only right for the products with N sub-units: o / down 1, q/k/v 2, gate/up 4.  TIMING ONLY (packed knob).
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2 3; do
  run generic $L/libsrgpt_hip_tuning.so "$P" 8 fp8 pub
  run fix1 $L/libsrgpt_hip_tuning_fix1.so "$P" 8 fp8 pub
  run fix2 $L/libsrgpt_hip_tuning_fix2.so "$P" 8 fp8 pub
  run fix4 $L/libsrgpt_hip_tuning_fix4.so "$P" 8 fp8 pub
done
} > $OUT/r06_skinny_specialised.txt 2>&1
python3 scripts/round6/parse_mv.py $OUT/r06_skinny_specialised.txt
