#!/bin/bash
# round 6: 3 / 4 resident blocks per CU for the packed fp8 products with a register-lean K loop (fragment batches of 2 k steps, one
# accumulator per sub-unit, activation fragments re-read per sub-unit; ONE sub-unit count compiled in: leanN is only right for shapes
# with N sub-units -- gate/up: 4 at 2 or 3 blocks per CU, 2 at 4; q/k/v: 2; o / down: 1).  TIMING ONLY on row-major data.
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2; do
    run packed $L/libsrgpt_hip_tuning.so "$P" 8 fp8 pub
    run lean4w2:bpc2 $L/libsrgpt_hip_tuning_lean4w2.so "$P" 8 fp8 pub
    for b in 2 3; do run lean4:bpc$b $L/libsrgpt_hip_tuning_lean4.so "$P SRGPT_SKINNY_BPC=$b" 8 fp8 pub; done
    for b in 2 3 4; do run lean2:bpc$b $L/libsrgpt_hip_tuning_lean2.so "$P SRGPT_SKINNY_BPC=$b" 8 fp8 pub; done
    for b in 2 3 4; do run lean2:w4:bpc$b $L/libsrgpt_hip_tuning_lean2.so "$P SRGPT_SKINNY_BPC=$b SRGPT_SKINNY_WAVES=4" 8 fp8 pub; done
    for b in 2 3 4; do run lean1:w4:bpc$b $L/libsrgpt_hip_tuning_lean1.so "$P SRGPT_SKINNY_BPC=$b SRGPT_SKINNY_WAVES=4" 8 fp8 pub; done
done
} > $OUT/r06_skinny_lean_occupancy.txt 2>&1
python3 scripts/round6/parse_mv.py $OUT/r06_skinny_lean_occupancy.txt
