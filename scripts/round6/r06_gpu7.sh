#!/bin/bash
# round 6: more resident blocks per CU for the packed fp8 products (TIMING ONLY on row-major data): the kernel compiled for 3 / 4 waves per SIMD
# (168 / 128 VGPRs, spills) with 3 / 4 blocks of 4 waves per CU
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2; do
    run packed $L/libsrgpt_hip_tuning.so "$P" 8 fp8 pub
    run packed+w4 $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_WAVES=4" 8 fp8 pub
    run wps3+bpc3 $L/libsrgpt_hip_tuning_wps3.so "$P SRGPT_SKINNY_BPC=3" 8 fp8 pub
    run wps3+bpc3+w4 $L/libsrgpt_hip_tuning_wps3.so "$P SRGPT_SKINNY_BPC=3 SRGPT_SKINNY_WAVES=4" 8 fp8 pub
    run wps4+bpc4 $L/libsrgpt_hip_tuning_wps4.so "$P SRGPT_SKINNY_BPC=4" 8 fp8 pub
    run wps4+bpc4+w4 $L/libsrgpt_hip_tuning_wps4.so "$P SRGPT_SKINNY_BPC=4 SRGPT_SKINNY_WAVES=4" 8 fp8 pub
    run wps3+bpc2 $L/libsrgpt_hip_tuning_wps3.so "$P" 8 fp8 pub
done
} > $OUT/r06_skinny_occupancy.txt 2>&1
python3 scripts/round6/parse_mv.py $OUT/r06_skinny_occupancy.txt
