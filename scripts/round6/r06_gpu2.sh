#!/bin/bash
# round 6, GPU call 3: the packed (MFMA-operand-order) weight layout of the batched decode products, TIMING ONLY: the tuning knob
# SRGPT_SKINNY_PACKED_TIMING=1 makes the kernel read row-major data as if it were packed (same bytes, wrong results)
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
{
for rep in 1 2; do
  for fmt in fp8 bf16; do
    for b in 8 4; do
    run old    $L/libsrgpt_hip_tuning_old.so "X=1" $b $fmt pub
    run rowmajor $L/libsrgpt_hip_tuning.so "X=1" $b $fmt pub
    run packed $L/libsrgpt_hip_tuning.so "SRGPT_SKINNY_PACKED_TIMING=1" $b $fmt pub
    done
  done
done
run old    $L/libsrgpt_hip_tuning_old.so "X=1" 16 fp8 pub
run packed $L/libsrgpt_hip_tuning.so "SRGPT_SKINNY_PACKED_TIMING=1" 16 fp8 pub
run old    $L/libsrgpt_hip_tuning_old.so "X=1" 2 fp8 pub
run packed $L/libsrgpt_hip_tuning.so "SRGPT_SKINNY_PACKED_TIMING=1" 2 fp8 pub
} > $OUT/r06_skinny_packed_timing.txt 2>&1
grep -E "==|qkv|o\+res|gateup|down|lm_head|sum" $OUT/r06_skinny_packed_timing.txt
