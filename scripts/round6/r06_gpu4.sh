#!/bin/bash
# round 6: packed weight layout with a run-time granule size, TIMING ONLY (SRGPT_SKINNY_PACKED_TIMING=1: row-major data read as packed):
# T1 = granule rows of the single-tile products (o / down), TN = of the others (q/k/v, gate/up, lm_head); R16: 8-wave blocks for o / down
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2; do
  for fmt in fp8 bf16; do
    run old    $L/libsrgpt_hip_tuning_old.so "X=1" 8 $fmt pub
    for t1 in 4 8 16; do for tn in 4 8; do
      run "packed:T1=$t1:TN=$tn" $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_PACKED_T1=$t1 SRGPT_SKINNY_PACKED_TN=$tn" 8 $fmt pub
    done; done
    run "packed:T1=16:TN=4:R16" $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_W8_RES_COLS=16" 8 $fmt pub
    run "packed:T1=4:TN=4:R16" $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_PACKED_T1=4 SRGPT_SKINNY_W8_RES_COLS=16" 8 $fmt pub
  done
done
for b in 2 4 16; do
  run old    $L/libsrgpt_hip_tuning_old.so "X=1" $b fp8 pub
  run packed $L/libsrgpt_hip_tuning.so "$P" $b fp8 pub
done
run old    $L/libsrgpt_hip_tuning_old.so "X=1" 4 bf16 pub
run packed $L/libsrgpt_hip_tuning.so "$P" 4 bf16 pub
} > $OUT/r06_skinny_packed_gr.txt 2>&1
python3 scripts/round6/parse_mv.py $OUT/r06_skinny_packed_gr.txt
