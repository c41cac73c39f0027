#!/bin/bash
# round 6, GPU call 1: batched decode products -- is the single-tile product (o_proj / down_proj) bound by stages in flight per CU?
#   old = round 5's kernel; tuning = activation ring (x requested as far ahead as the weight ring reaches), weight ring depth 2;
#   d3 / d4 = ring depth 3 / 4 (fp8); W8 = 8-wave blocks for every product (SRGPT_SKINNY_WAVES=8); R16 = 8-wave blocks for the
#   residual products that own <= 16 columns per CU (SRGPT_SKINNY_W8_RES_COLS=16)
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { # label lib env batch fmt mode
  echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
{
for rep in 1 2; do
  for fmt in fp8 bf16; do
    run old      $L/libsrgpt_hip_tuning_old.so "X=1" 8 $fmt pub
    run tuning   $L/libsrgpt_hip_tuning.so "X=1" 8 $fmt pub
    run tuning+R16 $L/libsrgpt_hip_tuning.so "SRGPT_SKINNY_W8_RES_COLS=16" 8 $fmt pub
    run tuning+W8 $L/libsrgpt_hip_tuning.so "SRGPT_SKINNY_WAVES=8" 8 $fmt pub
    [ $fmt = fp8 ] && for d in d3 d4; do
      run $d     $L/libsrgpt_hip_tuning_$d.so "X=1" 8 $fmt pub
      run $d+R16 $L/libsrgpt_hip_tuning_$d.so "SRGPT_SKINNY_W8_RES_COLS=16" 8 $fmt pub
    done
  done
done
for b in 2 4 16; do
  run old      $L/libsrgpt_hip_tuning_old.so "X=1" $b fp8 pub
  run tuning+R16 $L/libsrgpt_hip_tuning.so "SRGPT_SKINNY_W8_RES_COLS=16" $b fp8 pub
  run d3+R16 $L/libsrgpt_hip_tuning_d3.so "SRGPT_SKINNY_W8_RES_COLS=16" $b fp8 pub
done
} > $OUT/r06_skinny_inflight.txt 2>&1
# correctness of the new paths (8-wave publishing epilogue, activation ring) on the variants that may ship
for v in "tuning:SRGPT_SKINNY_W8_RES_COLS=16" "d3:SRGPT_SKINNY_W8_RES_COLS=16" "tuning:X=1"; do
  lib=${v%%:*}; e=${v#*:}; f=$L/libsrgpt_hip_tuning_$lib.so; [ $lib = tuning ] && f=$L/libsrgpt_hip_tuning.so
  ( env $e SRGPT_TEST_LIB=$f timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv or rowss or batched_decode or skinny" 2>&1 | tail -3 ) > $OUT/r06_t1_${lib}_${e%%=*}.log 2>&1
done
bash scripts/ab_libs_decode_step.sh r06_skinny_inflight_step.txt "fp8:8 bf16:8" $L/libsrgpt_hip_tuning_old.so $L/libsrgpt_hip_tuning.so $L/libsrgpt_hip_tuning_d3.so > /dev/null 2>&1
grep -E "==|qkv|o\+res|gateup|down|sum" $OUT/r06_skinny_inflight.txt; cat $OUT/r06_t1_*.log; cat $OUT/r06_skinny_inflight_step.txt
