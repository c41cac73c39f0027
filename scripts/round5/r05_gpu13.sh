#!/bin/bash
# round 5, GPU call 13: with the statistics published, does q/k/v still want one 8-wave block per CU?  4-wave blocks (two per CU), with
# and without 12-column ranges (6144 / 512 blocks), whole step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/ab_decode_step.sh r05_qkv_waves.txt "fp8:8 fp8:4 bf16:4 bf16:8" "" "SRGPT_SKINNY_WAVES=4" "SRGPT_SKINNY_WAVES=4 SRGPT_SKINNY_MINCW=12" "SRGPT_SKINNY_MINCW=8" > /dev/null 2>&1
cat gpurun_out/r05_qkv_waves.txt
