#!/bin/bash
# round 5, GPU call 7: blocks per CU of the short one-row GEMV launches (q/k/v, o_proj), whole step at bs 1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/ab_decode_step.sh r05_gemv_bpc.txt "bf16:1 fp8:1" "" "SRGPT_GEMV_BPC_SHORT=3" "SRGPT_GEMV_BPC_SHORT=4" "SRGPT_GEMV_BPC_SHORT=5" "SRGPT_GEMV_BLOCKS_PER_CU=3" "SRGPT_GEMV_BLOCKS_PER_CU=4" > /dev/null 2>&1
cat gpurun_out/r05_gemv_bpc.txt
