#!/bin/bash
# round 5, GPU call 1: the row-statistics hand-off (tests, whole-step A/B, phase stamps) and the 7-load GEMV batches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "rowss or gemv or decode_step or fp8_decode" 2>&1 | tail -15 ) > gpurun_out/r05_t1.log 2>&1
for m in "8 fp8" "8 fp8 pub" "8 bf16" "8 bf16 pub" "4 bf16" "4 bf16 pub"; do timeout 300 python scripts/ubench_skinny_stamps.py $m 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_skinny_stamps.txt
bash scripts/ab_decode_step.sh r05_rowss_ab.txt "fp8:8 fp8:4 fp8:16 bf16:8 bf16:4 bf16:2 bf16:1" "SRGPT_DECODE_ROWSS=0 SRGPT_GEMV_U7=0" "SRGPT_DECODE_ROWSS=1 SRGPT_GEMV_U7=1" > /dev/null 2>&1
cat gpurun_out/r05_t1.log gpurun_out/r05_skinny_stamps.txt gpurun_out/r05_rowss_ab.txt
