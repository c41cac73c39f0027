#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py -x -q -k "graph or greedy or composition or region" 2>&1 ) | grep -v amdgpu.ids | tail -5
( timeout 1200 python -m pytest tests/test_gpu_freerun_parity.py -q -s -k config3 2>&1 ) | grep -v "amdgpu.ids" | grep -E "FREERUN|passed|failed|Error|assert" | cut -c1-1200
scripts/ab_decode_step.sh r03l_step.txt "bf16:1" "SRGPT_DECODE_MFMA=1"
