#!/bin/bash
OUT=$PWD/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -3
for d in 1 0; do echo "== DIM=$d"; SRGPT_GEMM256_DIM=$d SRGPT_GEMM_FORCE_256=1 timeout 120 python scripts/ubench_gemm_big.py --tuning --only "sq4096,sq8192,vit b8 fc1,vit b8 qkv,prefill b8 gate/up,prefill b8 down" 2>&1 | grep -v amdgpu; done | tee $OUT/r02f_dim.txt
SRGPT_GEMM256_DIM=1 timeout 120 python scripts/ubench_gemm256_ts.py 2>&1 | grep -v amdgpu | head -20 | tee $OUT/r02f_ts.txt
