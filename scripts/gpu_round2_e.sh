#!/bin/bash
OUT=$PWD/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -3
echo "== product build"; timeout 200 python scripts/ubench_gemm_big.py 2>&1 | grep -v amdgpu | tee $OUT/r02h_gemm_new.txt
ABLATES="9 4" bash scripts/ubench_gemm256_ablate.sh 2>&1 | tee $OUT/r02h_ablate.txt
