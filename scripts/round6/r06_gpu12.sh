#!/bin/bash
# round 6: gemm256 persistent over tiles (one block per CU walks its XCD's run; the next tile's prologue requests go out before the
# finished tile's epilogue stores) against one block per tile, at the batched ViT / extractor / prefill shapes + correctness
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SRGPT_LIB=$PWD/spatialrgpt_amd/libsrgpt_hip_tuning.so
SH="qkv8:5832:3456:1152 out8:5832:1152:1152 fc1_8:5832:4352:1152 fc2_8:5832:1152:4352 qkv16:11664:3456:1152 out16:11664:1152:1152 fc1_16:11664:4352:1152 fc2_16:11664:1152:4352 dc1_8:2916:4608:1152 dc2_8:11664:4608:1152 pre_qkv8:2072:6144:4096 pre_gu8:2072:28672:4096 pre_down8:2072:4096:14336 sq4096:4096:4096:4096 sq8192:8192:8192:8192"
{
for rep in 1 2; do
for v in "SRGPT_GEMM_FORCE_256=1 SRGPT_GEMM_PERSIST=0" "SRGPT_GEMM_FORCE_256=1 SRGPT_GEMM_PERSIST=1" "SRGPT_GEMM_PERSIST=0" "SRGPT_GEMM_PERSIST=1"; do
  echo "## [$v]"; env $v UBENCH_CHECK=1 python scripts/experiments/ubench_gemm.py $SH 2>&1 | grep -v -i "transformers\|amdgpu.ids"
done; done
} > gpurun_out/r06_gemm256_persistent.txt 2>&1
cat gpurun_out/r06_gemm256_persistent.txt | grep -v check | head -80
grep check gpurun_out/r06_gemm256_persistent.txt | sort | uniq -c | head
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -4 )
