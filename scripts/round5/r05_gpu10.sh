#!/bin/bash
# round 5, GPU call 10: ViT / extractor products at the batched shapes (8 and 16 images: configs[2] / configs[4] per-GPU share) --
# the shipped dispatch against gemm256 forced (tuning build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SRGPT_LIB=$PWD/spatialrgpt_amd/libsrgpt_hip_tuning.so
SH="qkv8:5832:3456:1152 out8:5832:1152:1152 fc1_8:5832:4352:1152 fc2_8:5832:1152:4352 qkv16:11664:3456:1152 out16:11664:1152:1152 fc1_16:11664:4352:1152 fc2_16:11664:1152:4352 dc1_8:2916:4608:1152 dc2_8:11664:4608:1152"
{
for v in "" "SRGPT_GEMM_FORCE_256=1" "SRGPT_GEMM_FORCE_256=-1"; do
  echo "## [$v]"; env $v python scripts/experiments/ubench_gemm.py $SH 2>&1 | grep -v -i "transformers\|amdgpu.ids"
done
} > gpurun_out/r05_vit_batched_gemm.txt 2>&1
cat gpurun_out/r05_vit_batched_gemm.txt
