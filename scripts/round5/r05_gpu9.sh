#!/bin/bash
# round 5, GPU call 9: the whole-M kernel's 272 x 128 tile on the ViT products (M = 1458: six row tiles through grid.z) -- tuning-build probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SRGPT_LIB=$PWD/spatialrgpt_amd/libsrgpt_hip_tuning.so UBENCH_CHECK=1
SH="qkv:1458:3456:1152 out:1458:1152:1152 fc1:1458:4352:1152 fc2:1458:1152:4352 deconv1:729:4608:1152 deconv2:2916:4608:1152 proj1:196:4096:4608 connector:8:4096:1152"
{
for v in "" "SRGPT_GEMM_288=2 SRGPT_GEMM_288_ANYM=1" "SRGPT_GEMM_288=4 SRGPT_GEMM_288_ANYM=1" "SRGPT_GEMM_288=6 SRGPT_GEMM_288_ANYM=1"; do
  echo "## [$v]"; env $v python scripts/experiments/ubench_gemm.py $SH 2>&1 | grep -v -i "transformers\|amdgpu.ids"
done
} > gpurun_out/r05_vit_gemm288_probe.txt 2>&1
cat gpurun_out/r05_vit_gemm288_probe.txt
