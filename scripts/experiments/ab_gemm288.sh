# A/B of the whole-M prefill GEMM (gemm288.hip) against the 96 x 128 tiles at the four Llama-3-8B prefill products -- tuning build.
# SRGPT_GEMM_288: 0 = small tiles, 1 = the product's rule, 2 = the whole-M kernel for every M <= 272, 2 + n = that with n K splits.
# (The lockstep form and its ablations in profiles/r04_gemm288.txt were template arguments of the kernel while it was being built;
# they are not in the tree.)
export SRGPT_LIB=$PWD/spatialrgpt_amd/libsrgpt_hip_tuning.so
SH=${AB_SHAPES:-"qkv:259:6144:4096 o:259:4096:4096 gate/up:259:28672:4096 down:259:4096:14336"}
for mode in ${AB_MODES:-0 2}; do
  echo "## SRGPT_GEMM_288=$mode"
  SRGPT_GEMM_288=$mode python scripts/experiments/ubench_gemm.py $SH 2>&1 | grep -v -i "transformers\|amdgpu.ids"
done
