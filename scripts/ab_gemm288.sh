# A/B of the whole-M prefill GEMM (gemm288.hip) against the 96 x 128 tiles at the four Llama-3-8B prefill products -- tuning build.
# SRGPT_GEMM_288: 0 = small tiles, 1 = the product's rule, 2 = the whole-M kernel for every M <= 272;
# SRGPT_GEMM288_ABLATE: 7 = lockstep form, 1 / 4 / 5 / 6 = no requests / no MFMA / no barrier / no waits in the lockstep form (wrong results)
export SRGPT_LIB=$PWD/spatialrgpt_amd/libsrgpt_hip_tuning.so
SH=${AB_SHAPES:-"qkv:259:6144:4096 o:259:4096:4096 gate/up:259:28672:4096 down:259:4096:14336"}
for mode in ${AB_MODES:-0 2}; do
  for ab in ${AB_ABLATE:-0}; do
    echo "## SRGPT_GEMM_288=$mode SRGPT_GEMM288_ABLATE=$ab"
    SRGPT_GEMM_288=$mode SRGPT_GEMM288_ABLATE=$ab python scripts/ubench_gemm.py $SH 2>&1 | grep -v -i "transformers\|amdgpu.ids"
  done
done
