#!/bin/bash
# timing-only ablations of gemm256 (tuning build): which resource bounds the K loop?
for a in ${ABLATES:-0 1 2 3 4 5 6 7}; do
  echo "== ablate=$a (0 full, 1 no DMA in loop, 2 no DMA waits, 3 no fragment reads, 4 no MFMA, 5 no setprio, 6 late = odd waves, 7 lockstep)"
  SRGPT_GEMM256_ABLATE=$a SRGPT_GEMM_FORCE_256=1 timeout 120 python scripts/ubench_gemm_big.py --tuning --only "sq4096,vit b8 fc1,prefill b8 gate/up" 2>&1 | grep -v amdgpu.ids
done
