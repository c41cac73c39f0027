#!/bin/bash
# round 6: decode attention whose K / V (and q) requests do not wait for pos[b] (SRGPT_DECODE_SPEC=1, same bits) and, as a timing
# probe with wrong numbers, whose RoPE table row does not wait either (SRGPT_DECODE_SPEC=300 = a fixed table row)
cd $GRAFT_REPO_ROOT
bash scripts/ab_decode_step.sh r06_attn_spec.txt "bf16:1 bf16:8 fp8:8 bf16:1:1800" "SRGPT_DECODE_SPEC=0" "SRGPT_DECODE_SPEC=1" "SRGPT_DECODE_SPEC=300" | cut -c1-200
for s in 0 1; do SRGPT_DECODE_SPEC=$s timeout 300 python scripts/experiments/ubench_decode_stamps.py 8 fp8_w8a8 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_attn_spec_stamps.txt 2>&1
cat gpurun_out/r06_attn_spec_stamps.txt
