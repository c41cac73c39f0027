#!/bin/bash
# round 6: phase stamps of the decode attention kernel as shipped (bs=1 bf16 with the o_proj prefetch riding on the launch; 8 requests fp8 without it)
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
{ for a in "1 native" "8 native" "8 fp8_w8a8"; do for r in 1 2; do timeout 300 python scripts/experiments/ubench_decode_stamps.py $a 2>&1 | grep -v amdgpu.ids; done; done; } > $OUT/r06_attn_stamps.txt 2>&1
cat $OUT/r06_attn_stamps.txt
