#!/bin/bash
# round 3, GPU call a: full GPU suite on the fused attention + o_proj decode step, then the same-box A/B of the step
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 ) | tail -15 > $OUT/r03a_tests.txt; cat $OUT/r03a_tests.txt
scripts/ab_decode_step.sh r03a_step.txt "bf16:1" \
  "SRGPT_DECODE_FUSE_OPROJ=0" \
  "SRGPT_DECODE_FUSE_OPROJ=1 SRGPT_DECODE_PF_GATEUP_ROUNDS=0" \
  "SRGPT_DECODE_FUSE_OPROJ=1 SRGPT_DECODE_PF_GATEUP_ROUNDS=1" \
  "SRGPT_DECODE_FUSE_OPROJ=1 SRGPT_DECODE_PF_GATEUP_ROUNDS=1 SRGPT_DECODE_PF_GATEUP_PREFIX=4096"
( timeout 600 python bench.py 2>$OUT/r03a_bench.err ) | tail -1 > $OUT/r03a_bench.json; cat $OUT/r03a_bench.json
