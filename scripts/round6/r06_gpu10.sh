#!/bin/bash
# round 6: o_proj's packed tiles pulled into L2 by prefetch blocks of the attention launch (batched fp8 decode): loads in flight per prefetch
# wave 0 (off) / 1 / 2 / 4 / 8; + the per-layer outlier parity with the final bars
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
bash scripts/ab_decode_step.sh r06_decode_prefetch_tiles.txt "fp8:8 fp8:4 fp8:16" "SRGPT_DECODE_PREFETCH_TILES=0" "SRGPT_DECODE_PREFETCH_TILES=1" "SRGPT_DECODE_PREFETCH_TILES=2" "SRGPT_DECODE_PREFETCH_TILES=4" "SRGPT_DECODE_PREFETCH_TILES=8" > /dev/null 2>&1
cat $OUT/r06_decode_prefetch_tiles.txt
rm -f $OUT/r06_outlier_per_layer.json
( timeout 2400 python -m pytest tests/test_gpu_outlier_parity.py -q -k "alone_on_the_oracles_input" 2>&1 | tail -5 ) > $OUT/r06_t9_perlayer.log 2>&1
cat $OUT/r06_t9_perlayer.log
