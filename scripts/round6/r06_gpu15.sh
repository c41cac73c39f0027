#!/bin/bash
# round 6: o_proj's 16-row tiles of the ROW-MAJOR bf16 matrix pulled into L2 by the attention launch at 2+ rows (the tile prefetch of
# r06_gpu10 works for any matrix whose block p streams rows 16 p .. 16 p + 15); + the live resize_token_embeddings test
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_loader.py -x -q 2>&1 | tail -3 )
bash scripts/ab_decode_step.sh r06_decode_prefetch_tiles_bf16.txt "bf16:2 bf16:4 bf16:8" "SRGPT_DECODE_PREFETCH_TILES=0" "SRGPT_DECODE_PREFETCH_TILES=1" "SRGPT_DECODE_PREFETCH_TILES=2" "SRGPT_DECODE_PREFETCH_TILES=4" > /dev/null 2>&1
cat $OUT/r06_decode_prefetch_tiles_bf16.txt
