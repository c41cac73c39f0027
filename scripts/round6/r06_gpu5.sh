#!/bin/bash
# round 6: the packed decode layout in the product -- tests of the new paths, then the whole decode step packed vs row-major (same library)
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "pack or rowss or batched_decode or gemv or skinny" 2>&1 | tail -15 ) > $OUT/r06_t5_kernels.log 2>&1
cat $OUT/r06_t5_kernels.log
F=$OUT/r06_decode_step_packed.txt; : > $F
for rep in 1 2; do
  for lay in rowmajor packed; do
    UBENCH_DECODE_LAYOUT=$lay timeout 900 python scripts/ubench_decode_step.py fp8:8 fp8:4 fp8:2 fp8:16 2>&1 | grep "ms/step" | sed -E "s/^[^|]*\| //" | sed "s|^|[$lay] |" >> $F
  done
done
cat $F
