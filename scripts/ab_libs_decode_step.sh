#!/bin/bash
# A/B of the whole graph-captured decode step between library builds, same box, interleaved twice:
#   scripts/ab_libs_decode_step.sh OUTFILE "fp8:8 bf16:4 ..." libA.so libB.so ...
OUT=$PWD/gpurun_out; mkdir -p $OUT
F=$OUT/$1; shift
CFGS=$1; shift
: > $F
for rep in 1 2; do
  for lib in "$@"; do
    SRGPT_LIB=$lib timeout 600 python scripts/ubench_decode_step.py $CFGS 2>&1 | grep "ms/step" | sed -E "s/ \{[^}]*\}//" >> $F
  done
done
cat $F
