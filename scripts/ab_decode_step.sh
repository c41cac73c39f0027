#!/bin/bash
# A/B of the whole graph-captured decode step under knob settings of the TUNING build (libsrgpt_hip_tuning.so), same box:
#   scripts/ab_decode_step.sh OUTFILE "bf16:1 [bf16:4 ...]" "KNOB=v KNOB=v" "KNOB=v" ...
# every variant (an env-assignment string; "" = defaults) runs in its own process (knobs are read once per process), the whole list
# is run twice interleaved so box drift shows.  Output: one line per (variant, weights:batch) -> gpurun_out/OUTFILE.
OUT=$PWD/gpurun_out; mkdir -p $OUT
F=$OUT/$1; shift
CFGS=$1; shift
: > $F
for rep in 1 2; do
  for v in "$@"; do
    env $v SRGPT_LIB=spatialrgpt_amd/libsrgpt_hip_tuning.so timeout 600 python scripts/ubench_decode_step.py $CFGS 2>&1 \
      | grep "ms/step" | sed -E "s/^[^|]*\| //" | sed "s|^|[$v] |" >> $F
  done
done
cat $F
