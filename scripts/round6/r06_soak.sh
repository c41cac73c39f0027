#!/bin/bash
# round 6, final tree: soak / race screen with the packed decode layout live (fp8 formats) and the new beam KV reorder kernel (beam case)
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
{ for fmt in native fp8 fp8_w8a8; do timeout 900 python scripts/soak.py $fmt 600 2>&1 | tail -4; done; } > $OUT/r06_soak.txt 2>&1
cat $OUT/r06_soak.txt
