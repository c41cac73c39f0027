#!/bin/bash
# round 6: after limiting the activation ring to fp8 and the register-form RMSNorm staging to the published-statistics instances of up to 8 rows:
# per product against round 5's kernel (old) and whole step
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2; do
  for b in 2 4 8 16; do
    run old:rowmajor $L/libsrgpt_hip_tuning_old.so "X=1" $b bf16 pub
    run new:rowmajor $L/libsrgpt_hip_tuning.so "X=1" $b bf16 pub
  done
  run old:rowmajor $L/libsrgpt_hip_tuning_old.so "X=1" 8 fp8 pub
  run new:packed  $L/libsrgpt_hip_tuning.so "$P" 8 fp8 pub
done
} > $OUT/r06_skinny_final_products.txt 2>&1
python3 scripts/round6/parse_mv.py $OUT/r06_skinny_final_products.txt
F=$OUT/r06_decode_step_final.txt; : > $F
for rep in 1 2; do
  timeout 900 python scripts/ubench_decode_step.py fp8:8 fp8:4 fp8:2 fp8:16 bf16:1 bf16:2 bf16:4 bf16:8 bf16:16 2>&1 | grep "ms/step" | sed -E "s/^[^|]*\| //" >> $F
done
cat $F
