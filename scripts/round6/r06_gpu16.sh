#!/bin/bash
# round 6: down_proj at one row (K = 14336) on the register-resident GEMV (activations in 112 VGPRs, no LDS prologue / barrier / ds_read per chunk)
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
( SRGPT_GEMV_REG_LONG=1 SRGPT_TEST_LIB=spatialrgpt_amd/libsrgpt_hip_tuning.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv" 2>&1 | tail -3 )
bash scripts/ab_decode_step.sh r06_gemv_reg_long.txt "bf16:1" "SRGPT_GEMV_REG_LONG=0" "SRGPT_GEMV_REG_LONG=1" > /dev/null 2>&1
bash scripts/ab_decode_step.sh r06_gemv_reg_long_b.txt "bf16:1" "SRGPT_GEMV_REG_LONG=0" "SRGPT_GEMV_REG_LONG=1" > /dev/null 2>&1
cat $OUT/r06_gemv_reg_long.txt $OUT/r06_gemv_reg_long_b.txt
export TMPDIR=/tmp; cd /tmp
for v in 0 1; do rm -rf /tmp/prof_c
SRGPT_GEMV_REG_LONG=$v SRGPT_LIB=$GRAFT_REPO_ROOT/spatialrgpt_amd/libsrgpt_hip_tuning.so rocprofv3 --kernel-trace -d /tmp/prof_c -o run -- python $GRAFT_REPO_ROOT/scripts/ubench_decode_step.py bf16:1 > /tmp/c.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_c -name "*.db" | head -1) 8 | cut -c1-150 | grep -E "gemv|calls"
done
