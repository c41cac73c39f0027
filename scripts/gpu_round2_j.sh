#!/bin/bash
# round 2, GPU call j: SQ stall-share and LDS-conflict PMC passes over the decode products in the raw ubench (skinny kernel, old vs new)
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in ${LIBS:-old new}; do
  so=$R/spatialrgpt_amd/libsrgpt_hip_tuning.so; [ $lib = old ] && so=$R/spatialrgpt_amd/libsrgpt_hip_tuning_old.so
  for cfg in "8 fp8" "4 bf16"; do
    set -- $cfg
    tag=${lib}_b$1_$2
    rm -rf /tmp/pj
    SRGPT_SKINNY_W8_MODE=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d /tmp/pj -o run -- $R/scripts/ubench_decode_mv $so $1 $2 > /tmp/pj.log 2>&1
    echo "== $tag stalls" >> $OUT/r02j_pmc.txt
    python $R/scripts/pmc_summary.py $(find /tmp/pj -name "*.db" | head -1) SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES >> $OUT/r02j_pmc.txt 2>&1
    rm -rf /tmp/pj
    SRGPT_SKINNY_W8_MODE=1 timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES -d /tmp/pj -o run -- $R/scripts/ubench_decode_mv $so $1 $2 > /tmp/pj2.log 2>&1
    echo "== $tag insts" >> $OUT/r02j_pmc.txt
    python $R/scripts/pmc_summary.py $(find /tmp/pj -name "*.db" | head -1) SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES >> $OUT/r02j_pmc.txt 2>&1
    tail -2 /tmp/pj.log /tmp/pj2.log >> $OUT/r02j_logs.txt 2>&1
  done
done
cat $OUT/r02j_pmc.txt | cut -c1-300
