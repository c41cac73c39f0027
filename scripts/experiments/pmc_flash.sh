set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z0-9_]+" | sort -u > $OUT/sq_counters.txt
: > $OUT/r04e_flash_pmc.txt
for v2 in 1; do
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/prof_p
  AB_SHAPE=${AB_SHAPE:-1} rocprofv3 --pmc $set -d /tmp/prof_p -o run -- python $GRAFT_REPO_ROOT/scripts/experiments/ab_flash.py > /tmp/p.log 2>&1
  echo "== $set" >> $OUT/r04e_flash_pmc.txt
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/prof_p -name "*.db" | head -1) $set 2>&1 | grep -i -E "flash|kernel|name" | head -4 >> $OUT/r04e_flash_pmc.txt || tail -3 /tmp/p.log >> $OUT/r04e_flash_pmc.txt
 done
done
cat $OUT/r04e_flash_pmc.txt
