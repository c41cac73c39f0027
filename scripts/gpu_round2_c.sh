#!/bin/bash
# gemm256: compile-time ablations + SQ counters for the square shape
OUT=$PWD/gpurun_out
bash scripts/ubench_gemm256_ablate.sh > $OUT/r02c_ablate.txt 2>&1
cat $OUT/r02c_ablate.txt
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pmc_g
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pmc_g -o run -- python $GRAFT_REPO_ROOT/scripts/ubench_gemm_big.py --only "sq4096,vit b8 fc1" > /tmp/pmc_g.log 2>&1
tail -3 /tmp/pmc_g.log
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/pmc_g -name "*.db" | head -1) SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | grep -i -E "gemm|kernel|name" | head -20 | tee $OUT/r02c_pmc_gemm256.txt
