#!/bin/bash
# PMC passes for the fp8 GEMM next to the W8A16 one (counters only: no trace domains in these runs)
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pm /tmp/pl
CMD="python $GRAFT_REPO_ROOT/scripts/ubench_gemm_w8a8.py --pmc"
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm -o run -- $CMD > /tmp/pm.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > $OUT/r02_w8a8_pmc_mfma.txt 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d /tmp/pl -o run -- $CMD > /tmp/pl.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/pl -name "*.db" | head -1) SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES > $OUT/r02_w8a8_pmc_lds.txt 2>&1
grep -i "gemm\|dispatches" $OUT/r02_w8a8_pmc_mfma.txt | cut -c1-200; grep -i "gemm\|dispatches" $OUT/r02_w8a8_pmc_lds.txt | cut -c1-220
