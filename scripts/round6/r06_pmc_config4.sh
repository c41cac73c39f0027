#!/bin/bash
# round 6: HBM bytes (FETCH_SIZE, own pass) and LDS activity of the batched fp8 decode kernels in the configs[4] per-GPU shape
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
SHORT="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 8 --preset config4"
rm -rf /tmp/prof_f; rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o run -- $SHORT > /tmp/f.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/prof_f -name "*.db" | head -1) FETCH_SIZE > $OUT/r06_pmc_fetch_fp8a8b8.txt 2>&1
rm -rf /tmp/prof_l; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d /tmp/prof_l -o run -- $SHORT > /tmp/l.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/prof_l -name "*.db" | head -1) SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES > $OUT/r06_pmc_lds_fp8a8b8.txt 2>&1
grep -E "skinny|dispatch" $OUT/r06_pmc_fetch_fp8a8b8.txt | cut -c1-200 | head; grep -E "skinny|dispatch" $OUT/r06_pmc_lds_fp8a8b8.txt | cut -c1-220 | head
