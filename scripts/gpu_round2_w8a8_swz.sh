#!/bin/bash
# after the fragment-read swizzle change: the fp8 tests, the GEMM table, the LDS-conflict pass
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_fp8_mfma.py -q 2>&1 | tail -2
timeout 100 python scripts/ubench_gemm_w8a8.py --quick 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_w8a8_swz_quick.txt; cat gpurun_out/r02_w8a8_swz_quick.txt | cut -c1-200
OUT=$PWD/gpurun_out; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pl
timeout 100 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d /tmp/pl -o run -- python $GRAFT_REPO_ROOT/scripts/ubench_gemm_w8a8.py --pmc > /tmp/pl.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/pl -name "*.db" | head -1) SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES > $OUT/r02_w8a8_pmc_lds_swz.txt 2>&1
grep -i "gemm\|dispatches" $OUT/r02_w8a8_pmc_lds_swz.txt | cut -c1-220
