#!/bin/bash
# Round-2 GPU call B: whole GPU suite (no -x), GEMM microbench new vs old kernels, batched bench lines.
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/parity_measured.jsonl
export SRGPT_PARITY_LOG=$OUT/parity_measured.jsonl
( time timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 ) > $OUT/r02b_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r02b_tests.log
grep -E "passed|failed|FAILED|ERROR" $OUT/r02b_tests.log | tail -30
unset SRGPT_PARITY_LOG
echo "== new (product build)"; timeout 300 python scripts/ubench_gemm_big.py 2>&1 | tee $OUT/r02b_gemm_new.txt
echo "== old kernels (tuning build, 256 kernel off)"; SRGPT_GEMM_FORCE_256=-1 timeout 300 python scripts/ubench_gemm_big.py --tuning 2>&1 | tee $OUT/r02b_gemm_old.txt
for b in 4 8; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | tail -1 | cut -c1-220; done | tee $OUT/r02b_bench_batch.txt
