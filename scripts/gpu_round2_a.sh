#!/bin/bash
# Round-2 GPU call A: the whole GPU suite (with the measured-error log), the two full-depth parity tests, the default bench
# line (full-depth CPU baseline), and the N>1 leg of bench.py on the single leased GPU (both ranks on device 0, gloo exchange).
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f $OUT/parity_measured.jsonl
export SRGPT_PARITY_LOG=$OUT/parity_measured.jsonl
( time timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 ) > $OUT/r02a_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r02a_tests.log
tail -5 $OUT/r02a_tests.log
unset SRGPT_PARITY_LOG
( time timeout 600 python bench.py --steps 5 --warmup 2 ) > $OUT/r02a_bench_default.log 2>&1
tail -2 $OUT/r02a_bench_default.log | cut -c1-600
( time timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 ) > $OUT/r02a_bench_gpus2.log 2>&1
tail -2 $OUT/r02a_bench_gpus2.log | cut -c1-900
nproc; free -g | head -2
