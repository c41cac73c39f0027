#!/bin/bash
# round 2, GPU call m: whole GPU suite on the product build + the batched / fp8 bench lines
OUT=$PWD/gpurun_out
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 ) > $OUT/r02m_tests.log 2>&1
tail -6 $OUT/r02m_tests.log
for args in "" "--batch 4" "--weights fp8 --batch 8" "--batch 8" "--weights fp8"; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', '|', d['value'], 'tok/s | ms/step', d['ms_per_step'], '| decode ms/token', r['decode_ms_per_token'], '| dom frac', r['frac'], '| whole-step frac', r['decode_frac_whole_step'])"; done | tee $OUT/r02m_bench.txt
