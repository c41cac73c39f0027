#!/bin/bash
# round 2, GPU call ai: 96-row GEMM tiles in the dispatch -- whole GPU suite, per-shape table, default bench line
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 ) | tail -3
timeout 300 python scripts/ubench_gemm.py 2>&1 | grep -v Warn | tee $OUT/r02ai_gemm.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('default |', d['value'], 'tok/s | ms/step', d['ms_per_step'], '| decode ms/token', r['decode_ms_per_token'], '| frac', r['frac'])"
