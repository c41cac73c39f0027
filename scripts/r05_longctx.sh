#!/bin/bash
# round 5: long-context PERFORMANCE (VERDICT r4 missing #4): request lines at T = 2048 / 4000, decode step vs cached length,
# prefill by kernel at T = 2048 and 4000  ->  gpurun_out/r05_longctx_perf.txt
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT; F=$OUT/r05_longctx_perf.txt
{
echo "# long-context performance, VILA1.5-8B geometry, bf16, bs 1, one box (scripts/r05_longctx.sh)"
echo "# --- whole requests: python bench.py --prompt-len P (T = P - 1 + 196), 128 greedy tokens"
for P in 64 829 1853 3805; do
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --prompt-len $P 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(f\"prompt {$P} ids: {d['value']:.1f} tok/s, {d['ms_per_step']:.1f} ms per request, prefix {r['prefix_ms_per_call']:.2f} ms, decode {r['decode_ms_per_token']:.4f} ms/token\")"
done
echo "# --- decode step vs cached length (graph replay, 128 steps from T; scripts/ubench_decode_step.py weights:batch:T)"
timeout 900 python scripts/ubench_decode_step.py bf16:1:259 bf16:1:1024 bf16:1:2048 bf16:1:4000 bf16:4:2048 2>&1 | grep "ms/step" | sed -E "s/^[^|]*\| //"
timeout 900 python scripts/ubench_decode_step.py fp8:8:259 fp8:8:2048 2>&1 | grep "ms/step" | sed -E "s/^[^|]*\| //"
} > $F 2>&1
for P in 1853 3805; do
  TAG=r05_prefix_T$((P+195)) BENCH_ARGS="--prompt-len $P" LINES=24 bash scripts/prefix_trace.sh > /dev/null 2>&1
  echo "# --- request prefix by kernel, prompt $P ids (T = $((P+195)))" >> $F; head -24 $OUT/r05_prefix_T$((P+195)).txt | cut -c1-170 >> $F
done
cat $F
