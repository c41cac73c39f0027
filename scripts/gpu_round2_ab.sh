#!/bin/bash
# round 2, GPU call ab: pipelined GEMV (first weight batch across the prologue, next batch before the reduction) -- parity subset,
# raw products and decode step vs the same sources with -DSRGPT_GEMV_PIPE=0
OUT=$PWD/gpurun_out
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -x -k "gemv or greedy or graph or decode" 2>&1 ) | tail -3
for rep in 1 2; do for lib in libsrgpt_hip_tuning_nopipe.so libsrgpt_hip_tuning.so; do for b in 1 2; do echo "== $lib batch $b rep$rep"; scripts/ubench_decode_mv spatialrgpt_amd/$lib $b bf16; done; done; done > $OUT/r02ab_mv.txt 2>&1
python - <<'PY'
import re
rows={}
cur=None
for l in open('gpurun_out/r02ab_mv.txt'):
    if l.startswith('=='):
        cur=l.strip('= \n'); rows[cur]=[]
    elif 'us' in l and cur:
        m=re.search(r'([\d.]+) us',l); rows[cur].append(float(m.group(1)))
print("%-52s %7s %7s %7s %7s %8s %7s"%("variant","qkv","o","gateup","down","lm_head","sum4"))
for k,v in sorted(rows.items(), key=lambda kv: (kv[0].split()[2], kv[0])):
    print("%-52s "%k+" ".join("%7.2f"%x for x in v))
PY
{
for lib in libsrgpt_hip_tuning_nopipe.so libsrgpt_hip_tuning.so libsrgpt_hip_tuning_nopipe.so libsrgpt_hip_tuning.so; do
  SRGPT_LIB=spatialrgpt_amd/$lib timeout 300 python scripts/ubench_decode_step.py bf16:1 bf16:2
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | sed -E "s/\{[^}]*\} \| //" > $OUT/r02ab_step.txt
cat $OUT/r02ab_step.txt
