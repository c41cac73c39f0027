#!/bin/bash
# Where the waves of the batched decode product (skinny.hip) spend their cycles: SQ wave-cycle shares per kernel, own PMC passes
# (counters never share a run with trace domains).   gpurun -- 'bash scripts/experiments/pmc_skinny_stalls.sh'  -> gpurun_out/r03_pmc_skinny_stalls.txt
OUT=$PWD/gpurun_out/r03_pmc_skinny_stalls.txt; : > $OUT
export TMPDIR=/tmp; cd /tmp
for cfgname in "fp8_b8:--weights fp8 --batch 8" "bf16_b4:--batch 4"; do
  nm=${cfgname%%:*}; ar=${cfgname#*:}
  for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    rm -rf /tmp/prof_s
    rocprofv3 --pmc $pass -d /tmp/prof_s -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 4 $ar > /tmp/s.log 2>&1
    echo "== $nm: $pass" >> $OUT
    python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/prof_s -name "*.db" | head -1) $pass 2>&1 | grep -E "dispatches|skinny|decode_mfma|gemv" | cut -c1-230 >> $OUT
  done
done
cat $OUT
