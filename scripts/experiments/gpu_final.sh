#!/bin/bash
# The end-of-round measurement of a build: whole GPU suite + smoke + default bench (gpu_suite.sh), the round profile (trace + PMC passes),
# one bench line per config.   gpurun --timeout 3000 -- 'bash scripts/experiments/gpu_final.sh r04'
R=${1:-r04}; OUT=$PWD/gpurun_out; mkdir -p $OUT
bash scripts/gpu_suite.sh ${R}_suite
bash scripts/profile_round.sh ${R}_final > $OUT/${R}_profile.log 2>&1; tail -2 $OUT/${R}_profile.log
bash scripts/run_configs.sh $R > $OUT/${R}_configs.txt 2>&1; cat $OUT/${R}_configs.txt
