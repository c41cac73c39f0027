#!/bin/bash
# round 5, GPU call 8: the whole GPU suite + smoke + default bench of the current tree, the round profile (kernel trace + PMC passes),
# one bench line per BASELINE config
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/gpu_suite.sh r05_suite > /dev/null 2>&1
bash scripts/profile_round.sh r05_final > /dev/null 2>&1
bash scripts/run_configs.sh r05 > gpurun_out/r05_configs.txt 2>&1
cat gpurun_out/r05_suite_tests.txt; cat gpurun_out/r05_suite_bench.json; cat gpurun_out/r05_configs.txt; head -14 gpurun_out/r05_final_kernel_stats.txt | cut -c1-150
