#!/bin/bash
# round 2, GPU call r: the round profile of the final build (kernel trace + PMC passes) and one bench line per config
bash scripts/profile_round.sh r02_final > gpurun_out/r02r_profile.log 2>&1
bash scripts/run_configs.sh > gpurun_out/r02_configs.txt 2>&1
tail -20 gpurun_out/r02_configs.txt
