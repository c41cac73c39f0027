#!/bin/bash
# round 3, GPU call h: the final build -- region pooling check, full GPU suite, round profile (trace + PMC passes), config lines
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py -x -q -k "region or raw_uint8" 2>&1 ) | grep -v amdgpu.ids | tail -4
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 2>&1 ) | grep -v amdgpu.ids | tail -25 > $OUT/r03h_tests.txt; cat $OUT/r03h_tests.txt
bash scripts/profile_round.sh r03_final > $OUT/r03h_profile.log 2>&1; tail -3 $OUT/r03h_profile.log
grep -i "region\|decode_\|calls" $OUT/r03_final_prefix.txt | head; head -3 $OUT/r03_final_prefix.txt
bash scripts/run_configs.sh r03 > $OUT/r03_configs.txt 2>&1; cat $OUT/r03_configs.txt
