#!/bin/bash
# round 3, GPU call d: free-running greedy-id parity at full depth (configs[1], [2], [4] W8A16 / W8A8) on the peaked-margin weights
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 2400 python -m pytest tests/test_gpu_freerun_parity.py tests/test_gpu_kernels.py -q -s -k "free_running or composition" --durations=8 2>&1 ) | grep -v "amdgpu.ids" | tail -40 > $OUT/r03d_tests.txt; cat $OUT/r03d_tests.txt
free -g | head -2; nproc
