#!/bin/bash
# round 5, GPU call 5: outlier suite (two severities) + host-surface tests again + long-context performance table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r05_outlier_parity.json
( timeout 1800 python -m pytest tests/test_gpu_outlier_parity.py -q 2>&1 | tail -30 ) > gpurun_out/r05_t5a.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_pipeline.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -15 ) > gpurun_out/r05_t5b.log 2>&1
tail -5 gpurun_out/r05_t5a.log; tail -5 gpurun_out/r05_t5b.log
bash scripts/r05_longctx.sh
