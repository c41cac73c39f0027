#!/bin/bash
# round 5, GPU call 4: outlier-statistics parity suite, beam-sample / flag matrix, output class, sampler status
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r05_outlier_parity.json
( timeout 1500 python -m pytest tests/test_gpu_outlier_parity.py -q 2>&1 | tail -40 ) > gpurun_out/r05_t4a.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_pipeline.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -15 ) > gpurun_out/r05_t4b.log 2>&1
cat gpurun_out/r05_t4a.log gpurun_out/r05_t4b.log
