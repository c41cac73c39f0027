#!/bin/bash
# round 5, GPU call 3: device-side splice -- tests, request prefix (wall vs GPU-busy) at bs 1 and at configs[4]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_splice.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_loader.py -x -q 2>&1 | tail -15 ) > gpurun_out/r05_t3.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_bs1.json 2> gpurun_out/r05_bench_bs1.err
TAG=r05_prefix_bs1 bash scripts/prefix_trace.sh > /dev/null 2>&1
TAG=r05_prefix_config4 BENCH_ARGS="--preset config4" bash scripts/prefix_trace.sh > /dev/null 2>&1
cat gpurun_out/r05_t3.log; cat gpurun_out/r05_bench_bs1.json; tail -3 gpurun_out/r05_bench_bs1.err; head -12 gpurun_out/r05_prefix_bs1.txt | cut -c1-200; head -12 gpurun_out/r05_prefix_config4.txt | cut -c1-200
