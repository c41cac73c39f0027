#!/bin/bash
# round 3, GPU call i: region pooling v4 (4 double-buffered passes per block, 13 row slabs): tests + kernel trace
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -x -q -k "region or raw_uint8 or pipeline or stage or golden" 2>&1 ) | grep -v amdgpu.ids | tail -5 > $OUT/r03i_tests.txt; cat $OUT/r03i_tests.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --max-new-tokens 4 > /tmp/x.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 60 | cut -c1-170 > $OUT/r03i_kernels.txt
grep -i "region\|calls" $OUT/r03i_kernels.txt
rm -rf /tmp/prof_y; rocprofv3 --kernel-trace -d /tmp/prof_y -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --max-new-tokens 4 --preset config3 > /tmp/y.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_y -name "*.db" | head -1) 80 | cut -c1-170 | grep -i "region\|calls"
