#!/bin/bash
# The check of a build on the GPU box: whole GPU suite (optionally without the 5-minute free-running parity file), smoke(), default bench.
#   gpurun -- 'bash scripts/gpu_suite.sh [tag] [--quick]'   -> gpurun_out/<tag>_tests.txt, <tag>_bench.json
TAG=${1:-suite}; OUT=$PWD/gpurun_out; mkdir -p $OUT
DESEL=""; [ "${2:-}" = "--quick" ] && DESEL="--deselect tests/test_gpu_freerun_parity.py"
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 $DESEL 2>&1 ) | grep -v amdgpu.ids | tail -25 > $OUT/${TAG}_tests.txt; cat $OUT/${TAG}_tests.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
( timeout 900 python bench.py 2>$OUT/${TAG}_bench.err ) | tail -1 > $OUT/${TAG}_bench.json; cat $OUT/${TAG}_bench.json
