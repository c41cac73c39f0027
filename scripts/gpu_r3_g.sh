#!/bin/bash
# round 3, GPU call g: region pooling v3 (16-byte write-through partials, parallel denominators): tests + kernel trace; decode default check
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -x -q -k "region or raw_uint8 or composition or rope_append" 2>&1 ) | grep -v amdgpu.ids | tail -6 > $OUT/r03g_tests.txt; cat $OUT/r03g_tests.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --max-new-tokens 8 > /tmp/x.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 60 | cut -c1-170 > $OUT/r03g_kernels.txt
grep -i "region\|decode_\|calls" $OUT/r03g_kernels.txt
cd $GRAFT_REPO_ROOT && ( timeout 600 python bench.py --no-cpu-baseline 2>/dev/null ) | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['decode_ms_per_step'], d['roofline']['decode_frac_whole_step'], d['roofline']['frac'])"
