#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -x -q -k "region or raw_uint8 or pipeline or stage or golden" 2>&1 ) | grep -v amdgpu.ids | tail -5
python scripts/ubench_region_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r03_region_stamps_v5.txt
