#!/bin/bash
# round 5, GPU call 2: residual loads out of the way of the first weight request (gemv / gemv_w8 / skinny), rowss on top
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "rowss or gemv or decode_step or fp8_decode" 2>&1 | tail -15 ) > gpurun_out/r05_t2.log 2>&1
for m in "8 fp8 pub" "4 bf16 pub"; do timeout 300 python scripts/ubench_skinny_stamps.py $m 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_skinny_stamps2.txt
bash scripts/ab_libs_decode_step.sh r05_resfix_ab.txt "bf16:1 bf16:4 bf16:8 fp8:1 fp8:8 fp8:4 fp8:16" spatialrgpt_amd/libsrgpt_hip_tuning.so spatialrgpt_amd/libsrgpt_hip_tuning_rp.so > /dev/null 2>&1
cat gpurun_out/r05_t2.log gpurun_out/r05_skinny_stamps2.txt gpurun_out/r05_resfix_ab.txt
