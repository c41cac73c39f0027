#!/bin/bash
# round 5, GPU call 11: the final tree (product library only, no tuning build in it): quick suite + smoke + default bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ls spatialrgpt_amd/*.so
bash scripts/gpu_suite.sh r05_final_suite --quick
