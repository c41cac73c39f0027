set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py -q -x -k "region or pool or mask" > $OUT/r04d_region_tests.txt 2>&1
tail -3 $OUT/r04d_region_tests.txt
cd /tmp
: > $OUT/r04d_region_ab.txt
for shape in ${AB_SHAPES:-0 1 2 3 4}; do
for mode in ${AB_MODES:-0 1 2 4}; do
  rm -rf /tmp/prof_ab
  AB_SHAPE=$shape SRGPT_REGION_MFMA=$mode rocprofv3 --kernel-trace -d /tmp/prof_ab -o run -- python $GRAFT_REPO_ROOT/scripts/experiments/ab_region_pool.py > /tmp/ab_$mode.log 2>&1
  echo "=== SRGPT_REGION_MFMA=$mode" >> $OUT/r04d_region_ab.txt
  grep "^mode" /tmp/ab_$mode.log >> $OUT/r04d_region_ab.txt
  python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_ab -name "*.db" | head -1) 12 | grep -i -E "region_pool" >> $OUT/r04d_region_ab.txt
done
done
cat $OUT/r04d_region_ab.txt
