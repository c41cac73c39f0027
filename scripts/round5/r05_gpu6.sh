#!/bin/bash
# round 5, GPU call 6: what bounds the fp8 K loop of the batched decode product?  timing probes (WRONG results by construction):
# p1 = no fp8 -> bf16 conversions, p2 = weights never pass through LDS, p3 = no matrix instructions; + kernel tables of the current build
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
{
for rep in 1 2; do for v in tuning p1 p2 p3; do f=$L/libsrgpt_hip_tuning_$v.so; [ $v = tuning ] && f=$L/libsrgpt_hip_tuning.so
  for b in 8; do echo "== $v batch $b fp8 rep $rep"; scripts/ubench_decode_mv $f $b fp8 2>&1 | grep -v amdgpu.ids | tail -6; done; done; done
for v in tuning p2 p3; do f=$L/libsrgpt_hip_tuning_$v.so; [ $v = tuning ] && f=$L/libsrgpt_hip_tuning.so
  echo "== $v batch 8 bf16"; scripts/ubench_decode_mv $f 8 bf16 2>&1 | grep -v amdgpu.ids | tail -6; done
} > $OUT/r05_skinny_probes.txt 2>&1
bash scripts/ab_libs_decode_step.sh r05_skinny_probes_step.txt "fp8:8 bf16:8" $L/libsrgpt_hip_tuning.so $L/libsrgpt_hip_tuning_p1.so $L/libsrgpt_hip_tuning_p2.so $L/libsrgpt_hip_tuning_p3.so > /dev/null 2>&1
export TMPDIR=/tmp; cd /tmp
for cfgname in "bs1:" "fp8a8b8:--preset config4"; do
  nm=${cfgname%%:*}; ar=${cfgname#*:}
  rm -rf /tmp/prof_c
  rocprofv3 --kernel-trace -d /tmp/prof_c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline $ar > /tmp/c.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_c -name "*.db" | head -1) 14 > $OUT/r05_mid_kernel_stats_${nm}.txt
done
cd $GRAFT_REPO_ROOT
cat $OUT/r05_skinny_probes.txt $OUT/r05_skinny_probes_step.txt; head -20 $OUT/r05_mid_kernel_stats_bs1.txt | cut -c1-150; head -12 $OUT/r05_mid_kernel_stats_fp8a8b8.txt | cut -c1-150
