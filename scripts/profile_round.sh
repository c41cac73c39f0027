#!/bin/bash
# Round profile: kernel trace of the bench command + two separate PMC passes (counters never share a run with trace domains).
# Usage (on the GPU box, from the repo root): bash scripts/profile_round.sh <tag>   -> gpurun_out/<tag>_*.txt
set -u
TAG=${1:-r05_final}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_m
# 1. kernel trace (+ the un-profiled bench line of the same build for reference)
$CMD > $OUT/${TAG}_bench_line.json 2>/dev/null
python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_default.json 2>$OUT/${TAG}_bench_default.err
rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- $CMD > /tmp/kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $DB 40 > $OUT/${TAG}_kernel_stats.txt
python $GRAFT_REPO_ROOT/scripts/prof_prefix.py $DB > $OUT/${TAG}_prefix.txt
# 2. HBM bytes: FETCH_SIZE alone (costs 3 of 4 TCC slots)
SHORT="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 8"
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o run -- $SHORT > /tmp/f.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/prof_f -name "*.db" | head -1) FETCH_SIZE > $OUT/${TAG}_pmc_fetch.txt 2>&1
# 3. MFMA busy cycles vs wall cycles for the MFMA kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_m -o run -- $SHORT > /tmp/m.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/prof_m -name "*.db" | head -1) SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > $OUT/${TAG}_pmc_mfma.txt 2>&1
# 4. LDS bank conflicts of the GEMM kernels (prefill + ViT go through gemm_bf16_glds; the batch-8 line adds gemm256)
rm -rf /tmp/prof_l
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d /tmp/prof_l -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 4 --batch 8 > /tmp/l.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/prof_l -name "*.db" | head -1) SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES > $OUT/${TAG}_pmc_lds.txt 2>&1
# 5. kernel tables of the batched / fp8 configurations (configs[2] and [4] per-GPU shapes)
for cfgname in "b4:--preset config2" "fp8a8b8:--preset config4"; do
  nm=${cfgname%%:*}; ar=${cfgname#*:}
  rm -rf /tmp/prof_c
  rocprofv3 --kernel-trace -d /tmp/prof_c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline $ar > /tmp/c.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_c -name "*.db" | head -1) 16 > $OUT/${TAG}_kernel_stats_${nm}.txt
done
tail -3 /tmp/f.log /tmp/m.log /tmp/l.log > $OUT/${TAG}_pmc_logs.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|FETCH_SIZE|GUI_ACTIVE|SQ_BUSY" | head -30 > $OUT/${TAG}_counters_available.txt
echo done
