# kernel table of the per-request prefix (both ViT passes, pooling, projector, splice, prefill) from one traced bench request
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; mkdir -p $OUT; cd /tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /tmp/kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/prof_prefix.py $DB > $OUT/${TAG:-prefix}.txt
head -${LINES:-30} $OUT/${TAG:-prefix}.txt | cut -c1-160
