cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/prof_request_timeline.py $DB
