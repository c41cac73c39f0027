set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp
: > $OUT/r04e_flash_ab.txt
for shape in ${AB_SHAPES:-0 1 2 3 4 5}; do
for mode in 1; do
  rm -rf /tmp/prof_ab
  AB_SHAPE=$shape rocprofv3 --kernel-trace -d /tmp/prof_ab -o run -- python $GRAFT_REPO_ROOT/scripts/experiments/ab_flash.py > /tmp/ab_$mode.log 2>&1
  grep "^B=" /tmp/ab_$mode.log >> $OUT/r04e_flash_ab.txt || tail -5 /tmp/ab_$mode.log >> $OUT/r04e_flash_ab.txt
  python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/prof_ab -name "*.db" | head -1) 12 | grep -i -E "flash" >> $OUT/r04e_flash_ab.txt
done
done
cat $OUT/r04e_flash_ab.txt
