#!/bin/bash
# round 6, GPU call: the tile-packed weight layout (16-row tiles in MFMA-operand order, 8 + 8 gate / up tiles), TIMING ONLY
# (SRGPT_SKINNY_PACKED_TIMING=1 reads row-major data as if it were packed: same bytes, wrong results) + block-shape knobs
cd $GRAFT_REPO_ROOT; OUT=$PWD/gpurun_out; mkdir -p $OUT
L=spatialrgpt_amd
run() { echo "== $1 batch $4 $5 $6"; env $3 scripts/ubench_decode_mv $2 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -6; }
P="SRGPT_SKINNY_PACKED_TIMING=1"
{
for rep in 1 2; do
  for fmt in fp8 bf16; do
    run old    $L/libsrgpt_hip_tuning_old.so "X=1" 8 $fmt pub
    run packed $L/libsrgpt_hip_tuning.so "$P" 8 $fmt pub
    run packed+w4 $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_WAVES=4" 8 $fmt pub
    run packed+w8 $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_WAVES=8" 8 $fmt pub
    run packed+w4b4 $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_WAVES=4 SRGPT_SKINNY_PK_BLOCKS=4" 8 $fmt pub
    run packed+w8b2 $L/libsrgpt_hip_tuning.so "$P SRGPT_SKINNY_WAVES=8 SRGPT_SKINNY_PK_BLOCKS=2" 8 $fmt pub
  done
done
for b in 2 4 16; do
  run old    $L/libsrgpt_hip_tuning_old.so "X=1" $b fp8 pub
  run packed $L/libsrgpt_hip_tuning.so "$P" $b fp8 pub
done
run old    $L/libsrgpt_hip_tuning_old.so "X=1" 4 bf16 pub
run packed $L/libsrgpt_hip_tuning.so "$P" 4 bf16 pub
} > $OUT/r06_skinny_tilepacked_timing.txt 2>&1
python3 - <<'PY'
import re
rows=[];cur=None
for l in open('gpurun_out/r06_skinny_tilepacked_timing.txt'):
    if l.startswith('=='):
        cur={'name':l.strip()[3:]};rows.append(cur)
    else:
        m=re.match(r'\s+(\S+)\s.*?([\d.]+) us',l)
        if m and cur is not None: cur[m.group(1)]=float(m.group(2))
        elif cur is not None and l.strip(): cur.setdefault('msg', l.strip())
for r in rows:
    print(f"{r['name']:30s} qkv {r.get('qkv+norm',0):6.2f}  o {r.get('o+res',0):6.2f}  gu {r.get('gateup+norm+swiglu',0):6.2f}  down {r.get('down+res',0):6.2f}  lm {r.get('lm_head+norm',0):7.2f}  sum {r.get('sum',0):6.2f} {r.get('msg','')}")
PY
