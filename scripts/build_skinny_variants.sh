#!/bin/bash
# Tuning builds for the A/B runs of the skinny decode kernel (scripts/gpu_round2_[h-l].sh): the tuning library (environment knobs
# compiled in), optional variants that differ only in skinny.hip's compile-time ring depth / fragment batch, HEAD's skinny.hip
# as "old", and the raw-graph microbenchmark.
#   -> spatialrgpt_amd/libsrgpt_hip_tuning.so, libsrgpt_hip_tuning_{old,...}.so, scripts/ubench_decode_mv
# usage: build_skinny_variants.sh [name:flags ...]   e.g.  d3:-DSRGPT_SKINNY_DEPTH=3 fs8:-DSRGPT_SKINNY_FS=8
set -e
cd "$(dirname "$0")/../spatialrgpt_amd/csrc"
make TUNING=1 -j8 >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DSRGPT_TUNING_KNOBS"
OTHERS=$(ls *.tuning.o | grep -v '^skinny')
build() {  # name, source, extra flags
  /opt/rocm/bin/hipcc $FLAGS $3 -c $2 -o skinny.$1.tuningv.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS skinny.$1.tuningv.o -o ../libsrgpt_hip_tuning_$1.so
}
for v in "$@"; do build "${v%%:*}" skinny.hip "${v#*:}" & done
if git show "${OLD_REV:-HEAD}":spatialrgpt_amd/csrc/skinny.hip > skinny_old_tmp.hip 2>/dev/null; then build old skinny_old_tmp.hip "" & fi
wait
rm -f skinny_old_tmp.hip
cd ../../scripts
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 ubench_decode_mv.hip -o ubench_decode_mv -ldl
ls -la ../spatialrgpt_amd/*.so ubench_decode_mv
