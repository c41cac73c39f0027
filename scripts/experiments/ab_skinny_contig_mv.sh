L=spatialrgpt_amd
for rep in 1 2; do for v in tuning contig; do f=$L/libsrgpt_hip_tuning_$v.so; [ $v = tuning ] && f=$L/libsrgpt_hip_tuning.so; for c in "8 fp8" "4 bf16" "8 bf16"; do echo "== $v batch $c rep $rep"; scripts/ubench_decode_mv $f $c 2>&1 | grep -v amdgpu.ids | tail -7; done; done; done
