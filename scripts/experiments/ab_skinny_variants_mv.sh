L=spatialrgpt_amd
for rep in 1 2; do for v in old tuning nx d3 d3nx; do f=$L/libsrgpt_hip_tuning_$v.so; [ $v = tuning ] && f=$L/libsrgpt_hip_tuning.so; for b in 8 4; do echo "== $v batch $b fp8 rep $rep"; scripts/ubench_decode_mv $f $b fp8 2>&1 | grep -v amdgpu.ids | tail -8; done; done; done
