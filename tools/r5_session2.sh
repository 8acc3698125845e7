#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
bash tools/gputest.sh; echo "pytest rc=$?"
cp gpurun_out/pytest_gpu.log gpurun_out/r5_pytest_gpu_s2.log
for lib in abl/ret1.so scalable_collision_avoidance_rl_amd/libdronesim.so; do echo "== $lib"; DRONESIM_LIB=$lib timeout 300 python tools/retbench.py 2>&1 | grep -v amdgpu.ids; done > $OUT/r5_retbench.log 2>&1
cat $OUT/r5_retbench.log
timeout 1500 python tools/abtest.py 3 c3re,c3rr,c5re,c5rr,c3r,64x32768:28:1.0re abl/ret1.so abl/ra.so abl/ra_pin.so > $OUT/r5_abtest_rollout_action_source.log 2>&1
tail -8 $OUT/r5_abtest_rollout_action_source.log
timeout 1500 python tools/abtest.py 3 c2re,c2rr,5x65536:5:1.0re,128x4096:56:1.0re,128x4096:56:1.0rr,200x2048:160:2.0re abl/waves4.so abl/ra_pin.so > $OUT/r5_abtest_rollout_epi_waves.log 2>&1
tail -6 $OUT/r5_abtest_rollout_epi_waves.log
