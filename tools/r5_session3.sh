#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
# (1) merged candidate walk at the C5 shard and neighbours: correctness first (the product build = merge 2), then A/B
python -m pytest tests -m gpu -q -x -k "not launcher and not big_batch" > $OUT/r5_pytest_gpu_s3.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5_pytest_gpu_s3.log
timeout 1500 python tools/abtest.py 3 c5,c5e,256x4096:256:2.5e,130x1024:130:1.0e,200x512:160:2.0 abl/pol2.so abl/merge0.so abl/merge2.so abl/merge3.so > $OUT/r5_abtest_block_merged_walk.log 2>&1
tail -9 $OUT/r5_abtest_block_merged_walk.log
# (2) policy kernels: round-4 dealing (abl/ra_pin.so) vs tr_plan
for lib in abl/ra_pin.so scalable_collision_avoidance_rl_amd/libdronesim.so; do echo "== $lib"; DRONESIM_LIB=$lib PB_PREC=f32 timeout 600 python tools/pbench.py c5 c3 2>&1 | grep -v amdgpu.ids; done > $OUT/r5_pbench_f32_dealing.log 2>&1
cat $OUT/r5_pbench_f32_dealing.log
