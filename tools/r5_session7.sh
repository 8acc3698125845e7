#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
python -m pytest tests -m gpu -q -x -k "polic or gaussian or mlp" > $OUT/r5_pytest_gpu_s7.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5_pytest_gpu_s7.log
for lib in abl/base6.so scalable_collision_avoidance_rl_amd/libdronesim.so abl/base6.so scalable_collision_avoidance_rl_amd/libdronesim.so; do echo "== $lib"; DRONESIM_LIB=$lib PB_PREC=f32,f16x2 timeout 600 python tools/pbench.py c5 2>&1 | grep -v amdgpu.ids; done > $OUT/r5_pbench_finish_diet.log 2>&1
cat $OUT/r5_pbench_finish_diet.log
