#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
python -m pytest tests -m gpu -q -x -k "polic or gaussian or mlp" > $OUT/r5_pytest_gpu_s8.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5_pytest_gpu_s8.log
DRONESIM_POLICY_PASSES=2 python -m pytest tests -m gpu -q -x -k "polic or gaussian or mlp" > $OUT/r5_pytest_gpu_s8b.log 2>&1; echo "pytest (2 passes forced) rc=$?"; tail -3 $OUT/r5_pytest_gpu_s8b.log
for p in 1 2 4 1 2 4; do echo "== passes $p"; DRONESIM_POLICY_PASSES=$p PB_PREC=f32 timeout 600 python tools/pbench.py c5 c3 2>&1 | grep -v amdgpu.ids; done > $OUT/r5_pbench_f32_passes.log 2>&1
cat $OUT/r5_pbench_f32_passes.log
