#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
python -m pytest tests -m gpu -q -x -k "polic or gaussian or mlp" > $OUT/r5_pytest_gpu_s4.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5_pytest_gpu_s4.log
PB_PREC=f16x2,bf16x3 PB_KINDS=gaussian timeout 600 python tools/pbench.py c5 c3 2>&1 | grep -v amdgpu.ids > $OUT/r5_pbench_f16x2_scaled.log; cat $OUT/r5_pbench_f16x2_scaled.log
timeout 300 python tools/policy_accuracy.py c5 2>&1 | grep -v amdgpu.ids | tail -12 > $OUT/r5_policy_accuracy.log; cat $OUT/r5_policy_accuracy.log
timeout 1500 python tools/abtest.py 3 c5,c5e,c3,c3e abl/base5.so abl/blk54.so abl/sym19.so > $OUT/r5_abtest_lds_pad_dispatch.log 2>&1
tail -7 $OUT/r5_abtest_lds_pad_dispatch.log
