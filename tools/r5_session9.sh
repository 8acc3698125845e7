#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
python -m pytest tests -m gpu -q -x -k "not launcher and not polic and not mlp and not big_batch" > $OUT/r5_pytest_gpu_s9.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5_pytest_gpu_s9.log
timeout 1500 python tools/abtest.py 3 256x4096:256:2.5e,c5e,130x4096:130:1.0e,c5rr,256x4096:256:2.5rr,600x256:600:2.5e abl/base7.so abl/alias.so > $OUT/r5_abtest_sampling_table_alias.log 2>&1
tail -6 $OUT/r5_abtest_sampling_table_alias.log
