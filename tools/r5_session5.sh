#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
python -m pytest tests -m gpu -q -x -k "not launcher and not polic and not mlp" > $OUT/r5_pytest_gpu_s5.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5_pytest_gpu_s5.log
timeout 1500 python tools/abtest.py 3 c5,c5e,256x4096:256:2.5e,130x1024:130:1.0e,c5r,c5rr,300x256:300:2.5e abl/cells64.so abl/cells128.so abl/cells256.so > $OUT/r5_abtest_block_cells.log 2>&1
tail -8 $OUT/r5_abtest_block_cells.log
