#!/bin/bash
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
python -m pytest tests -m gpu -q -x -k "not launcher" > $OUT/r5_pytest_gpu_s6.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r5_pytest_gpu_s6.log
timeout 1500 python tools/abtest.py 3 c5,c5e,256x4096:256:2.5e,130x1024:130:1.0e,200x512:160:2.0 abl/cells_final.so abl/cm2.so abl/cm3.so > $OUT/r5_abtest_block_cells_merged.log 2>&1
tail -8 $OUT/r5_abtest_block_cells_merged.log
