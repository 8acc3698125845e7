#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command only (the profiler's per-launch overhead differs from box to box:
# run it on several boxes and keep every result) -> gpurun_out/<tag>_c3_bench_kernel_stats.csv
TAG=${1:-r3x}
OUT=$(pwd)/gpurun_out; ROOT=$(pwd); export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-workloads --no-rccl-probe"
cd /tmp; rm -rf $OUT/prof_$TAG
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python $ROOT/bench.py --steps 1000 --warmup 100 $Q > $OUT/prof_$TAG.log 2>&1
cp $(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_c3_bench_kernel_stats.csv
cp $(find $OUT/prof_$TAG -name '*domain_stats.csv' | head -1) $OUT/${TAG}_c3_bench_domain_stats.csv
grep "drone_kernel<2, false, 0, 1, true>" $OUT/${TAG}_c3_bench_kernel_stats.csv | cut -d, -f8-11
grep -o '"kernel_ms": [0-9.e-]*' $OUT/prof_$TAG.log | head -1
