#!/bin/bash
# Round 5: everything profiles/ holds for the round, re-measured on the MI355X box:
#   gpurun --timeout 2400 -- 'bash tools/r5_profiles.sh'
# = tools/refresh_profiles.sh r5 + the launch PERIOD of the step kernel from the raw rocprofv3 kernel trace of the bench command
# (tools/trace_period.py -> r5_c3_period.json / r5_c5_period.json) + one rocprofv3 --stats CSV per policy shape.
set -u
TAG=r5
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/refresh_profiles.sh $TAG > $OUT/${TAG}_refresh.log 2>&1
tail -3 $OUT/${TAG}_refresh.log
T="timeout 280"
Q="--no-cpu-baseline --no-other-workloads --no-rccl-probe"
prof() {   # name, command...
    local name=$1; shift
    rm -rf $OUT/prof_$name
    (cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- "$@" > $OUT/prof_$name.log 2>&1)
    cp $(find $OUT/prof_$name -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
}
# (b) begin->begin period of consecutive step launches inside the replays of the profiled bench command
python tools/trace_period.py $OUT/prof_c3_bench --kernel "drone_kernel<2, false, 0, 1, true>" --bench-log $OUT/prof_c3_bench.log --bytes 19976192 > $OUT/${TAG}_c3_period.json
cat $OUT/${TAG}_c3_period.json | head -40
prof c5_bench python $ROOT/bench.py --workload c5 --steps 1000 --warmup 100 $Q
python tools/trace_period.py $OUT/prof_c5_bench --kernel "drone_kernel<2, false, 0, 2, true>" --bench-log $OUT/prof_c5_bench.log --bytes 9968128 > $OUT/${TAG}_c5_period.json
# (c) one CSV per policy shape (exact f32) at the C5 shard and at C3
for kind in gaussian softmax16 critic; do
    for spec in c5 c3; do
        PB_PREC=f32 PB_KINDS=$kind prof ${spec}_policy_f32_$kind python $ROOT/tools/pbench.py $spec
        grep -h "mlp3" $OUT/${TAG}_${spec}_policy_f32_${kind}_kernel_stats.csv | cut -c1-160
    done
done
PB_PREC=f32,bf16x3,f16x2 $T python tools/pbench.py c5 c3 > $OUT/${TAG}_pbench.log 2>&1
# SQ counters of the C5-shard step kernel
bash tools/sq_counters.sh $TAG c5 > $OUT/${TAG}_sq_c5.log 2>&1
# raw traces stay on the box (the merge back is capped at 64 MiB)
find $OUT -name '*kernel_trace.csv' -size +2M -delete; find $OUT -name '*.db' -delete; find $OUT -name '*counter_collection.csv' -size +2M -delete
du -sh $OUT | tail -1
