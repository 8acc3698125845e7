#!/bin/bash
# Round 6: everything profiles/r6_* holds, re-measured on the MI355X box:
#   gpurun --timeout 2700 -- 'bash tools/r6_profiles.sh'
# = tools/refresh_profiles.sh r6 (bench lines, rocprofv3 --kernel-trace --stats summaries, PMC traffic, SQ counters, tool logs)
#   + the launch PERIOD of the step kernel from the raw kernel trace of the bench command, raw and with the profiler's per-dispatch
#     cost -- measured in the same call on a kernel of known duration (tools/prof_overhead.py) -- taken off (r6_c3_period.json)
#   + one rocprofv3 --stats CSV per policy shape, SQ counters of the C5-shard step kernel.
set -u
TAG=r6
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
# (a) period of consecutive step launches inside the replays of the profiled bench command (refresh_profiles.sh left prof_c3_bench)
python tools/trace_period.py $OUT/prof_c3_bench --kernel "drone_kernel<2, false, 0, 1, true>" --bench-log $OUT/prof_c3_bench.log --bytes 19976192 > $OUT/${TAG}_c3_period_raw.json
# (b) the profiler's per-dispatch cost on a kernel of known duration, same box, same call
$T python tools/prof_overhead.py run > $OUT/${TAG}_calib_plain.log 2>/dev/null
rm -rf $OUT/prof_calib
(cd /tmp && $T rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_calib -- python $ROOT/tools/prof_overhead.py run > $OUT/${TAG}_calib_prof.log 2>/dev/null)
python tools/prof_overhead.py combine $OUT/${TAG}_calib_plain.log $OUT/prof_calib $OUT/${TAG}_calib_prof.log $OUT/${TAG}_c3_period_raw.json $OUT/${TAG}_c3_bench.json > $OUT/${TAG}_c3_period.json
grep -E "period_corrected|per_dispatch_overhead|us_per_step" $OUT/${TAG}_c3_period.json
prof c5_bench python $ROOT/bench.py --workload c5 --steps 1000 --warmup 100 $Q
python tools/trace_period.py $OUT/prof_c5_bench --kernel "drone_kernel<2, false, 0, 2, true>" --bench-log $OUT/prof_c5_bench.log --bytes 9968128 > $OUT/${TAG}_c5_period.json
# (c) one CSV per policy shape (exact f32) at the C5 shard and at C3
for kind in gaussian softmax16 critic; do
    for spec in c5 c3; do
        PB_PREC=f32 PB_KINDS=$kind prof ${spec}_policy_f32_$kind python $ROOT/tools/pbench.py $spec
        grep -h "mlp3" $OUT/${TAG}_${spec}_policy_f32_${kind}_kernel_stats.csv | cut -c1-160
    done
done
# the float16 row-tile policy kernel (nout <= 4): stats CSV, split kernel vs row-tile same box, phase stamps, its own bench line
PB_PREC=f16x2 PB_KINDS=gaussian prof c5_policy_f16x2_gaussian python $ROOT/tools/pbench.py c5
(for sp in 1 "" 1 ""; do echo "# PB_SPLIT=$sp (1 = split kernel of rounds 2-5, empty = row-tile kernel)"; PB_SPLIT=$sp PB_PREC=f16x2 PB_KINDS=gaussian,critic,softmax16 $T python tools/pbench.py c5 c3 2>&1 | grep -v amdgpu.ids; done) > $OUT/${TAG}_pbench_f16x2_rowtile.log
[ -f abl/libdronesim_trace.so ] && (for k in gaussian critic; do DRONESIM_LIB=abl/libdronesim_trace.so $T python tools/trace_rt16.py $k c5 2>&1 | grep -v amdgpu.ids; done) > $OUT/${TAG}_trace_rt16.log
$T python bench.py --workload c5 --policy gaussian --policy-precision f16x2 > $OUT/${TAG}_c5_gaussian_f16x2_bench.json 2>/dev/null
# SQ counters of the C5-shard step kernel
bash tools/sq_counters.sh $TAG c5 > $OUT/${TAG}_sq_c5.log 2>&1
# raw traces stay on the box (the merge back is capped at 64 MiB)
find $OUT -name '*kernel_trace.csv' -size +2M -delete; find $OUT -name '*.db' -delete; find $OUT -name '*counter_collection.csv' -size +2M -delete
du -sh $OUT | tail -1
