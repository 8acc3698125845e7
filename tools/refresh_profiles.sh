#!/bin/bash
# Re-measure everything profiles/ holds, on the MI355X box:
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r3'
# writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (see profiles/README.md).
set -u
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout 280"
[ -f tools/libcalib.so ] || make -C tools libcalib.so   # PMC calibration copies (pmc_run.py)
$T python bench.py --steps 2000 --warmup 200 > $OUT/${TAG}_c3_bench.json 2> $OUT/${TAG}_c3_bench.err
tail -1 $OUT/${TAG}_c3_bench.json | cut -c1-400
Q="--no-cpu-baseline --no-other-workloads --no-rccl-probe"
$T python bench.py --steps 20 --warmup 5 $Q > $OUT/${TAG}_c3_bench_steps20.json 2>> $OUT/${TAG}_c3_bench.err
$T python bench.py --steps 2000 --warmup 200 $Q --no-episode-layer > $OUT/${TAG}_c3_bench_plain.json 2>> $OUT/${TAG}_c3_bench.err
$T python bench.py --steps 200 --warmup 20 $Q --scaling strong > $OUT/${TAG}_c4_one_gpu_bench.json 2>> $OUT/${TAG}_c3_bench.err
$T python bench.py --workload c5 --steps 1000 --warmup 100 $Q > $OUT/${TAG}_c5_bench.json 2>> $OUT/${TAG}_c3_bench.err
$T python bench.py --workload c5 --policy gaussian --steps 400 --warmup 20 $Q > $OUT/${TAG}_c5_gaussian_f32_bench.json 2>> $OUT/${TAG}_c3_bench.err
$T python bench.py --workload c5 --policy gaussian --policy-precision bf16x3 --steps 400 --warmup 20 $Q > $OUT/${TAG}_c5_gaussian_bf16x3_bench.json 2>> $OUT/${TAG}_c3_bench.err
$T python bench.py --workload c2 --steps 2000 --warmup 200 $Q > $OUT/${TAG}_c2_bench.json 2>> $OUT/${TAG}_c3_bench.err
cd /tmp
prof() {   # name, command...
    local name=$1; shift
    rm -rf $OUT/prof_$name
    $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- "$@" > $OUT/prof_$name.log 2>&1
    cp $(find $OUT/prof_$name -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
}
(cd $ROOT && prof c3_bench python bench.py --steps 1000 --warmup 100 $Q)
cp $(find $OUT/prof_c3_bench -name '*domain_stats.csv' | head -1) $OUT/${TAG}_c3_bench_domain_stats.csv 2>/dev/null
(cd $ROOT && prof c3_kbench python tools/kbench.py c3)
(cd $ROOT && prof rollout python tools/rbench.py c3 c5)
(cd $ROOT && PB_PREC=f32,bf16x3 prof c5_policy python tools/pbench.py c5)
(cd $ROOT && prof far_kbench python tools/abtest.py --one c3f,c5f)
(cd $ROOT && prof c5_kbench python tools/kbench.py c5 c5x8)
(cd $ROOT && prof fbench python tools/fbench.py)
(cd $ROOT && prof reset_probe python tools/reset_probe.py c3)
for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/pmc_$TAG/$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    rm -rf $d
    (cd $ROOT && $T rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python tools/pmc_run.py c3 > $OUT/pmc_${TAG}_$c.log 2>&1)
done
(cd $ROOT/tools && python pmc_parse.py $OUT/pmc_$TAG c3 > $OUT/${TAG}_c3_pmc_traffic.json)
tail -5 $OUT/${TAG}_c3_pmc_traffic.json
(cd $ROOT && bash tools/sq_counters.sh $TAG c3 > $OUT/${TAG}_sq.log 2>&1)
(cd $ROOT && $T python tools/kbench.py c3 c2 c5 256x4096:256:2.5 c3x8 > $OUT/${TAG}_kbench.log 2>&1; $T python tools/kbench.py 64x512:28:1.0 64x1024:28:1.0 64x2048:28:1.0 64x4096:28:1.0 64x8192:28:1.0 64x16384:28:1.0 > $OUT/${TAG}_esweep.log 2>&1; $T python tools/probe_floor.py > $OUT/${TAG}_probe_floor.log 2>&1; PB_PREC=f32,bf16x3,f16x2,bf16 $T python tools/pbench.py c3 c5 > $OUT/${TAG}_pbench.log 2>&1; $T python tools/rbench.py c3 c5 c5x8 > $OUT/${TAG}_rbench.log 2>&1; $T python tools/epibench.py 5 c3 > $OUT/${TAG}_epibench.log 2>&1; $T python tools/reset_probe.py c3 c5 c2 > $OUT/${TAG}_reset_probe.log 2>&1; $T python tools/fbench.py > $OUT/${TAG}_fbench.log 2>&1)
cat $OUT/${TAG}_kbench.log
