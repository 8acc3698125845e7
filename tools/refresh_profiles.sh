#!/bin/bash
# Re-measure everything profiles/ holds, on the MI355X box:
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r1'
# writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (see profiles/README.md).
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 2000 --warmup 200 > $OUT/${TAG}_c3_bench.json 2> $OUT/${TAG}_c3_bench.err
tail -1 $OUT/${TAG}_c3_bench.json
cd /tmp
prof() {   # name, command...
    local name=$1; shift
    rm -rf $OUT/prof_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- "$@" > $OUT/prof_$name.log 2>&1
    cp $(find $OUT/prof_$name -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
}
(cd $ROOT && prof c3_bench python bench.py --steps 1000 --warmup 100 --no-cpu-baseline)
cp $(find $OUT/prof_c3_bench -name '*domain_stats.csv' | head -1) $OUT/${TAG}_c3_bench_domain_stats.csv 2>/dev/null
(cd $ROOT && prof c3_kbench python tools/kbench.py c3)
(cd $ROOT && prof rollout python tools/rbench.py c3 c5)
(cd $ROOT && prof c3_policy python tools/pbench.py c3)
for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/pmc_$TAG/$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    rm -rf $d
    (cd $ROOT && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python tools/pmc_run.py c3 > $OUT/pmc_${TAG}_$c.log 2>&1)
done
(cd $ROOT/tools && python pmc_parse.py $OUT/pmc_$TAG c3 > $OUT/${TAG}_c3_pmc_traffic.json)
tail -5 $OUT/${TAG}_c3_pmc_traffic.json
(cd $ROOT && python tools/kbench.py c3 c2 c5 256x4096:256:2.5 c3x8 > $OUT/${TAG}_kbench.log 2>&1; python tools/pbench.py c3 c5 > $OUT/${TAG}_pbench.log 2>&1; python tools/rbench.py c3 c5 > $OUT/${TAG}_rbench.log 2>&1)
cat $OUT/${TAG}_kbench.log
