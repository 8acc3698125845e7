set -u
TAG=r2; ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp; T="timeout 280"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/pmc_$TAG/$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    rm -rf $d
    (cd $ROOT && $T rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python tools/pmc_run.py c3 > $OUT/pmc_${TAG}_$c.log 2>&1)
done
(cd $ROOT/tools && python pmc_parse.py $OUT/pmc_$TAG c3 > $OUT/${TAG}_c3_pmc_traffic.json)
tail -5 $OUT/${TAG}_c3_pmc_traffic.json
(cd $ROOT && bash tools/sq_counters.sh $TAG c3 > $OUT/${TAG}_sq.log 2>&1)
cat $OUT/${TAG}_sq_c3.csv | head -60
