set -u
OUT=$(pwd)/gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -4 > $OUT/r3_pytest_gpu.log
cat $OUT/r3_pytest_gpu.log
timeout 280 python bench.py --steps 2000 --warmup 200 > $OUT/r3_c3_bench.json 2> $OUT/r3_c3_bench.err
Q="--no-cpu-baseline --no-other-workloads --no-rccl-probe"
timeout 280 python bench.py --steps 20 --warmup 5 $Q > $OUT/r3_c3_bench_steps20.json 2>> $OUT/r3_c3_bench.err
ROOT=$(pwd)
cd /tmp
rm -rf $OUT/prof_c3_bench
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3_bench -- python $ROOT/bench.py --steps 1000 --warmup 100 $Q > $OUT/prof_c3_bench.log 2>&1
cp $(find $OUT/prof_c3_bench -name '*kernel_stats.csv' | head -1) $OUT/r3_c3_bench_kernel_stats.csv
cp $(find $OUT/prof_c3_bench -name '*domain_stats.csv' | head -1) $OUT/r3_c3_bench_domain_stats.csv
grep "drone_kernel<2, false, 0, 1, true>" $OUT/r3_c3_bench_kernel_stats.csv | cut -c1-200
tail -1 $OUT/prof_c3_bench.log | cut -c1-300
cd $ROOT
python -c "
import json
d=json.loads(open('gpurun_out/r3_c3_bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])
print(json.dumps(d['fused_rollout'])[:900])
"
