#!/bin/bash
# round 5, first GPU session: full GPU suite, default bench line, returns / advantage, rollout occupancy A/B
mkdir -p gpurun_out; OUT=$(pwd)/gpurun_out
bash tools/gputest.sh; echo "pytest rc=$?"
cp gpurun_out/pytest_gpu.log gpurun_out/r5_pytest_gpu_s1.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r5_c3_bench_steps20_s1.json 2> $OUT/r5_c3_bench_s1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_c3_bench_steps20_s1.json").read().strip().split("\n")[-1])
print("value", d["value"], "us/step", d["ms_per_step"] * 1e3, "frac", d["roofline"]["frac"], d["roofline"]["frac_survey_bytes"])
print("fused", json.dumps(d["fused_rollout"])[:600])
for k, v in d["other_workloads"].items():
    if k == "aux_kernels":
        for kk, vv in v.items():
            print("  aux", kk, vv if "error" in vv else (round(vv["us_per_call"], 2), round(vv["roofline"]["frac"], 3)))
    else:
        print(" ", k, v.get("error") or (round(v["ms_per_step"] * 1e3, 3), round(v["roofline"]["frac"], 3), v.get("policy_kernel_ms"), (v.get("policy_roofline") or {}).get("frac")))
PY
timeout 300 python tools/fbench.py > $OUT/r5_fbench_s1.log 2>&1; cat $OUT/r5_fbench_s1.log | tail -5
DRONESIM_LIB=build/libdronesim_a.so timeout 300 python tools/fbench.py 2>&1 | tail -5
timeout 1500 python tools/abtest.py 3 c2re,c2rr,5x65536:5:1.0re,128x4096:56:1.0re,128x4096:56:1.0rr,200x2048:160:2.0re,c5re,c3re build/libdronesim_a.so scalable_collision_avoidance_rl_amd/libdronesim.so > $OUT/r5_abtest_rollout_epi_waves.log 2>&1
cat $OUT/r5_abtest_rollout_epi_waves.log | tail -8
