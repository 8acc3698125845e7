#!/usr/bin/env python3
"""What rocprofv3's kernel trace adds to a dispatch inside a hipGraph replay, measured on a kernel of KNOWN duration.

The step launches of the benchmark read 5.1-5.2 us per step by the benchmark's clock and by HIP events, but 5.2-5.5 us per launch in
the rocprofv3 --kernel-trace of the same command (profiles/r5_c3_period.json).  This tool measures the difference on a kernel
that does nothing but stream the step kernel's I/O volume (tools/calib_copy.hip: probe_stream, 1024 x 256 threads, 16 B read + 56 B
written per agent of a 64 x 4096 batch; 3.9-4.0 us per launch by HIP events), in a replay of the same length as the benchmark's
(4000 dependent launches):

    python tools/prof_overhead.py run > plain.log                                           # HIP events, no profiler
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/prof_overhead.py run > prof.log
    python tools/prof_overhead.py combine plain.log DIR prof.log <period.json of the profiled bench command> [bench.json] > profiles/r6_c3_period.json

`combine` reports the stream kernel's begin->begin period under the profiler next to its HIP-event time without it.  Finding (round 6):
the profiler's cost is not an additive term -- the 3.3-us kernel reads 5.1 us per launch in the trace: the kernel trace puts a FLOOR of
about 5.1 us under the dispatch period of a replay, and the step kernel's own trace period (5.2 us) sits just above that floor, within 1.5 %
of the unprofiled benchmark's 5.13 us per step."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHES = 4000


def run():
    import torch
    lib = C.CDLL(os.path.join(ROOT, "tools", "libcalib.so"))
    lib.probe_launch_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    n = 64 * 4096
    a = torch.rand(n, 2, device="cuda"); b = torch.rand(n, 2, device="cuda")
    o = torch.empty(8, n, 2, device="cuda")
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fn = lambda: lib.probe_launch_stream(a.data_ptr(), b.data_ptr(), o.data_ptr(), 1024, 256, 7, 1, st())
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(LAUNCHES):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / LAUNCHES * 1e3)
    print("CALIB_JSON " + json.dumps({"kernel": "probe_stream", "grid": "1024 x 256", "bytes_per_launch": n * (16 + 56),
                                      "launches_per_replay": LAUNCHES, "replays": len(ts),
                                      "hip_event_us_per_launch": {"median": float(np.median(ts)), "p10": float(np.percentile(ts, 10)),
                                                                  "p90": float(np.percentile(ts, 90))}}))


def calib_of(log):
    for line in open(log):
        if line.startswith("CALIB_JSON "):
            return json.loads(line[11:])
    sys.exit(f"no CALIB_JSON line in {log}")


def trace_period(d, kernel):
    import csv
    import glob
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kernel in r["Kernel_Name"]))
    rows.sort()
    b = np.array([r[0] for r in rows], np.int64); e = np.array([r[1] for r in rows], np.int64); k = np.array([r[2] for r in rows], bool)
    pair = k[:-1] & k[1:] & ((b[1:] - e[:-1]) < 20e3)
    per = (b[1:] - b[:-1])[pair]
    return {"pairs": int(pair.sum()), "median": float(np.median(per)) / 1e3, "p10": float(np.percentile(per, 10)) / 1e3,
            "p90": float(np.percentile(per, 90)) / 1e3, "mean": float(np.mean(per)) / 1e3}


def combine(plain_log, prof_dir, prof_log, period_json, bench_json=None):
    plain, under = calib_of(plain_log), calib_of(prof_log)
    per = trace_period(prof_dir, "probe_stream")
    step = json.load(open(period_json))
    out = dict(step)
    out["profiler_overhead_calibration"] = {
        "kernel_of_known_duration": "probe_stream (tools/calib_copy.hip): 1024 x 256 threads, 16 B read + 56 B written per agent of 64 x 4096, "
                                    f"{plain['launches_per_replay']} dependent launches per hipGraph replay",
        "hip_events_no_profiler_us": plain["hip_event_us_per_launch"],
        "hip_events_under_rocprofv3_us": under["hip_event_us_per_launch"],
        "trace_period_under_rocprofv3_us": per}
    restate(out, bench_json)
    print(json.dumps(out, indent=1))


def restate(out, bench_json=None):
    """The reading of the calibration.  The profiler's cost is NOT an additive per-dispatch term: a kernel that takes 3.3 us per launch by
    HIP events shows a 5.1 us begin->begin period in the trace -- rocprofv3's kernel trace puts a FLOOR under the dispatch period of a
    hipGraph replay (its completion-signal handling serialises dispatches), and any kernel shorter than the floor reads as the floor."""
    cal = out["profiler_overhead_calibration"]
    ev0 = cal["hip_events_no_profiler_us"]["median"]
    floor = cal["trace_period_under_rocprofv3_us"]["median"]
    raw = out["period_begin_to_begin"]["median"]
    cal["profiler_dispatch_floor_us"] = floor
    cal["floor_over_true_duration"] = floor / ev0
    cal["reading"] = (f"a {ev0:.2f}-us kernel reads {floor:.2f} us per launch in the trace: the kernel trace puts a floor of ~{floor:.1f} us under the "
                      "dispatch period of a replay on this box (not an additive cost), so the trace cannot resolve a step launch shorter than "
                      "that; the step kernel's own period in the trace sits just above the floor")
    out["period_minus_profiler_floor_us"] = raw - floor
    out.pop("period_corrected_us", None); out.pop("period_corrected_frac_of_8TBps", None); out.pop("period_corrected_over_bench_us_per_step", None)
    cal.pop("per_dispatch_overhead_us", None); cal.pop("how", None)
    if bench_json and os.path.exists(bench_json):
        for line in open(bench_json):
            if line.startswith("{"):
                d = json.loads(line)
                out["bench_line_without_profiler"] = {"us_per_step": d["ms_per_step"] * 1e3, "kernel_us_hip_events": d["roofline"]["kernel_ms"] * 1e3,
                                                      "frac_survey_bytes": d["roofline"].get("frac_survey_bytes")}
    if "bench_line_without_profiler" in out:
        out["period_over_unprofiled_bench_us_per_step"] = raw / out["bench_line_without_profiler"]["us_per_step"]
    return out


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 6 and sys.argv[1] == "combine":
        combine(*sys.argv[2:7])
    elif len(sys.argv) == 3 and sys.argv[1] == "restate":      # re-derive the reading from an existing JSON
        print(json.dumps(restate(json.load(open(sys.argv[2]))), indent=1))
    else:
        sys.exit(__doc__)
