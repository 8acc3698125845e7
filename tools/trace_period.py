#!/usr/bin/env python3
"""Launch PERIOD of the step kernel from a raw rocprofv3 --kernel-trace of the bench command.

`--stats` averages the begin->end durations of all launches; inside a hipGraph replay what the benchmark's
`ms_per_step` measures is the begin->begin PERIOD of consecutive step launches (duration + the dependent-launch
boundary).  This tool reads the raw trace (`*kernel_trace.csv`), keeps the launches of one kernel, and for every pair of
CONSECUTIVE dispatches of it that belong to the same replay (the next launch begins < `--max-gap-us` after the previous
one ended and no other kernel ran in between) reports the medians of

    period   = begin[i+1] - begin[i]      (what a step costs inside a replay)
    duration = end[i]     - begin[i]      (what --stats averages)
    gap      = begin[i+1] - end[i]        (dependent-launch boundary as the profiler's timestamps see it)

so that `period <= ms_per_step x (1 + profiler overhead)` can be checked against the bench line of the SAME run
(pass it with --bench-log).  Writes one JSON object.

    python tools/trace_period.py <rocprof-output-dir> --kernel "drone_kernel<2, false, 0, 1, true>" \
        [--bench-log prof.log] [--bytes 19976192] > profiles/r5_c3_period.json"""
import argparse
import csv
import glob
import json
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--kernel", default="drone_kernel<2, false, 0, 1, true>")
    ap.add_argument("--max-gap-us", type=float, default=20.0)
    ap.add_argument("--bench-log", default=None, help="stdout of the profiled bench command (its JSON line is attached)")
    ap.add_argument("--bytes", type=float, default=None, help="algorithmic bytes per launch: adds achieved GB/s and frac of 8 TB/s")
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit(f"no *kernel_trace.csv under {a.dir}")
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            for r in rd:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), a.kernel in r["Kernel_Name"]))
    rows.sort()
    b = np.array([r[0] for r in rows], np.int64); e = np.array([r[1] for r in rows], np.int64)
    k = np.array([r[2] for r in rows], bool)
    pair = k[:-1] & k[1:] & ((b[1:] - e[:-1]) < a.max_gap_us * 1e3)          # consecutive launches of the kernel, same replay
    period, dur, gap = (b[1:] - b[:-1])[pair], (e - b)[:-1][pair], (b[1:] - e[:-1])[pair]
    q = lambda x: {"median": float(np.median(x)) / 1e3, "mean": float(np.mean(x)) / 1e3, "p10": float(np.percentile(x, 10)) / 1e3,
                   "p90": float(np.percentile(x, 90)) / 1e3}
    out = {"kernel": a.kernel, "trace_files": [os.path.basename(f) for f in files], "launches_of_the_kernel": int(k.sum()),
           "consecutive_pairs_inside_replays": int(pair.sum()), "unit": "us",
           "period_begin_to_begin": q(period), "duration_begin_to_end": q(dur), "gap_end_to_begin": q(gap),
           "stats_average_duration_all_launches": float(np.mean((e - b)[k])) / 1e3,
           "note": "timestamps of rocprofv3 --kernel-trace; period = what a step costs inside a hipGraph replay under the profiler"}
    if a.bytes:
        for name in ("period_begin_to_begin", "duration_begin_to_end"):
            out[name]["GBps_at_median"] = a.bytes / (out[name]["median"] * 1e-6) / 1e9
            out[name]["frac_of_8TBps_at_median"] = out[name]["GBps_at_median"] / 8000.0
        out["algorithmic_bytes_per_launch"] = a.bytes
    if a.bench_log and os.path.exists(a.bench_log):
        for line in open(a.bench_log):
            if line.startswith("{"):
                d = json.loads(line)
                out["bench_line_of_this_run"] = {"ms_per_step": d["ms_per_step"], "us_per_step": d["ms_per_step"] * 1e3, "value": d["value"],
                                                 "kernel_ms_hip_events": d["roofline"]["kernel_ms"], "frac": d["roofline"]["frac"],
                                                 "frac_survey_bytes": d["roofline"].get("frac_survey_bytes")}
                out["period_over_bench_us_per_step"] = out["period_begin_to_begin"]["median"] / (d["ms_per_step"] * 1e3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
