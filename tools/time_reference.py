#!/usr/bin/env python3
"""Time the REFERENCE's own `drones.step()` (drone_env.py:214-258) in the build container.

The reference is pure Python and never ships to the GPU box, so this is the one place its own speed
can be measured.  It is imported unmodified from /root/reference
with the two import-time shims of SURVEY.md 8c (stub `IPython.display`, `np.infty`).  For every
(N, grid, Delta) of the BASELINE configs:

  * 1 core:     one env, U(-1,1)^2 actions (RandomAgent, SAC_agents.py:22), 200-step episodes,
                `time.perf_counter` around `step()` only;
  * whole host: one independent env per core (multiprocessing), each timed the same way over the same
                wall-clock window, agent-steps/s summed.

Writes profiles/reference_cpu.json; bench.py attaches that record as `cpu_baseline.reference`
beside the figure of the C port it times live on the GPU host.

    python tools/time_reference.py [--budget 20] [--procs 8]
"""
import argparse
import contextlib
import io
import json
import multiprocessing as mp
import os
import platform
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {   # label: (N, G, Delta)   SURVEY.md 8d
    "c2": (5, 5.0, 1.0),
    "c3": (64, 28.0, 1.0),
    "c5": (256, 256.0, 2.5),
}


def import_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    import numpy as np
    np.infty = np.inf
    for name in ("IPython", "IPython.display"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["IPython"].display = sys.modules["IPython.display"]
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    import drone_env
    return drone_env


def time_one(shape, budget_s, seed, start_at=None):
    """(steps, seconds inside step(), agent-steps/s) of one reference env on the calling core."""
    import numpy as np
    drone_env = import_reference()
    N, G, delta = SHAPES[shape]
    with contextlib.redirect_stdout(io.StringIO()):
        env = drone_env.drones(N, 0, [G, G], "O", k_closest=2, deltas=np.ones(N) * delta, simplify_zstate=True)
    rng = np.random.default_rng(seed)
    if start_at is not None:                       # whole-host leg: all processes time the same window
        while time.time() < start_at:
            time.sleep(0.001)
    steps, inside, t_begin = 0, 0.0, time.perf_counter()
    while time.perf_counter() - t_begin < budget_s:
        act = [rng.uniform(-1, 1, 2) for _ in range(N)]
        t0 = time.perf_counter()
        _, _, _, _, finished, _ = env.step(act)
        inside += time.perf_counter() - t0
        steps += 1
        if finished:                               # train_problem.py:132
            with contextlib.redirect_stdout(io.StringIO()):
                env.reset(renew_obstacles=False)
    return steps, inside, N * steps / inside


def _worker(args):
    return time_one(*args)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=float, default=20.0, help="seconds of step() timing per shape and leg")
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu.json"))
    args = ap.parse_args()
    import numpy as np
    rec = {"what": "reference drones.step() (/root/reference/drone_env.py:214-258), unmodified, imported with the "
                   "IPython / np.infty shims; U(-1,1)^2 actions; perf_counter around step() only",
           "host": {"cpu": cpu_model(), "logical_cpus": os.cpu_count(), "usable_cpus": len(os.sched_getaffinity(0)),
                    "python": platform.python_version(), "numpy": np.__version__,
                    "where": "build container (the reference does not travel to the GPU box)"},
           "budget_s": args.budget, "shapes": {}}
    for shape, (N, G, delta) in SHAPES.items():
        steps, inside, rate1 = time_one(shape, args.budget, 0)
        with mp.get_context("fork").Pool(args.procs) as pool:
            start_at = time.time() + 3.0 + (10.0 if N >= 256 else 0.0)   # the ctor at G=256 takes seconds
            res = pool.map(_worker, [(shape, args.budget, 100 + p, start_at) for p in range(args.procs)])
        rate_all = sum(r[2] for r in res)
        rec["shapes"][shape] = {
            "n_agents": N, "grid": G, "delta": delta,
            "one_core": {"cores": 1, "steps": steps, "ms_per_step": inside / steps * 1e3, "agent_steps_per_s": rate1},
            "whole_host": {"cores": args.procs, "processes": args.procs, "steps": [r[0] for r in res],
                           "ms_per_step_mean": float(np.mean([r[1] / r[0] for r in res]) * 1e3),
                           "agent_steps_per_s": rate_all},
        }
        print(shape, json.dumps(rec["shapes"][shape]), flush=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
