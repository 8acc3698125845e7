#!/usr/bin/env python3
"""Developer micro-benchmark: per-step time of the step kernel under hipGraph replay.

usage: python tools/kbench.py [c2|c3|c5|NxE:G:delta ...]   (env DRONESIM_LIB selects an alternative build)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import drones

PRESETS = {"c2": (5, 1024, 5.0, 1.0), "c3": (64, 4096, 28.0, 1.0), "c5": (256, 512, 256.0, 2.5),
           "c3x8": (64, 32768, 28.0, 1.0), "c2x32": (5, 32768, 5.0, 1.0), "c5x8": (256, 4096, 256.0, 2.5)}


def run(spec, steps=200, reps=20):
    if spec in PRESETS:
        N, E, G, delta = PRESETS[spec]
    else:
        ne, G, delta = spec.split(":")
        N, E = (int(x) for x in ne.split("x"))
        G, delta = float(G), float(delta)
    env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    npool = int(os.environ.get("KB_POOL", steps))
    pool = torch.rand(npool, E, N, 2, device="cuda", generator=g) * 2 - 1
    for s in range(10):
        env.step(pool[s % npool])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for s in range(steps):
            env.step(pool[s % npool])
    graph.replay(); torch.cuda.synchronize()
    env.reset(renew_obstacles=False)
    times = []
    for _ in range(reps):
        env.reset(renew_obstacles=False)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / steps * 1e3)
    us = float(np.median(times))
    byt = 76 * N * E + 13 * E
    print(f"{spec:>16}: N={N} E={E}  {us:8.2f} us/step (min {min(times):.2f})  {N*E/us*1e6:.3e} agent-steps/s  "
          f"{byt/us/1e3:7.1f} GB/s algorithmic ({byt/us/1e3/8000*100:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    for spec in (sys.argv[1:] or ["c2", "c3", "c5", "c3x8"]):
        run(spec)
