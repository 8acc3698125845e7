#!/usr/bin/env python3
"""Developer micro-benchmark: cost of the episode layer on the step kernel (plain vs track_episodes vs auto_reset),
C3 shape by default, hipGraph of 200 steps, configurations interleaved over several rounds (box clocks drift).

usage: python tools/epibench.py [rounds] [c3|c5|c2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import drones
from tools.kbench import PRESETS


def build(spec, never=False, **kw):
    N, E, G, delta = PRESETS[spec]
    env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1, **kw)
    if never:                                   # the 200-step limit never fires: per-step overhead of auto_reset alone
        env._params().max_steps = 1 << 30
    g = torch.Generator(device="cuda").manual_seed(0)
    pool = torch.rand(200, E, N, 2, device="cuda", generator=g) * 2 - 1
    for s in range(10):
        env.step(pool[s])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for s in range(200):
            env.step(pool[s])
    graph.replay(); torch.cuda.synchronize()
    return env, pool, graph


def measure(env, graph, reps=10):
    ts = []
    for _ in range(reps):
        env.reset(renew_obstacles=False)          # same agent density for every configuration and replay
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 200 * 1e3)
    return float(np.median(ts))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    spec = sys.argv[2] if len(sys.argv) > 2 else "c3"
    cfgs = {"plain": {}, "track_episodes": dict(track_episodes=True), "auto_reset": dict(auto_reset=True),
            "auto_reset_never": dict(auto_reset=True, never=True)}
    built = {k: build(spec, **kw) for k, kw in cfgs.items()}
    res = {k: [] for k in cfgs}
    for _ in range(rounds):
        for k in cfgs:
            res[k].append(measure(built[k][0], built[k][2]))
    for k, v in res.items():
        print(f"{spec} {k:>16}: median {np.median(v):.3f} us/step  (min {min(v):.3f}, rounds {['%.2f' % x for x in v]})", flush=True)


if __name__ == "__main__":
    main()
