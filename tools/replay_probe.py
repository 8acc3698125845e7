#!/usr/bin/env python3
"""Developer probe: what does a hipGraph replay cost per step as a function of the launches it holds?
Graphs of n = 250 ... 8000 C3 step launches (episode layer on, as in bench.py), replayed back to back for ~0.3 s:
wall clock per step and HIP-event time per step.   usage: python tools/replay_probe.py [n ...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import drones

N, E, G = 64, 4096, 28.0


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [250, 500, 1000, 2000, 4000, 8000]
    env = drones(N, 0, [G, G], "O", k_closest=2, deltas=np.ones(N), simplify_zstate=True, n_envs=E, seed=1,
                 batched=True, auto_reset=True, track_episodes=True)
    g = torch.Generator(device="cuda").manual_seed(0)
    pool = torch.rand(200, E, N, 2, device="cuda", generator=g) * 2 - 1
    for s in range(10):
        env.step(pool[s])
    torch.cuda.synchronize()
    for n in sizes:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for s in range(n):
                env.step(pool[s % 200])
        graph.replay(); graph.replay(); torch.cuda.synchronize()
        reps = max(2, int(0.3 / (n * 5.7e-6)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        t_host = time.perf_counter() - t0                  # host time to enqueue all replays
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ev = e0.elapsed_time(e1) * 1e-3
        print(f"n={n:5d} x {reps:3d} replays: wall {wall / (n * reps) * 1e6:6.3f} us/step, events {ev / (n * reps) * 1e6:6.3f} us/step, "
              f"host enqueue {t_host / reps * 1e6:8.1f} us per replay ({t_host / (n * reps) * 1e6:5.3f} us/launch)", flush=True)
        del graph


if __name__ == "__main__":
    main()
