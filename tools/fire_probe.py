#!/usr/bin/env python3
"""Developer probe: what does the launch in which every env finishes its episode cost INSIDE a graph replay?
Total time of graphs of 199 / 200 / 201 / 210 steps from a fresh reset (the 200th launch fires the in-kernel reset)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_collision_avoidance_rl_amd import drones
from tools.kbench import PRESETS

spec = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1, auto_reset=True)
g = torch.Generator(device="cuda").manual_seed(0)
pool = torch.rand(200, E, N, 2, device="cuda", generator=g) * 2 - 1
for s in range(10):
    env.step(pool[s])
torch.cuda.synchronize()
res = {}
for L in (150, 199, 200, 201, 210, 400):
    env.reset(renew_obstacles=False)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for s in range(L):
            env.step(pool[s % 200])
    ts = []
    for _ in range(12):
        env.reset(renew_obstacles=False); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    res[L] = float(np.median(ts))
    print(f"{spec} graph of {L:3d} steps: {res[L]:8.1f} us total, {res[L] / L:.3f} us/step", flush=True)
print(f"step 200 (fires): +{res[200] - res[199]:.1f} us; step 201: +{res[201] - res[200]:.1f} us; steps 202-210: {(res[210] - res[201]) / 9:.2f} us each; "
      f"steps 151-199: {(res[199] - res[150]) / 49:.2f} us each")
