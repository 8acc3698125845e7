#!/usr/bin/env python3
"""Developer probe: host-side cost of env.step() in an eager Python loop (no hipGraph)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_collision_avoidance_rl_amd import drones
env = drones(64, 0, [28, 28], "O", deltas=np.ones(64), simplify_zstate=True, n_envs=4096, batched=True, seed=1)
act = torch.rand(4096, 64, 2, device="cuda") * 2 - 1
for _ in range(50): env.step(act)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): env.step(act)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"eager: host issue {1e6*(t1-t0)/2000:.1f} us/step, incl. drain {1e6*(t2-t0)/2000:.1f} us/step")
