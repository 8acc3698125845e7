#!/usr/bin/env python3
"""Developer micro-benchmark of dronesim_rollout (T fused steps per launch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import drones
from tools.kbench import PRESETS

for spec in (sys.argv[1:] or ["c3"]):
    N, E, G, delta = PRESETS[spec]
    for T in (8, 50, 200):
        env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
        g = torch.Generator(device="cuda").manual_seed(0)
        act = torch.rand(T, E, N, 2, device="cuda", generator=g) * 2 - 1
        out = env.rollout(act); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); out = env.rollout(act); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / T)
        us = min(ts)
        byt = 52 * N * E
        print(f"{spec} rollout T={T:4d}: {us:7.2f} us/step  {N*E/us*1e6:.3e} agent-steps/s  {byt/us/1e3:7.1f} GB/s (52 B/agent-step)", flush=True)
    # the same T = 200 steps with the actions drawn in the kernel (dronesim_rollout_random: 44 B/agent-step, no pool),
    # plain and with the episode layer (records + in-kernel reset: the rollout runs across episode ends)
    for label, kw in (("random actions", {}), ("random + records + auto-reset", dict(auto_reset=True))):
        T = 200
        env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1, **kw)
        out = env.rollout_random(T); torch.cuda.synchronize(); del out
        ts = []
        for _ in range(5):
            env.reset(renew_obstacles=False)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); out = env.rollout_random(T); b.record(); torch.cuda.synchronize(); del out
            ts.append(a.elapsed_time(b) * 1e3 / T)
        us = min(ts)
        print(f"{spec} rollout T={T:4d}, {label}: {us:7.2f} us/step  {N*E/us*1e6:.3e} agent-steps/s  {44*N*E/us/1e3:7.1f} GB/s (44 B/agent-step)", flush=True)
    # pool actions WITH the episode layer (dronesim_rollout_ex on an auto_reset env: what bench.py's fused_rollout times)
    T = 200
    env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1, auto_reset=True)
    g = torch.Generator(device="cuda").manual_seed(0)
    act = torch.rand(T, E, N, 2, device="cuda", generator=g) * 2 - 1
    out = env.rollout(act); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = env.rollout(act); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / T)
    us = min(ts)
    print(f"{spec} rollout T={T:4d}, pool actions + records + auto-reset: {us:7.2f} us/step  {N*E/us*1e6:.3e} agent-steps/s  {52*N*E/us/1e3:7.1f} GB/s (52 B/agent-step)", flush=True)
