#!/usr/bin/env python3
"""Developer probe: wall time of the E = 1 drop-in (reference-typed) step(), for comparison with the
reference's own CPU step (0.35 ms at N=5, 17.6 ms at N=64, 193 ms at N=256)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scalable_collision_avoidance_rl_amd import drones
for N, G, d in [(5, 5, 1.0), (64, 28, 1.0), (256, 256, 2.5)]:
    env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * d, simplify_zstate=True)
    rng = np.random.default_rng(0)
    acts = [list(rng.uniform(-1, 1, (N, 2))) for _ in range(200)]
    for a in acts[:20]: env.step(a)
    t0 = time.perf_counter()
    for a in acts: env.step(a)
    dt = (time.perf_counter() - t0) / 200
    print(f"compat step N={N}: {dt*1e3:.3f} ms/step -> {N/dt:.3e} agent-steps/s")
