#!/usr/bin/env python3
"""Developer fuzz: dronesim_rollout (T fused steps, candidate lists kept between steps) against T dronesim_step launches,
bit for bit, over random shapes / densities / Delta kinds / action patterns (slow, bursty, stand-still, teleporting).

    python tools/fuzz_rollout.py [iterations] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import drones, formation_O

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
done = 0
for it in range(iters):
    N = int(rng.choice([5, 24, 48, 64, 64, 65, 100, 128, 130, 192, 200, 250, 256, 256, 300]))
    k = int(rng.integers(1, min(N - 1, int(os.environ.get("FUZZ_KMAX", 3))) + 1))
    c = int(rng.choice([2, 2, 2, 5]))
    G = float(rng.choice([0.25, 0.45, 1.0])) * N + 6.0
    d_hat = formation_O(N, [G, G])[1]
    if d_hat.min() <= 0.05:
        continue
    kind = rng.choice(["uniform", "uniform", "hetero", "none"])
    deltas = (np.ones(N) * float(rng.uniform(0.2, 0.95)) * d_hat.min() if kind == "uniform"
              else rng.uniform(0.1, 1.3, N) * d_hat.min() if kind == "hetero" else None)
    E = int(rng.integers(1, 40)) if N <= 130 else int(rng.integers(1, 10))
    T = int(rng.integers(20, 70))
    auto = bool(rng.integers(0, 2))
    kw = dict(auto_reset=True) if auto else {}
    mk = lambda: drones(N, 0, [G, G], "O", k_closest=k, deltas=deltas, simplify_zstate=(c == 2), n_envs=E, batched=True,
                        device="cuda:0", seed=100 + it, **kw)
    a, b = mk(), mk()
    box = float(rng.uniform(0.1, 0.9)) * G
    pos0 = (G / 2 + (rng.random((E, N, 2)) - 0.5) * box).astype(np.float32)
    t0 = rng.integers(150, 199, E).astype(np.int32) if auto else np.zeros(E, np.int32)
    a.set_state(pos0, None, t0); b.set_state(pos0, None, t0)
    g = torch.Generator(device="cuda:0").manual_seed(it)
    act = torch.rand(T, E, N, 2, device="cuda:0", generator=g) * 2 - 1
    act[::5] *= float(rng.uniform(1, 6))
    act[T // 3:T // 3 + 6, ::3] = 0.0
    act[T - 5] *= 30.0
    out = a.rollout(act)
    for s in range(T):
        res = b.step(act[s])
        for name, ref in (("reward", res.rewards), ("true_reward", res.true_rewards), ("z", res.z_states),
                          ("nbr_idx", b.nbr_idx), ("n_coll", res.n_collisions), ("done", res.finished)):
            assert torch.equal(out[name][s], ref), (it, N, k, c, kind, E, T, auto, name, s)
    assert torch.equal(a.pos, b.pos) and torch.equal(a.t, b.t), (it, N)
    done += 1
print(f"fuzz_rollout: {done} configurations, rollout == step launches bit for bit")
