import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scalable_collision_avoidance_rl_amd import drones, formation_O
rng = np.random.default_rng(int(sys.argv[1]))
for it in range(int(sys.argv[2])):
    N = int(rng.choice([2, 3, 5, 7, 16, 31, 32, 33, 48, 63, 64, 65, 100, 128, 200, 300]))
    k = int(rng.integers(1, min(N - 1, 8) + 1)); c = int(rng.choice([2, 2, 5]))
    G = float(max(6.0, 0.45 * N + 2 * rng.random()))
    d_hat = formation_O(N, [G, G])[1]
    if d_hat.min() <= 0.05:
        continue
    mode = rng.choice(["uniform", "hetero", "none"])
    deltas = (np.ones(N) * float(rng.uniform(0.2, 0.95)) * d_hat.min() if mode == "uniform"
              else rng.uniform(0.1, 1.3, N) * d_hat.min() if mode == "hetero" else d_hat.copy())
    E = int(rng.integers(1, 40)) if N > 64 else int(rng.integers(1, 150))
    seed = int(rng.integers(1, 1 << 30))
    t0 = rng.integers(185, 200, E)
    print(f"episode fuzz#{it} N={N} k={k} c={c} G={G:.2f} {mode} E={E}", flush=True)
    try:
        A = drones(N, 0, [G, G], "O", k_closest=k, deltas=deltas, simplify_zstate=(c == 2), n_envs=E, batched=True, device="cuda:0", seed=seed, auto_reset=True)
        A.step(torch.zeros(E, N, 2, device="cuda:0"))
    except Exception as ex:
        print("   FAILED:", str(ex)[:200])
