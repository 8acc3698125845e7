#!/usr/bin/env python3
"""Which side of the rollout-vs-step fuzz is nondeterministic?  Re-creates fuzz configuration (seed, it) of
tests/test_gpu_fuzz.py::test_rollout_fuzz_against_step_launches and runs the rollout and the step sequence several times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scalable_collision_avoidance_rl_amd import drones, formation_O
seed, target, big = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rng = np.random.default_rng(seed)
shapes = ([300, 320, 400, 512, 600, 1024, 257, 384] if big else [5, 24, 48, 64, 64, 65, 100, 128, 130, 192, 200, 250, 256, 256, 300])
for it in range(target + 1):
    N = int(rng.choice(shapes)); k = int(rng.integers(1, min(N - 1, 8) + 1)); c = int(rng.choice([2, 2, 2, 5]))
    G = float(rng.choice([0.25, 0.45, 1.0])) * N + 6.0
    d_hat = formation_O(N, [G, G])[1]
    if d_hat.min() <= 0.05: continue
    kind = rng.choice(["uniform", "uniform", "hetero", "none"])
    deltas = (np.ones(N) * float(rng.uniform(0.2, 0.95)) * d_hat.min() if kind == "uniform" else rng.uniform(0.1, 1.3, N) * d_hat.min() if kind == "hetero" else None)
    E = int(rng.integers(1, 40)) if N <= 130 else int(rng.integers(1, 10)) if N <= 300 else int(rng.integers(1, 4))
    T = int(rng.integers(20, 70)) if N <= 300 else int(rng.integers(12, 30))
    auto = bool(rng.integers(0, 2))
    box = float(rng.uniform(0.1, 0.9)) * G
    pos0 = (G / 2 + (rng.random((E, N, 2)) - 0.5) * box).astype(np.float32)
    t0 = rng.integers(150, 199, E).astype(np.int32) if auto else np.zeros(E, np.int32)
    mult = float(rng.uniform(1, 6))
print("cfg", dict(N=N, k=k, c=c, G=G, kind=str(kind), E=E, T=T, auto=auto))
kw = dict(auto_reset=True) if auto else {}
g = torch.Generator(device="cuda:0").manual_seed(target)
act = torch.rand(T, E, N, 2, device="cuda:0", generator=g) * 2 - 1
act[::5] *= mult; act[T // 3:T // 3 + 6, ::3] = 0.0; act[T - 5] *= 30.0
def mk():
    e = drones(N, 0, [G, G], "O", k_closest=k, deltas=deltas, simplify_zstate=(c == 2), n_envs=E, batched=True, device="cuda:0", seed=100 + target, **kw)
    e.set_state(pos0, None, t0); return e
def run_roll():
    o = mk().rollout(act); torch.cuda.synchronize(); return {n: o[n].clone() for n in ("z", "nbr_idx", "reward")}
def run_steps():
    e = mk(); zs, ns, rs, ds, ps = [], [], [], [], []
    for s in range(T):
        r = e.step(act[s]); zs.append(r.z_states.clone()); ns.append(e.nbr_idx.clone()); rs.append(r.rewards.clone()); ds.append(r.finished.clone()); ps.append(e.pos.clone())
    torch.cuda.synchronize(); return dict(z=torch.stack(zs), nbr_idx=torch.stack(ns), reward=torch.stack(rs), done=torch.stack(ds), pos=torch.stack(ps))
R = [run_roll() for _ in range(reps)]; S = [run_steps() for _ in range(reps)]
eq = lambda a, b, n: torch.equal(torch.nan_to_num(a[n].float(), nan=7.0), torch.nan_to_num(b[n].float(), nan=7.0))
for n in ("z", "nbr_idx", "reward"):
    print(n, "rollout runs equal to run 0:", [eq(R[0], r, n) for r in R], " step runs equal to run 0:", [eq(S[0], r, n) for r in S], " rollout0 == steps0:", eq(R[0], S[0], n))
for name, runs in (("rollout", R), ("steps", S)):
    for i, r in enumerate(runs[1:], 1):
        if not eq(runs[0], r, "z"):
            d = (torch.nan_to_num(runs[0]["z"], nan=7.0) != torch.nan_to_num(r["z"], nan=7.0)).nonzero()
            print(name, "run", i, "differs from run 0 at", len(d), "entries; first (step, env, agent, col):", d[:6].tolist())
            s_, e_, a_, _ = d[0].tolist()
            print("   z run0", runs[0]["z"][s_, e_, a_].tolist(), "\n   z runi", r["z"][s_, e_, a_].tolist())
            print("   nbr run0", runs[0]["nbr_idx"][s_, e_, a_].tolist(), " runi", r["nbr_idx"][s_, e_, a_].tolist())
            if name == "steps":
                for q in ("z", "nbr_idx", "reward", "pos"):
                    df = (torch.nan_to_num(runs[0][q].float(), nan=7.0) != torch.nan_to_num(r[q].float(), nan=7.0))
                    per = df.reshape(df.shape[0], df.shape[1], -1).sum(-1)
                    print("   ", q, "differing entries per (step, env):", {(int(a), int(b)): int(per[a, b]) for a, b in per.nonzero().tolist()})
                print("    done[step, env] run0:", runs[0]["done"].nonzero().tolist())
            break
