#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the .npz files it
writes are committed and are what travels to the GPU box.  The reference is
imported unmodified from where it lies; two import-time shims make it load on
this image: stub `IPython.display`, and `np.infty` (removed in
NumPy 2).  For the episode fixture `utils.py` additionally needs stub `turtle`
and `autograd` modules (both unused on the path exercised here).

Fixtures hold DATA only (inputs and the reference's outputs):
  formation.npz          generate_formation outputs            drone_env.py:115-153
  single_step_<cfg>.npz  teacher-forced env.step() cases       drone_env.py:214-401
  init_states.npz        seeded ctor/reset states + z/Ni       drone_env.py:171-212
  episode_n5.npz         config C1: one full episode, N=5, seeded softmax-16 policy
                         (rollout loop of train_problem.py:82-107)
"""
import os
import platform
import random
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

np.infty = np.inf  # shim 2
for name in ("IPython", "IPython.display", "turtle", "autograd"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["IPython"].display = sys.modules["IPython.display"]
sys.modules["turtle"].forward = None
sys.modules["autograd"].numpy = np
sys.modules["autograd"].grad = lambda f: f
sys.path.insert(0, "/root/reference")
import contextlib
import io

import drone_env  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))
META = dict(python=platform.python_version(), numpy=np.__version__)


def quiet_env(*a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return drone_env.drones(*a, **kw)


def pack_ni(Ni, K1):
    out = np.full((len(Ni), K1), -1, np.int32)
    for i, lst in enumerate(Ni):
        out[i, :len(lst)] = lst
    return out


def analyse(env, state):
    """margin of every discrete decision + per-row tie information (reference's own matrices)."""
    d, _, nd, _ = env.distance_data(state, env.deltas, env.d_safety)
    N, k = env.n_agents, env.k_closest
    m = np.inf
    off = ~np.eye(N, dtype=bool)
    m = min(m, np.abs(d[off]).min())
    m = min(m, np.abs(d - env.deltas[None, :])[off].min())
    near = (d < env.d_safety[:, None]) & off
    if near.any():
        m = min(m, (env.d_safety[:, None] - d)[near].min())
    tie_free = np.ones(N, bool)
    selection_tied = False
    for i in range(N):
        s = np.sort(d[i])[: k + 2]
        gaps = np.diff(s)
        clipped = s >= env.d_safety[i]
        for a in range(len(gaps)):
            if clipped[a] and clipped[a + 1]:
                if a <= k:
                    tie_free[i] = False      # identity of a selected clipped entry is tie-ordered
            else:
                m = min(m, gaps[a])
        in_range = nd[i].sum() - 1
        # a real neighbour slot filled from a tied (clipped) entry -> reference order undefined
        for kth in range(1, k + 1):
            if kth <= in_range and kth < len(s) and clipped[kth] and (
                    (kth + 1 < len(s) and clipped[kth + 1]) or clipped[kth - 1]):
                selection_tied = True
    xF = env.end_points.reshape(N, 2)
    err = np.linalg.norm(xF - state[:, :2], axis=1)
    m = min(m, np.abs(err - 0.2).min())
    return m, tie_free, selection_tied


def f32_exact(a):
    """Round inputs to float32-representable float64 values: the reference (float64 arithmetic) and the
    float32 HIP path then start from IDENTICAL inputs."""
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


def run_case(env, pos, vel, t, act):
    N, k = env.n_agents, env.k_closest
    pos, vel, act = f32_exact(pos), f32_exact(vel), f32_exact(act)
    env.state[:, 0:2] = pos
    env.state[:, 2:4] = vel
    env.internal_t = int(t)
    state, z, r, ncoll, fin, tr = env.step([a for a in act])
    margin, tie_free, sel_tied = analyse(env, state)
    return dict(pos0=pos, vel0=vel, t0=np.int32(t), act=act, pos1=state[:, 0:2].copy(),
                vel1=state[:, 2:4].copy(), reward=np.asarray(r), true_reward=np.asarray(tr),
                n_coll=np.int64(ncoll), done=np.bool_(fin), z=np.stack(z),
                nbr_idx=pack_ni(env.Ni, k + 1), margin=margin, row_tie_free=tie_free), sel_tied


def gen_single_step(name, N, G, k, simplify, deltas, n_cases, box, rng, extra=(), min_margin=1e-4,
                    centre=None):
    env = quiet_env(N, 0, [G, G], "O", k_closest=k, deltas=deltas, simplify_zstate=simplify)
    cases = []

    def try_add(pos, vel, t, act, force=False):
        case, sel_tied = run_case(env, np.asarray(pos, float), np.asarray(vel, float), t,
                                  np.asarray(act, float))
        if force or (case["margin"] > min_margin and not sel_tied and np.isfinite(case["z"]).all()):
            cases.append(case)
            return True
        return False

    for (pos, vel, t, act) in extra:
        assert try_add(pos, vel, t, act), f"{name}: hand-made case rejected"
    tries = 0
    c0 = np.array([G / 2, G / 2]) if centre is None else np.asarray(centre, float)
    while len(cases) < n_cases + len(extra):
        tries += 1
        assert tries < 200 * n_cases, name
        pos = c0 + (rng.random((N, 2)) - 0.5) * box
        vel = rng.uniform(-1, 1, (N, 2))
        act = rng.uniform(-1, 1, (N, 2))
        t = int(rng.integers(0, 150))
        try_add(pos, vel, t, act)
    keys = cases[0].keys()
    data = {kk: np.stack([c[kk] for c in cases]) for kk in keys}
    data.update(N=N, G=float(G), k=k, c=2 if simplify else 5,
                deltas_arg=np.nan if deltas is None else np.asarray(deltas, float),
                deltas=env.deltas, d_hat=env.d_safety, xF=env.end_points.reshape(N, 2),
                collision_weight=env.collision_weight, **{f"meta_{a}": b for a, b in META.items()})
    np.savez_compressed(os.path.join(OUT, f"single_step_{name}.npz"), **data)
    ncol = int((data["n_coll"] > 0).sum())
    nnb = int((data["nbr_idx"][:, :, 1:] >= 0).sum())
    print(f"single_step_{name}: {len(cases)} cases ({tries} tries), {ncol} with collisions, "
          f"{nnb} real neighbour slots, min margin {data['margin'].min():.2e}")


def gen_formation():
    data = {}
    for (N, G) in [(2, 5), (4, 5), (5, 5), (8, 5), (10, 5), (64, 5), (64, 28), (70, 32), (256, 256)]:
        env = quiet_env(N, 0, [G, G], "O", k_closest=1, deltas=None, simplify_zstate=True)
        data[f"xF_{N}_{G}"] = env.end_points.reshape(N, 2)
        data[f"dhat_{N}_{G}"] = env.d_safety
    env = quiet_env(6, 0, [7, 4], "O", k_closest=1, deltas=None, simplify_zstate=True)   # non-square
    data["xF_6_7x4"] = env.end_points.reshape(6, 2)
    data["dhat_6_7x4"] = env.d_safety
    np.savez_compressed(os.path.join(OUT, "formation.npz"), **data, **{f"meta_{a}": b for a, b in META.items()})
    print("formation:", len(data) // 2, "geometries")


def gen_init_states():
    data = {}
    for (N, G, simp) in [(5, 5, True), (64, 28, True), (5, 5, False)]:
        for s in range(8):
            random.seed(s)
            env = quiet_env(N, 0, [G, G], "O", k_closest=2, deltas=np.ones(N), simplify_zstate=simp)
            margin, tie_free, sel_tied = analyse(env, env.state)
            tag = f"{N}_{G}_{'c2' if simp else 'c5'}_s{s}"
            data[f"state_{tag}"] = env.state.copy()
            data[f"z_{tag}"] = np.stack(env.z_states)
            data[f"nbr_{tag}"] = pack_ni(env.Ni, 3)
            data[f"margin_{tag}"] = margin
            data[f"tiefree_{tag}"] = tie_free
            if s == 0:
                env.step([np.zeros(2)] * N)
                env.reset(renew_obstacles=False)          # drone_env.py:98-102
                assert env.internal_t == 0
                data[f"reset_state_{tag}"] = env.state.copy()
    np.savez_compressed(os.path.join(OUT, "init_states.npz"), **data, **{f"meta_{a}": b for a, b in META.items()})
    print("init_states:", len(data), "arrays")


def gen_episode():
    """Config C1: N=5, E=1, one episode, seeded DiscreteSoftmaxNN(16) actors (train_problem.py:82-107)."""
    import torch
    from SAC_agents import SA2CAgents
    random.seed(7); np.random.seed(7); torch.manual_seed(7)
    N = 5
    env = quiet_env(n_agents=N, n_obstacles=0, grid=[5, 5], end_formation="O", deltas=np.ones(N) * 1.0,
                    simplify_zstate=True)
    env.collision_weight = 0.2
    agents = SA2CAgents(n_agents=N, dim_local_state=env.local_state_space,
                        dim_local_action=env.local_action_space, discount=0.99, epochs=10)
    from utils import ExperienceBuffers
    rec = {kk: [] for kk in ("act", "pos", "vel", "reward", "true_reward", "n_coll", "done", "z", "nbr_idx", "margin",
                             "nbr_idx_pre")}
    state0 = env.state.copy()
    z0 = np.stack(env.z_states); nbr0 = pack_ni(env.Ni, 3)
    buffers = ExperienceBuffers(N)
    finished = False
    while not finished:
        z_states, Ni = env.z_states, env.Ni
        rec["nbr_idx_pre"].append(pack_ni(Ni, 3))
        actions = agents.forward(z_states, Ni)                    # deque of N arrays [2]
        new_state, new_z, r, ncoll, finished, tr = env.step(actions)
        buffers.append(z_states, actions, r, new_z, Ni, finished)  # train_problem.py:96
        m, _, _ = analyse(env, new_state)
        rec["act"].append(np.stack(list(actions))); rec["pos"].append(new_state[:, 0:2].copy())
        rec["vel"].append(new_state[:, 2:4].copy()); rec["reward"].append(r); rec["true_reward"].append(tr)
        rec["n_coll"].append(ncoll); rec["done"].append(finished); rec["z"].append(np.stack(new_z))
        rec["nbr_idx"].append(pack_ni(env.Ni, 3)); rec["margin"].append(m)
    data = {kk: np.stack(v) for kk, v in rec.items()}
    # learner-side quantities of the same episode: Monte-Carlo returns and critic baselines as
    # the reference computes them (SAC_agents.py:359-397), and the neighbour-summed advantage weight of the
    # actor loss, evaluated with the reference's own objects exactly as lines :333-351 do
    Gts, V_approxs = agents.benchmark_cirtic(buffers, only_one_NN=False)
    T = len(rec["act"])
    G = np.stack(list(Gts), axis=1)                     # [T, N]
    V = np.stack(list(V_approxs), axis=1)               # [T, N] float32 critic outputs
    w = np.zeros((T, N))
    for i in range(N):
        for t in range(T):
            Nit = buffers.buffers[i][t].Ni
            adv = 0
            for j in Nit:
                adv += (Gts[j][t] - V[t, i])
            w[t, i] = 1 / N * agents.discount ** t * adv
    data.update(mc_return=G, critic_value=V.astype(np.float64), adv_weight=w, discount=agents.discount)
    data.update(state0=state0, z0=z0, nbr0=nbr0, N=N, G=5.0, k=2, c=2, deltas=env.deltas, d_hat=env.d_safety,
                xF=env.end_points.reshape(N, 2), collision_weight=0.2,
                local_state_space=env.local_state_space, local_action_space=env.local_action_space)
    np.savez_compressed(os.path.join(OUT, "episode_n5.npz"), **data, **{f"meta_{a}": b for a, b in META.items()})
    print(f"episode_n5: {len(rec['act'])} steps, collisions {int(np.sum(rec['n_coll']))}, "
          f"min margin {np.min(rec['margin']):.2e}")


def gen_controllers():
    """gradient_control / proportional_control outputs on random states   drone_env.py:609-679"""
    rng = np.random.default_rng(77)
    data = {}
    for (N, G, box) in [(4, 5, 2.5), (5, 5, 3.0), (64, 28, 9.0), (70, 32, 14.0)]:
        env = quiet_env(N, 0, [G, G], "O", k_closest=1, deltas=np.ones(N), simplify_zstate=True)
        pos, grad, prop, marg = [], [], [], []
        while len(pos) < 8:
            x = f32_exact(G / 2 + (rng.random((N, 2)) - 0.5) * box)
            if len(pos) == 0:
                x = f32_exact(env.end_points.reshape(N, 2) * 0.3 + G * 0.35)      # far from goals -> saturated
            state = np.zeros((N, 5)); state[:, :2] = x; state[:, 4] = 0.1
            # decisions: gap vs d_hat (d_ij <= d_hat_i) and vs 0 (sign flip of the barrier term)
            d = np.linalg.norm(x[:, None] - x[None], axis=-1) - 0.2
            np.fill_diagonal(d, 1e9)
            m = min(np.abs(d - env.d_safety[:, None]).min(), np.abs(d).min())
            if m < 1e-3:
                continue
            pos.append(x); marg.append(m)
            grad.append(np.stack(drone_env.gradient_control(state, env)))
            prop.append(np.stack(drone_env.proportional_control(state, env)))
        data[f"pos_{N}_{G}"] = np.stack(pos); data[f"grad_{N}_{G}"] = np.stack(grad)
        data[f"prop_{N}_{G}"] = np.stack(prop); data[f"margin_{N}_{G}"] = np.array(marg)
    np.savez_compressed(os.path.join(OUT, "controllers.npz"), **data, **{f"meta_{a}": b for a, b in META.items()})
    print("controllers:", len(data) // 4, "geometries x 8 states")


def gen_policies():
    """Forward outputs of the reference's own per-agent networks (utils.py:14-117, 255-302) with their
    weights, for the batched policy kernel.  Weights are stored [in, out] per agent."""
    import torch
    from utils import CriticNN, DiscreteSoftmaxNN, NormalActorNN
    torch.manual_seed(11)
    N, B, d_in = 2, 9, 6
    x = (torch.rand(B, N, d_in) * 6 - 3)
    data = {"x": x.numpy().astype(np.float64)}

    def grab(prefix, mods, layers):
        for li, name in enumerate(layers):
            data[f"{prefix}_w{li}"] = np.stack([getattr(m, name).weight.detach().numpy().T for m in mods])   # float32
            data[f"{prefix}_b{li}"] = np.stack([getattr(m, name).bias.detach().numpy() for m in mods])

    soft = [DiscreteSoftmaxNN(d_in, lr=1e-3, n_actions=16) for _ in range(N)]
    grab("soft", soft, ["input_layer", "hidden_layer1", "out_1"])
    data["soft_out"] = np.stack([np.stack([soft[i].forward(x[b, i]).detach().numpy() for i in range(N)])
                                 for b in range(B)]).astype(np.float64)
    data["soft_action_list"] = soft[0].action_list
    norm = [NormalActorNN(d_in, lr=1e-3, dim_action=2) for _ in range(N)]
    grab("norm", norm, ["input_layer", "hidden_layer1", "hidden_layer2", "out_1", "out_2"])
    data["norm_out"] = np.stack([np.stack([np.concatenate([t.detach().numpy() for t in norm[i].forward(x[b, i])])
                                           for i in range(N)]) for b in range(B)]).astype(np.float64)
    crit = [CriticNN(d_in, output_size=1) for _ in range(N)]
    grab("crit", crit, ["input_layer", "hidden_layer1", "output_layer"])
    data["crit_out"] = np.stack([np.stack([crit[i].forward(x[b, i]).detach().numpy() for i in range(N)])
                                 for b in range(B)]).astype(np.float64)
    np.savez_compressed(os.path.join(OUT, "policies.npz"), **data, **{f"meta_{a}": b for a, b in META.items()})
    print("policies:", {k: v.shape for k, v in data.items() if k.endswith("_out")})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "policies":
        gen_policies()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "controllers":
        gen_controllers()
        return
    rng = np.random.default_rng(20240929)
    gen_formation()
    gen_init_states()

    # hand-checkable KAT of SURVEY.md 8c (N=4, G=5, uniform Delta=1, simplified z)
    kat = ([(1, 1), (2, 1), (1, 2.2), (4, 4)], np.zeros((4, 2)), 0, [(1, 0), (-1, 0), (0, -1), (.5, .5)])
    gen_single_step("n4_c2_uniform", 4, 5, 2, True, np.ones(4), 24, 3.0, rng, extra=[kat])
    # collision KAT: zero actions so step() == rewards() on the injected state; exhibits Q1+Q6
    ckat = ([(1, 1), (1.15, 1), (1, 2.2), (4, 4)], np.zeros((4, 2)), 0, np.zeros((4, 2)))
    gen_single_step("n4_c5_hetero", 4, 5, 2, False, np.array([.3, 2, .3, 2]), 24, 2.5, rng, extra=[ckat])
    gen_single_step("n2_k1_c2", 2, 5, 1, True, np.ones(2), 12, 1.5, rng)
    # C2 geometry + done-logic cases: everyone within 0.2 of goal / t = 198, 199
    envg = quiet_env(5, 0, [5, 5], "O", deltas=np.ones(5), simplify_zstate=True)
    xF5 = envg.end_points.reshape(5, 2)
    z2 = np.zeros((5, 2))
    nz = lambda a: rng.uniform(-a, a, (5, 2))          # break the ring's symmetry (no tied distances)
    mid = xF5 * 0.5 + 1.0
    done_cases = [(xF5 + nz(0.1), z2, 3, z2), (xF5 + nz(0.08), z2, 0, z2 + 0.1),
                  (xF5 + nz(0.05) + np.array([[0.3, 0]] + [[0, 0]] * 4), z2, 10, z2),
                  (mid + nz(0.2), z2, 198, z2 + 0.2), (mid + nz(0.2), z2, 199, z2 - 0.2),
                  (mid + nz(0.2), z2, 250, z2)]
    gen_single_step("n5_c2", 5, 5, 2, True, np.ones(5), 40, 2.5, rng, extra=done_cases)
    gen_single_step("n5_c5_nodelta", 5, 5, 2, False, None, 16, 2.2, rng)
    gen_single_step("n8_c5_hetero_k3", 8, 8, 3, False, rng.uniform(0.3, 2.5, 8), 24, 4.0, rng)
    gen_single_step("n64_c2", 64, 28, 2, True, np.ones(64), 12, 26.0, rng)
    gen_single_step("n64_c5_dense_k4", 64, 28, 4, False, np.ones(64) * 0.8, 12, 7.0, rng)
    gen_single_step("n70_c2", 70, 32, 2, True, np.ones(70), 6, 14.0, rng)
    gen_single_step("n256_c2", 256, 256, 2, True, np.ones(256) * 2.5, 3, 60.0, rng)
    gen_episode()
    gen_controllers()
    gen_policies()


if __name__ == "__main__":
    main()
