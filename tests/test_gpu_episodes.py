"""GPU tests of the episode layer of the hot path (round 2): per-env episode records accumulated by the step
kernel itself (train_problem.py:98-100, 118-121), in-kernel auto-reset (drone_env.py:98-102 via
train_problem.py:132), in-kernel RandomAgent actions (SAC_agents.py:9-22), checkpoint / resume.

Everything goes through the C ABI (dronesim_step_ex / dronesim_rollout_ex / dronesim_rollout_random /
dronesim_reset_ex / dronesim_episode_reduce) via the `drones` host class."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle.oracle import Oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch


def make_env(N, G, E, k=2, c=2, deltas=None, **kw):
    from scalable_collision_avoidance_rl_amd import drones
    deltas = np.ones(N) if deltas is None else deltas
    return drones(N, 0, [G, G], "O", k_closest=k, deltas=deltas, simplify_zstate=(c == 2),
                  n_envs=E, batched=True, device="cuda:0", seed=kw.pop("seed", 11), **kw)


def host(t):
    return t.detach().cpu().numpy()


def records(env):
    return {k: host(v).copy() for k, v in env.episode_stats().items()}


# ------------------------------------------------------------------------------- per-step accumulation
@pytest.mark.parametrize("N,G,E,T", [(64, 28.0, 512, 200), (5, 5.0, 1000, 200), (130, 130.0, 24, 60), (48, 24.0, 100, 80)])
def test_episode_records_accumulate_every_step(torch, N, G, E, T):
    """ep_return / ep_true_return / ep_collisions / ep_len after T steps equal the sums of the per-step outputs
    (train_problem.py:98-100 adds np.mean(rewards), np.mean(true_rewards), n_collisions on EVERY step), and agree
    with the float64 oracle stepped on the same states and actions (teacher-forced, envs that stay margin-safe)."""
    env = make_env(N, G, E, track_episodes=True, seed=5)
    orc = Oracle(N, [G, G], 2, np.ones(N), True, threads=8)
    g = torch.Generator(device="cuda:0").manual_seed(3)
    own = np.zeros((3, E))
    ref = np.zeros((3, E))
    safe_all = np.ones(E, bool)
    for s in range(T):
        act = torch.rand(E, N, 2, device="cuda:0", generator=g) * 2 - 1
        pos0 = host(env.pos).astype(np.float64); vel0 = host(env.vel).astype(np.float64); t0 = host(env.t).copy()
        res = env.step(act)
        own[0] += host(res.rewards).astype(np.float64).mean(1)
        own[1] += host(res.true_rewards).astype(np.float64).mean(1)
        own[2] += host(res.n_collisions)
        if s % 4 == 0 or N <= 5:                       # oracle on a subsample of steps keeps the test in seconds
            r = orc.step(pos0, vel0, t0, host(act).astype(np.float64))
            safe = orc.margins(pos0) > H.MARGIN       # pos0 was integrated in place: the post-step state
            safe_all &= safe
            ref[0] += np.where(safe, r["reward"].mean(1), 0); ref[1] += np.where(safe, r["true_reward"].mean(1), 0)
            ref[2] += np.where(safe, r["n_coll"], 0)
            own_s = host(res.rewards).astype(np.float64).mean(1)
            H.assert_close(own_s[safe], r["reward"].mean(1)[safe], f"step {s} mean reward", rtol=1e-5, atol=1e-5)
    torch.cuda.synchronize()
    rec = records(env)
    assert np.array_equal(rec["ep_len"], np.full(E, T)) and np.array_equal(rec["ep_collisions"], own[2].astype(np.int64))
    # the kernel adds float32 per-env sums (one fixed-order wave reduction per step) into float64 records
    np.testing.assert_allclose(rec["ep_return"], own[0], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(rec["ep_true_return"], own[1], rtol=2e-6, atol=1e-6)
    assert np.all(rec["episodes"] == 0) and np.all(rec["done_len"] == 0)
    assert safe_all.any()
    # explicit reset retires the episode (train_problem.py:118-121 then :132)
    env.reset(renew_obstacles=False)
    torch.cuda.synchronize()
    rec2 = records(env)
    assert np.all(rec2["episodes"] == 1) and np.all(rec2["ep_len"] == 0) and np.all(rec2["ep_return"] == 0)
    np.testing.assert_array_equal(rec2["done_return"], rec["ep_return"])
    np.testing.assert_array_equal(rec2["done_collisions"], rec["ep_collisions"])
    np.testing.assert_array_equal(rec2["done_len"], np.full(E, T))
    # the reduction every rank feeds into the exchange: fixed order, equals the host sum
    tot = host(env.episode_totals())
    a = host(env.episode_acc)
    np.testing.assert_allclose(tot[0], a[:, 4].sum(), rtol=1e-13)
    np.testing.assert_allclose(tot[1], a[:, 5].sum(), rtol=1e-13)
    assert tot[2] == rec2["done_collisions"].sum() and tot[3] == E * T and tot[4] == E and tot[7] == 0
    tot2 = host(env.episode_totals()).copy()
    assert np.array_equal(tot, tot2)


def test_bookkeeping_does_not_change_the_step(torch):
    """The *_ex kernels with records on produce bit-identical step outputs to the plain kernels."""
    for N, G, E, c in [(64, 28.0, 300, 2), (5, 5.0, 77, 2), (9, 8.0, 50, 5), (200, 200.0, 6, 2), (64, 28.0, 33, 5)]:
        a = make_env(N, G, E, c=c, seed=9)
        b = make_env(N, G, E, c=c, seed=9, track_episodes=True)
        g = torch.Generator(device="cuda:0").manual_seed(1)
        for s in range(6):
            act = torch.rand(E, N, 2, device="cuda:0", generator=g) * 2 - 1
            ra, rb = a.step(act), b.step(act)
            for name in ("pos", "vel", "t", "reward", "true_reward", "z", "nbr_idx", "n_coll", "done"):
                assert torch.equal(getattr(a, name), getattr(b, name)), (N, c, s, name)


# ------------------------------------------------------------------------------- in-kernel auto-reset
AUTO_SHAPES = [(5, 5.0, 300, 2, 2), (64, 28.0, 128, 2, 2), (48, 24.0, 40, 3, 2), (130, 130.0, 10, 2, 2),
               (9, 8.0, 60, 2, 5), (64, 28.0, 20, 2, 5), (300, 300.0, 3, 2, 2)]


@pytest.mark.parametrize("N,G,E,k,c", AUTO_SHAPES, ids=lambda v: str(v))
def test_auto_reset_equals_step_then_masked_reset(torch, N, G, E, k, c):
    """auto_reset=True (reset + re-observation inside the step launch) is bit-identical to what the reference's
    loop does: step, then reset the envs whose `finished` fired (train_problem.py:82, 132): same fresh states
    (same Philox stream as dronesim_reset), same observation, same retired episode records."""
    A = make_env(N, G, E, k=k, c=c, seed=21, auto_reset=True, keep_final_obs=True)
    B = make_env(N, G, E, k=k, c=c, seed=21, track_episodes=True)
    assert torch.equal(A.pos, B.pos)
    # terminal observation (DroneEpisodeCtl.z_final / nbr_final / pos_final): rows of envs that have not finished in a
    # launch must be left untouched -- tracked here from a recognisable fill
    A.z_final.fill_(-7.0); A.nbr_final.fill_(-7); A.pos_final.fill_(-7.0)
    zf, nf, pf = A.z_final.clone(), A.nbr_final.clone(), A.pos_final.clone()
    # stagger the time limit over the envs so that resets hit different envs at different steps, and drive a third
    # of the envs with the P-controller so that some finish by ARRIVAL (drone_env.py:251)
    t0 = (torch.arange(E, device="cuda:0", dtype=torch.int32) * 7) % 23 + 180
    A.t.copy_(t0); B.t.copy_(t0)
    g = torch.Generator(device="cuda:0").manual_seed(4)
    n_done = 0
    for s in range(40):
        act = torch.rand(E, N, 2, device="cuda:0", generator=g) * 2 - 1
        if N <= 9:
            ctrl = B.control("proportional")
            act[::3] = ctrl[::3]
        ra = A.step(act)
        rb = B.step(act, copy=True)
        done = rb.finished.bool()
        n_done += int(done.sum())
        # what the reference's step() returns on the last call of an episode (drone_env.py:258) -- z-states, Ni and
        # state of the FINAL state -- is what B holds before its reset, and what A's launch kept in *_final, bit for bit
        zf[done] = rb.z_states[done]; nf[done] = B.nbr_idx[done]; pf[done] = rb.state.pos[done]
        assert torch.equal(A.z_final, zf) and torch.equal(A.nbr_final, nf) and torch.equal(A.pos_final, pf), s
        if bool(done.any()):
            B.reset(renew_obstacles=False, mask=done)
        for name, x, y in (("reward", ra.rewards, rb.rewards), ("true_reward", ra.true_rewards, rb.true_rewards),
                           ("n_coll", ra.n_collisions, rb.n_collisions), ("done", ra.finished, rb.finished)):
            assert torch.equal(x, y), (s, name)
        for name in ("pos", "vel", "t", "z", "nbr_idx", "episode"):
            assert torch.equal(getattr(A, name), getattr(B, name)), (s, name)
        assert torch.equal(A.episode_acc, B.episode_acc), s
    assert n_done >= E                                   # every env ended at least one episode
    rec = records(A)
    assert rec["episodes"].min() >= 1 and np.array_equal(rec["done_len"] + rec["ep_len"], 40 + 0 * rec["ep_len"])
    if N <= 9:
        assert (rec["done_len"][::3] < 100).any()        # arrival-terminated episodes are short


def test_episode_layer_shape_fuzz(torch):
    """Seeded random shapes (N 2..300, k 1..8, c 2 / 5, uniform / heterogeneous / default Delta, ragged E) through the
    episode layer: auto_reset == step + reset(mask=done) bit for bit over staggered episode ends, and the fused
    random-action rollout with auto_reset == the same steps launched one by one."""
    import os
    from scalable_collision_avoidance_rl_amd import formation_O
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", 5)))
    iters, ran, ends = int(os.environ.get("FUZZ_ITERS", 10)), 0, 0
    for it in range(iters):
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
        tag = f"episode fuzz#{it} N={N} k={k} c={c} G={G:.2f} {mode} E={E}"
        A = make_env(N, G, E, k=k, c=c, deltas=deltas, seed=seed, auto_reset=True)
        B = make_env(N, G, E, k=k, c=c, deltas=deltas, seed=seed, track_episodes=True)
        t0 = torch.tensor(rng.integers(185, 200, E).astype(np.int32), device="cuda:0")
        A.t.copy_(t0); B.t.copy_(t0)
        g = torch.Generator(device="cuda:0").manual_seed(it)
        for s in range(18):
            act = torch.rand(E, N, 2, device="cuda:0", generator=g) * 2 - 1
            ra = A.step(act); rb = B.step(act, copy=True)
            done = rb.finished.bool()
            ends += int(done.sum())
            if bool(done.any()):
                B.reset(renew_obstacles=False, mask=done)
            for name, x, y in (("reward", ra.rewards, rb.rewards), ("n_coll", ra.n_collisions, rb.n_collisions), ("done", ra.finished, rb.finished)):
                assert torch.equal(x, y), (tag, s, name)
            for name in ("pos", "vel", "t", "z", "nbr_idx", "episode"):
                x, y = getattr(A, name), getattr(B, name)
                assert torch.equal(x, y) or (name == "z" and torch.equal(torch.nan_to_num(x, nan=7.0), torch.nan_to_num(y, nan=7.0))), (tag, s, name)
            assert torch.equal(A.episode_acc, B.episode_acc), (tag, s)
        # fused rollout with in-kernel actions and resets == the same steps one by one
        C1 = make_env(N, G, E, k=k, c=c, deltas=deltas, seed=seed, auto_reset=True)
        C2 = make_env(N, G, E, k=k, c=c, deltas=deltas, seed=seed, auto_reset=True)
        C1.t.copy_(t0); C2.t.copy_(t0)
        T = 12
        out = C1.rollout_random(T, record_actions=True)
        for s in range(T):
            r2 = C2.step(out["actions"][s])
            assert torch.equal(r2.rewards, out["reward"][s]) and torch.equal(r2.finished, out["done"][s]), (tag, "rollout", s)
        for name in ("pos", "vel", "t", "episode"):
            assert torch.equal(getattr(C1, name), getattr(C2, name)), (tag, "rollout", name)
        assert torch.equal(C1.episode_acc, C2.episode_acc), (tag, "rollout records")
        ran += 1
    assert ran >= iters // 2 and ends >= ran                     # shapes were run and episodes did end inside them


def test_auto_reset_first_state_matches_the_oracle_reset(torch):
    """The fresh state an env gets from the in-kernel reset is the oracle's `reset` draw for that episode counter."""
    N, G, E = 64, 28.0, 64
    env = make_env(N, G, E, seed=4242, auto_reset=True)
    orc = Oracle(N, [G, G], 2, np.ones(N), True, threads=4)
    rpos, rvel, rt, rnode, repi = orc.reset(E, 4242)
    np.testing.assert_allclose(host(env.pos), rpos, rtol=1e-6, atol=1e-6)
    env.t.fill_(199)
    env.step(torch.zeros(E, N, 2, device="cuda:0"))
    torch.cuda.synchronize()
    r2 = orc.reset(E, 4242, pos=rpos, vel=rvel, t=rt, episode=repi)
    np.testing.assert_allclose(host(env.pos), r2[0], rtol=1e-6, atol=1e-6)
    assert int(env.t.abs().max()) == 0 and int(env.vel.abs().max()) == 0 and host(env.episode).tolist() == [2] * E
    ref = orc.observe(host(env.pos).astype(np.float64))
    safe = orc.margins(host(env.pos).astype(np.float64)) > H.MARGIN
    np.testing.assert_array_equal(host(env.nbr_idx)[safe], ref["nbr_idx"][safe])


def test_auto_reset_under_graph_replay_and_rollout(torch):
    """A captured step with auto_reset replays across episode ends (all counters live on the device); the fused
    rollout with auto_reset equals the same steps launched one by one."""
    N, G, E, T = 64, 28.0, 96, 30
    A = make_env(N, G, E, seed=3, auto_reset=True)
    B = make_env(N, G, E, seed=3, auto_reset=True)
    A.t.fill_(185); B.t.fill_(185)
    g = torch.Generator(device="cuda:0").manual_seed(8)
    act = torch.rand(T, E, N, 2, device="cuda:0", generator=g) * 2 - 1
    out = A.rollout(act)
    # B: the same steps through a captured graph of ONE step, replayed T times
    buf = torch.zeros(E, N, 2, device="cuda:0")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        B.step(buf)
    for s in range(T):
        buf.copy_(act[s])
        graph.replay()
        torch.cuda.synchronize()
        for name, ref in (("reward", B.reward), ("true_reward", B.true_reward), ("z", B.z), ("nbr_idx", B.nbr_idx),
                          ("n_coll", B.n_coll), ("done", B.done)):
            assert torch.equal(out[name][s], ref), (name, s)
    assert torch.equal(A.pos, B.pos) and torch.equal(A.t, B.t) and torch.equal(A.episode, B.episode)
    assert torch.equal(A.episode_acc, B.episode_acc)
    assert int(out["done"].sum()) == E and int(A.t[0]) == T - 15 and host(A.episode).tolist() == [2] * E


# ------------------------------------------------------------------------------- in-kernel random actions
@pytest.mark.parametrize("N,G,E,c", [(64, 28.0, 200, 2), (5, 5.0, 333, 2), (130, 130.0, 8, 2), (9, 8.0, 40, 5), (48, 24.0, 50, 2)])
def test_rollout_random_stream_and_equivalence(torch, N, G, E, c):
    """dronesim_rollout_random: the actions are bit-exactly the oracle's restatement of the documented Philox
    stream (RandomAgent.forward, SAC_agents.py:9-22), the rollout equals dronesim_rollout fed with those actions,
    splitting T over two calls changes nothing, and a shard sees exactly its slice."""
    T = 21
    A = make_env(N, G, E, c=c, seed=77)
    B = make_env(N, G, E, c=c, seed=77)
    A.t.fill_(3); B.t.fill_(3)
    out = A.rollout_random(T, record_actions=True)
    torch.cuda.synchronize()
    acts = host(out["actions"])
    epi = host(B.episode)
    for s in range(T):
        want = O.rand_actions(N, np.full(E, 3 + s, np.int32), epi, 77).astype(np.float32)
        assert np.array_equal(acts[s], want), s
    assert acts.min() >= -1.0 and acts.max() < 1.0
    ref = B.rollout(out["actions"])
    for name in ("reward", "true_reward", "z", "nbr_idx", "n_coll", "done"):
        assert torch.equal(out[name], ref[name]), name
    assert torch.equal(A.pos, B.pos) and torch.equal(A.vel, B.vel) and torch.equal(A.t, B.t)
    # two calls == one call; no action record needed
    Cc = make_env(N, G, E, c=c, seed=77); Cc.t.fill_(3)
    o1 = Cc.rollout_random(8); o2 = Cc.rollout_random(T - 8)
    assert torch.equal(torch.cat([o1["reward"], o2["reward"]]), out["reward"]) and torch.equal(Cc.pos, A.pos)
    assert "actions" not in o1
    # shard invariance (streams keyed by the global env id)
    for r in range(2):
        part = make_env(N, G, E, c=c, seed=77, rank=r, world_size=2); part.t.fill_(3)
        po = part.rollout_random(T)
        assert torch.equal(po["reward"], out["reward"][:, part.env_lo:part.env_hi])
        assert torch.equal(part.pos, A.pos[part.env_lo:part.env_hi])


def test_rollout_random_statistics_and_auto_reset(torch):
    """U(-1,1) moments of the in-kernel actions; a random rollout with auto_reset runs through episode ends and keeps
    the episode records (200-step episodes, one retired per env per 200 steps)."""
    N, G, E, T = 64, 28.0, 256, 450
    env = make_env(N, G, E, seed=5, auto_reset=True)
    out = env.rollout_random(T, record_actions=True)
    torch.cuda.synchronize()
    a = host(out["actions"]).astype(np.float64)
    n = a.size
    assert abs(a.mean()) < 4 / np.sqrt(3 * n) and abs(a.var() - 1 / 3) < 1e-3
    assert abs(np.corrcoef(a[:-1].ravel(), a[1:].ravel())[0, 1]) < 5e-3          # consecutive steps
    assert abs(np.corrcoef(a[..., 0].ravel(), a[..., 1].ravel())[0, 1]) < 5e-3    # the two components
    rec = records(env)
    assert np.all(rec["episodes"] == 2) and np.all(rec["done_len"] == 400) and np.all(rec["ep_len"] == 50)
    done = host(out["done"])
    assert done[199].all() and done[399].all() and done.sum() == 2 * E
    # actions of step 200 (first step of episode 2) use the incremented episode counter
    want = O.rand_actions(N, np.zeros(E, np.int32), np.full(E, 2, np.int32), 5).astype(np.float32)
    assert np.array_equal(host(out["actions"][200]), want)
    own = host(out["reward"]).astype(np.float64).mean(2)          # [T, E]
    np.testing.assert_allclose(rec["done_return"], own[:400].sum(0), rtol=2e-6)
    np.testing.assert_allclose(rec["ep_return"], own[400:].sum(0), rtol=2e-6)


# ------------------------------------------------------------------------------- checkpoint / live constants
def test_checkpoint_resumes_the_same_streams(torch):
    """get_state() / load_state(): a resumed env replays the SAME reset streams and in-kernel actions."""
    N, G, E = 5, 5.0, 64
    a = make_env(N, G, E, seed=31, auto_reset=True)
    a.rollout_random(150)
    ck = a.get_state()
    cont = a.rollout_random(120)                       # crosses the 200-step limit: in-kernel resets
    b = make_env(N, G, E, seed=999, auto_reset=True)  # a different env object, different seed
    b.load_state(ck)
    cont_b = b.rollout_random(120)
    for name in ("reward", "z", "nbr_idx", "done"):
        assert torch.equal(cont[name], cont_b[name]), name
    assert torch.equal(a.pos, b.pos) and torch.equal(a.episode, b.episode) and torch.equal(a.episode_acc, b.episode_acc)
    assert int(cont["done"].sum()) == E


def test_live_constants_are_reuploaded(torch):
    """deltas / d_safety / drone_radius are read on every step by the reference (drone_env.py:242): assigning them
    on the env re-uploads the device copies."""
    N, G, E = 5, 5.0, 128
    env = make_env(N, G, E, seed=2)
    rng = np.random.default_rng(0)
    pos = (2.5 + (rng.random((E, N, 2)) - 0.5) * 2.5).astype(np.float32)
    zero = torch.zeros(E, N, 2, device="cuda:0")
    env.set_state(pos); env.step(zero); r1 = env.reward.clone(); nb1 = env.nbr_idx.clone()
    env.deltas = np.ones(N) * 0.3
    env.set_state(pos); env.step(zero)
    orc = Oracle(N, [G, G], 2, np.ones(N) * 0.3, True)
    ref = orc.observe(pos.astype(np.float64))
    safe = orc.margins(pos.astype(np.float64)) > H.MARGIN
    np.testing.assert_array_equal(host(env.nbr_idx)[safe], ref["nbr_idx"][safe])
    H.assert_close(host(env.reward)[safe], ref["reward"][safe], "reward after deltas change")
    assert not torch.equal(nb1, env.nbr_idx) and not torch.equal(r1, env.reward)


def test_step_copy_returns_fresh_tensors(torch):
    env = make_env(5, 5.0, 16, seed=1)
    act = torch.rand(16, 5, 2, device="cuda:0")
    r1 = env.step(act, copy=True)
    keep = r1.rewards.clone()
    r2 = env.step(act, copy=True)
    assert r1.rewards.data_ptr() != r2.rewards.data_ptr() and torch.equal(r1.rewards, keep)
    assert not torch.equal(r1.rewards, r2.rewards)
    live = env.step(act)
    assert live.rewards.data_ptr() == env.reward.data_ptr()


def test_ex_entry_points_reject_bad_arguments(torch):
    import ctypes as C
    env = make_env(5, 5.0, 4, seed=1, track_episodes=True)
    lib, nat = env._lib, env._native
    p = env._params()
    ctl = nat.DroneEpisodeCtl(); ctl.auto_reset = 1                      # no episode pointer
    args = [env.pos, env.vel, env.t, env._act, env.reward, env.true_reward, env.z, env.nbr_idx, env.n_coll, env.done]
    rc = lib.dronesim_step_ex(C.byref(p), C.byref(ctl), *[C.c_void_p(t.data_ptr()) for t in args], 4, None)
    assert rc == nat.EINVAL and b"episode" in lib.dronesim_last_error()
    rc = lib.dronesim_rollout_random(C.byref(p), None, *[C.c_void_p(t.data_ptr()) for t in args], 4, 3, None)
    assert rc == nat.EINVAL
    assert lib.dronesim_episode_reduce(None, 4, None, None) == nat.EINVAL
    assert lib.dronesim_reset_ex(C.byref(p), None, None, None, None, None, None, 4, None) == nat.EINVAL


# ------------------------------------------------------------------------------- round 4: ADVICE r3 items
@pytest.mark.parametrize("N,G,E,k,c", [(64, 28.0, 40, 2, 2), (5, 5.0, 100, 2, 2), (130, 130.0, 6, 2, 2), (9, 8.0, 30, 2, 5)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("random_actions", [False, True])
def test_rollout_keeps_terminal_observations_per_step(torch, N, G, E, k, c, random_actions):
    """Fused rollout + keep_final_obs: the kernel writes the terminal rows of an env that finishes at step s at
    [s][e][i] (like z), so rollout() hands it [T, E, N, ...] buffers of its own (never the [E, ...] home buffers of
    step(): ADVICE r3 high) and returns them; they equal what step-by-step launches keep, bit for bit, untouched rows
    stay at their fill, and nothing outside the buffers is written (a guard tensor allocated right behind stays intact)."""
    T = 45
    A = make_env(N, G, E, k=k, c=c, seed=31, auto_reset=True, keep_final_obs=True)
    B = make_env(N, G, E, k=k, c=c, seed=31, auto_reset=True, keep_final_obs=True)
    t0 = (torch.arange(E, device="cuda:0", dtype=torch.int32) * 5) % 31 + 165       # every env finishes inside the rollout
    A.t.copy_(t0); B.t.copy_(t0)
    if random_actions:
        out = A.rollout_random(T, record_actions=True)
        act = out["actions"]
    else:
        g = torch.Generator(device="cuda:0").manual_seed(8)
        act = torch.rand(T, E, N, 2, device="cuda:0", generator=g) * 2 - 1
        out = A.rollout(act)
    assert out["z_final"].shape == (T, E, N, (k + 1) * c) and out["nbr_final"].shape == (T, E, N, k + 1)
    n_done = 0
    for s in range(T):
        B.z_final.fill_(0.0); B.nbr_final.fill_(-1); B.pos_final.fill_(0.0)       # rollout()'s fill values
        rb = B.step(act[s])
        d = rb.finished.bool()
        n_done += int(d.sum())
        assert torch.equal(out["done"][s], rb.finished)
        for name in ("z_final", "nbr_final", "pos_final"):
            assert torch.equal(out[name][s], getattr(B, name)), (name, s)
    assert n_done >= E
    for name in ("pos", "t", "z", "nbr_idx", "z_final", "nbr_final", "pos_final"):
        assert torch.equal(getattr(A, name), getattr(B, name)), name
    assert torch.equal(A.episode_acc, B.episode_acc)


def test_state_injection_does_not_clobber_a_bound_storage_slot(torch):
    """While the env is bound to a RolloutStorage slot its observation attributes are views of the storage; reset /
    set_state / load_state / rollout return to the env's own buffers first."""
    from scalable_collision_avoidance_rl_amd.rollout_buffer import RolloutStorage
    env = make_env(5, 5.0, 16, seed=2, auto_reset=True)
    st = RolloutStorage(env, 4).begin()
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for t in range(4):
        env.step(torch.rand(16, 5, 2, device="cuda:0", generator=g) * 2 - 1, into=(st, t))
    keep = {n: getattr(st, n).clone() for n in ("zbuf", "nbrbuf", "reward", "true_reward", "n_coll", "done")}
    z_now = env.z.clone()
    ck = env.get_state()
    env.set_state(env.pos.clone() + 0.3)
    env.reset(renew_obstacles=False)
    env.rollout_random(3)
    env.load_state(ck)
    for n, v in keep.items():
        assert torch.equal(getattr(st, n), v), n
    assert env.z.data_ptr() == env._home["z"].data_ptr() and torch.equal(env.z, z_now)


@pytest.mark.parametrize("N,G,E,k,c,default_delta", [(64, 28.0, 64, 2, 2, False), (5, 5.0, 200, 2, 2, False), (130, 130.0, 8, 2, 2, False),
                                                      (256, 256.0, 6, 2, 2, False), (300, 300.0, 3, 3, 2, False), (9, 8.0, 40, 2, 5, False),
                                                      (64, 28.0, 20, 3, 5, False), (48, 24.0, 30, 4, 2, False)], ids=lambda v: str(v))
@pytest.mark.parametrize("entry", ["rollout", "rollout_random", "step"])
def test_reobservation_after_in_kernel_reset_matches_the_oracle(torch, N, G, E, k, c, default_delta, entry):
    """ORACLE evidence for the cold path (the rollout == step tests are HIP against HIP): every env's episode
    ends in the LAST step of a fused rollout (or in a single step), so what the launch leaves in z / nbr_idx is the
    in-kernel re-observation of the freshly drawn state (drone_env.py:208-210 after :98-102) -- compared with the float64
    oracle's observation of that state, neighbour ids exactly (drone_env.py:346, 362-365)."""
    deltas = None if default_delta else np.ones(N)
    from scalable_collision_avoidance_rl_amd import drones
    env = drones(N, 0, [G, G], "O", k_closest=k, deltas=deltas, simplify_zstate=(c == 2), n_envs=E, batched=True,
                 device="cuda:0", seed=77, auto_reset=True)
    orc = Oracle(N, [G, G], k, deltas, c == 2, threads=4)
    T = 6
    env.t.fill_(200 - T if entry != "step" else 199)
    if entry == "rollout":
        g = torch.Generator(device="cuda:0").manual_seed(3)
        out = env.rollout(torch.rand(T, E, N, 2, device="cuda:0", generator=g) * 2 - 1)
        z, nb, done = out["z"][-1], out["nbr_idx"][-1], out["done"]
        assert not bool(done[:-1].any())
    elif entry == "rollout_random":
        out = env.rollout_random(T)
        z, nb, done = out["z"][-1], out["nbr_idx"][-1], out["done"]
    else:
        res = env.step(torch.rand(E, N, 2, device="cuda:0") * 2 - 1)
        z, nb, done = res.z_states, env.nbr_idx, res.finished[None]
    torch.cuda.synchronize()
    assert bool(done[-1].all()) and int(env.t.abs().max()) == 0 and int(env.vel.abs().max()) == 0
    p = host(env.pos).astype(np.float64)
    ref = orc.observe(p, np.zeros_like(p))
    safe = orc.margins(p) > H.MARGIN
    assert safe.mean() > 0.5
    np.testing.assert_array_equal(host(nb)[safe], ref["nbr_idx"][safe])
    zz = host(z).reshape(E, N, k + 1, c)
    m = H.z_compare_mask(ref["nbr_idx"], np.ones((E, N), bool) if not default_delta else np.zeros((E, N), bool), c)
    H.assert_close(np.where(m, zz, 0)[safe], np.where(m, ref["z"], 0)[safe], "re-observed z", atol=H.atol_coord(G))
    assert torch.equal(env.z, z) and torch.equal(env.nbr_idx, nb)


@pytest.mark.parametrize("N,G,E", [(1024, 1030.0, 2), (300, 300.0, 6), (130, 130.0, 16), (256, 256.0, 12)], ids=lambda v: str(v))
def test_in_kernel_reset_of_multi_wave_envs_draws_the_oracle_state_every_time(torch, N, G, E):
    """Envs of several waves: every wave must draw its agents' lattice nodes from the SAME episode counter.  Through
    round 3 agent 0 stored the incremented counter while slower waves of the env could still be reading it (a 1-in-5
    flake found by the rollout fuzz: 64 agents of an env on another Philox stream).  The state that ends the episode is
    UNBALANCED on purpose -- the first wave's agents far from everybody (no pair work: it reaches the reset first), the
    others crowded -- and the step is repeated: the fresh state is the oracle's draw (drone_env.py:98-102, 171-205 restated
    in oracle_reset) and the same bits on every repetition.  (The race needed a wave to reach its read ~0.5 us late -- e.g. an
    instruction-cache miss on the cold path -- so this test did not fail reliably on the old code; the repeated rollout fuzz
    of tests/test_gpu_fuzz.py is the statistical guard: 5 of 6 runs failed before the fix, 0 of 8 after.)"""
    orc = Oracle(N, [G, G], 2, np.ones(N), True, threads=4)
    rpos, rvel, rt, rnode, repi = orc.reset(E, 99)
    want = orc.reset(E, 99, pos=rpos.copy(), vel=rvel.copy(), t=rt.copy(), episode=repi.copy())[0]
    rng = np.random.default_rng(1)
    pos0 = np.empty((E, N, 2), np.float32)
    pos0[:, :64, 0] = np.linspace(0.05 * G, 0.95 * G, 64); pos0[:, :64, 1] = 0.02 * G          # a sparse line
    pos0[:, 64:] = (0.5 * G + (rng.random((E, N - 64, 2)) - 0.5) * min(0.05 * G, 12.0)).astype(np.float32)   # a crowd
    act = torch.zeros(E, N, 2, device="cuda:0")
    first = None
    for rep in range(20):
        env = make_env(N, G, E, seed=99, auto_reset=True)
        env.set_state(pos0, None, np.full(E, 199, np.int32))
        res = env.step(act)
        torch.cuda.synchronize()
        assert bool(res.finished.all())
        got = host(env.pos)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6, err_msg=f"repetition {rep}")
        if first is None:
            first = got
        assert np.array_equal(got, first), (rep, int((got != first).sum()))
        assert host(env.episode).tolist() == [2] * E
