"""Seeded fuzzers of the HIP path, in the driver-run suite (round 4; they lived under tools/ through round 3, where
`fuzz_rollout.py` caught a register-allocation-dependent mis-lowering of the re-observation after an in-kernel reset).

* rollout fuzz: dronesim_rollout_ex (T fused steps, candidate lists kept between steps, prefetched pool actions,
  in-kernel resets) against T dronesim_step_ex launches, bit for bit, over random shapes -- k = 1..8, every geometry
  (packed, kSym64, workgroup-per-env up to 256 agents AND the N > 256 class), c = 2 / 5, uniform / heterogeneous /
  default (`deltas=None`: the FAR variant) Delta, `auto_reset` on and off, slow / bursty / stand-still / teleporting
  actions.  The chain to the oracle: the step launches themselves are held to the oracle by the shape fuzz below and
  by tests/test_gpu_parity.py; the in-kernel re-observation by test_auto_reset_equals_step_then_masked_reset.
* the big shape fuzz (step + observe vs the float64 oracle, up to 1024 agents, crowded boxes), two seeds.
* the episode-layer fuzz (auto_reset == step + reset(mask), fused random-action rollouts == steps), two seeds.

Semantics guarded: /root/reference/drone_env.py:346, 362-365 (in-range count and neighbour selection)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch


@pytest.mark.parametrize("seed,iters,big_n", [(7, 150, False), (9, 150, False), (31, 150, False), (2026, 80, True), (2027, 80, True), (2028, 80, True), (2029, 80, True)])
def test_rollout_fuzz_against_step_launches(torch, seed, iters, big_n):
    from scalable_collision_avoidance_rl_amd import drones, formation_O
    rng = np.random.default_rng(seed)
    shapes = ([300, 320, 400, 512, 600, 1024, 257, 384] if big_n else
              [5, 24, 48, 64, 64, 65, 100, 128, 130, 192, 200, 250, 256, 256, 300])
    done, resets, classes = 0, 0, set()
    for it in range(iters):
        N = int(rng.choice(shapes))
        k = int(rng.integers(1, min(N - 1, 8) + 1))
        c = int(rng.choice([2, 2, 2, 5]))
        G = float(rng.choice([0.25, 0.45, 1.0])) * N + 6.0
        d_hat = formation_O(N, [G, G])[1]
        if d_hat.min() <= 0.05:
            continue
        kind = rng.choice(["uniform", "uniform", "hetero", "none"])
        deltas = (np.ones(N) * float(rng.uniform(0.2, 0.95)) * d_hat.min() if kind == "uniform"
                  else rng.uniform(0.1, 1.3, N) * d_hat.min() if kind == "hetero" else None)
        E = int(rng.integers(1, 40)) if N <= 130 else int(rng.integers(1, 10)) if N <= 300 else int(rng.integers(1, 4))
        T = int(rng.integers(20, 70)) if N <= 300 else int(rng.integers(12, 30))
        auto = bool(rng.integers(0, 2))
        kw = dict(auto_reset=True) if auto else {}
        try:
            mk = lambda: drones(N, 0, [G, G], "O", k_closest=k, deltas=deltas, simplify_zstate=(c == 2), n_envs=E,
                                batched=True, device="cuda:0", seed=100 + it, **kw)
            a, b = mk(), mk()
        except Exception as ex:                                # the one documented size limit (LDS tile at N ~ 1024, k = 8)
            assert "160 KiB LDS tile" in str(ex) and N > 900, (N, k, c, str(ex))
            continue
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
        tag = (seed, it, N, k, c, kind, E, T, auto)
        for s in range(T):
            res = b.step(act[s])
            for name, ref in (("reward", res.rewards), ("true_reward", res.true_rewards), ("z", res.z_states),
                              ("nbr_idx", b.nbr_idx), ("n_coll", res.n_collisions), ("done", res.finished)):
                x = out[name][s]
                assert torch.equal(x, ref) or (x.is_floating_point() and torch.equal(torch.nan_to_num(x, nan=7.0), torch.nan_to_num(ref, nan=7.0))), (tag, name, s)
        assert torch.equal(a.pos, b.pos) and torch.equal(a.t, b.t), tag
        resets += int(out["done"].sum()) if auto else 0
        classes.add((N <= 64, N == 64, N > 256, c, kind == "none", auto))
        done += 1
    assert done >= iters * 2 // 3 and resets > 0 and len(classes) >= (4 if big_n else 10), (done, resets, len(classes))


@pytest.mark.parametrize("seed", [11, 14])
def test_big_shape_fuzz_against_oracle(torch, seed, monkeypatch):
    """tools/fuzz_big.sh of round 3: the shape fuzz of test_gpu_parity with up to 1024 agents and crowded boxes."""
    from tests import test_gpu_parity as P
    monkeypatch.setenv("FUZZ_BIG", "1"); monkeypatch.setenv("FUZZ_SEED", str(seed)); monkeypatch.setenv("FUZZ_ITERS", "80")
    P.test_shape_fuzz_against_oracle(torch)


@pytest.mark.parametrize("seed", [101, 202])
def test_episode_layer_fuzz_more_seeds(torch, seed, monkeypatch):
    from tests import test_gpu_episodes as Ep
    monkeypatch.setenv("FUZZ_SEED", str(seed)); monkeypatch.setenv("FUZZ_ITERS", "60")
    Ep.test_episode_layer_shape_fuzz(torch)
