"""Pin the CPU oracle (oracle/drone_oracle.c) against outputs of the reference itself.

CPU-only.  Fixtures: tests/golden/*.npz written by tests/golden/gen_golden.py from the
unmodified /root/reference/drone_env.py.  Tolerance is float64 round-off (both sides are
float64; libm vs NumPy differ by an ulp in sqrt/log)."""
import numpy as np
import pytest

from oracle.oracle import Oracle, formation
from tests import helpers as H

F64 = dict(rtol=1e-11, atol=1e-11)


def test_formation_matches_reference():
    fx = H.load("formation.npz")
    keys = [k[3:] for k in fx.files if k.startswith("xF_")]
    assert len(keys) >= 10
    for tag in keys:
        n, g = tag.split("_")
        grid = [float(x) for x in g.split("x")] if "x" in g else [float(g), float(g)]
        xF, dhat = formation(int(n), grid)
        H.assert_close(xF, fx[f"xF_{tag}"], f"xF {tag}", **F64)
        np.testing.assert_array_equal(dhat, fx[f"dhat_{tag}"])   # floored to 0.01 -> exact


def test_known_answer_survey_8c():
    """Hand-checkable KAT of SURVEY.md 8c (values from the reference, float64)."""
    o = Oracle(4, [5, 5], 2, np.ones(4), True)
    assert np.allclose(o.xF, [(4.75, 2.5), (2.5, 4.75), (0.25, 2.5), (2.5, 0.25)]) and np.all(o.d_hat == 2.98)
    pos = np.array([[(1, 1), (2, 1), (1, 2.2), (4, 4)]], float)
    vel = np.zeros_like(pos); t = np.zeros(1, np.int32)
    out = o.step(pos, vel, t, np.array([[(1, 0), (-1, 0), (0, -1), (.5, .5)]], float))
    H.assert_close(pos[0], [(1.05, 1), (1.95, 1), (1, 2.15), (4.025, 4.025)], "pos", **F64)
    H.assert_close(out["reward"][0], [-1.6199067186825171, -1.4509859824445603, -0.07992073623795659,
                                      -1.6576250000000003], "r", **F64)
    H.assert_close(out["true_reward"][0], [-1.6199067186825171, -1.4593460619240424, -0.08828081571743865,
                                           -1.6576250000000003], "true_r", **F64)
    assert out["n_coll"][0] == 0 and out["done"][0] == 0 and t[0] == 1
    assert out["nbr_idx"][0].tolist() == [[0, 1, 2], [1, 0, -1], [2, 0, -1], [3, -1, -1]]
    H.assert_close(out["z"][0, 0].ravel(), [-3.7, -1.5, 0.9, 0, -0.05, 1.15], "z0", **F64)
    H.assert_close(out["z"][0, 1].ravel(), [-0.55, -3.75, -0.9, 0, -0.1596256061711044, -1.0883564057120754],
                   "z1", **F64)
    # collision KAT: heterogeneous deltas, exhibits Q1 (Delta_j) and Q6 (sort order vs mask count)
    o = Oracle(4, [5, 5], 2, np.array([.3, 2, .3, 2]), False)
    out = o.observe(np.array([[(1, 1), (1.15, 1), (1, 2.2), (4, 4)]], float))
    assert out["n_coll"][0] == 2
    assert out["nbr_idx"][0].tolist() == [[0, 1, -1], [1, 0, -1], [2, 0, -1], [3, -1, -1]]
    assert np.allclose(out["reward"][0], [-101.5313, -101.4885, -0.0761, -1.6312], atol=5e-4)
    assert np.allclose(out["true_reward"][0], [-101.5422, -101.4993, -0.0870, -1.6312], atol=5e-4)


@pytest.mark.parametrize("path", H.single_step_files(), ids=lambda p: p.split("single_step_")[1][:-4])
def test_single_step_matches_reference(path):
    fx = np.load(path)
    o = H.oracle_for(fx)
    np.testing.assert_array_equal(o.d_hat, fx["d_hat"])
    H.assert_close(o.delta, fx["deltas"], "deltas", **F64)
    H.assert_close(o.xF, fx["xF"], "xF", **F64)
    pos = fx["pos0"].copy(); vel = fx["vel0"].copy(); t = fx["t0"].astype(np.int32).copy()
    out = o.step(pos, vel, t, fx["act"])
    H.assert_close(pos, fx["pos1"], "pos", **F64)
    H.assert_close(vel, fx["vel1"], "vel", **F64)
    np.testing.assert_array_equal(t, fx["t0"] + 1)
    H.assert_close(out["reward"], fx["reward"], "reward", **F64)
    H.assert_close(out["true_reward"], fx["true_reward"], "true_reward", **F64)
    np.testing.assert_array_equal(out["n_coll"], fx["n_coll"])
    np.testing.assert_array_equal(out["done"].astype(bool), fx["done"])
    np.testing.assert_array_equal(out["nbr_idx"], fx["nbr_idx"])
    m = H.z_compare_mask(fx["nbr_idx"], fx["row_tie_free"], int(fx["c"]))
    H.assert_close(np.where(m, out["z"], 0), np.where(m, fx["z"], 0), "z", **F64)
    # the oracle's own margin estimate agrees with the generator's (used by the GPU tests)
    assert np.all(o.margins(fx["pos1"]) > 0.99 * H.MARGIN)


def test_init_states_observe_matches_reference():
    fx = H.load("init_states.npz")
    tags = [k[6:] for k in fx.files if k.startswith("state_")]
    assert len(tags) == 24
    for tag in tags:
        n, g, cc, _ = tag.split("_")
        N, G, c = int(n), float(g), int(cc[1])
        o = Oracle(N, [G, G], 2, np.ones(N), c == 2)
        st = fx[f"state_{tag}"]
        # lattice nodes: multiples of the pitch, distinct, inside the grid (drone_env.py:193-205)
        q = st[:, :2] / 0.22000000000000003
        assert np.allclose(q, np.round(q), atol=1e-9) and len({tuple(r) for r in np.round(q)}) == N
        assert np.all(st[:, 2:4] == 0) and np.all(st[:, 4] == 0.1)
        out = o.observe(st[None, :, :2], st[None, :, 2:4])
        np.testing.assert_array_equal(out["nbr_idx"][0], fx[f"nbr_{tag}"])
        m = H.z_compare_mask(fx[f"nbr_{tag}"][None], fx[f"tiefree_{tag}"][None], c)
        H.assert_close(np.where(m, out["z"], 0)[0], np.where(m[0], fx[f"z_{tag}"], 0), f"z {tag}", **F64)


def test_episode_c1_matches_reference():
    """Config C1 (N=5, one episode, softmax-16 policy): teacher-forced and free-running."""
    fx = H.load("episode_n5.npz")
    o = H.oracle_for(fx)
    T = fx["act"].shape[0]
    assert T == 200 and fx["done"][-1] and not fx["done"][:-1].any()
    out0 = o.observe(fx["state0"][None, :, :2])
    np.testing.assert_array_equal(out0["nbr_idx"][0], fx["nbr0"])
    H.assert_close(out0["z"][0], fx["z0"], "z0", **F64)
    # free-running float64: same arithmetic as the reference -> stays at round-off
    pos = fx["state0"][None, :, :2].copy(); vel = fx["state0"][None, :, 2:4].copy(); t = np.zeros(1, np.int32)
    for s in range(T):
        out = o.step(pos, vel, t, fx["act"][s][None])
        H.assert_close(pos[0], fx["pos"][s], f"pos@{s}", **F64)
        H.assert_close(out["reward"][0], fx["reward"][s], f"reward@{s}", rtol=1e-9, atol=1e-9)
        H.assert_close(out["true_reward"][0], fx["true_reward"][s], f"true_reward@{s}", rtol=1e-9, atol=1e-9)
        H.assert_close(out["z"][0], fx["z"][s], f"z@{s}", rtol=1e-9, atol=1e-9)
        assert out["n_coll"][0] == fx["n_coll"][s] and bool(out["done"][0]) == bool(fx["done"][s])
        np.testing.assert_array_equal(out["nbr_idx"][0], fx["nbr_idx"][s])
    assert t[0] == T


def test_oracle_batching_and_threads_are_consistent():
    fx = H.load("single_step_n64_c2.npz")
    o1 = H.oracle_for(fx, threads=1); o4 = H.oracle_for(fx, threads=4)
    a = o1.observe(fx["pos1"], fx["vel1"]); b = o4.observe(fx["pos1"], fx["vel1"])
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    one = o1.observe(fx["pos1"][3:4], fx["vel1"][3:4])
    np.testing.assert_array_equal(one["reward"][0], a["reward"][3])


def test_controllers_match_reference():
    """gradient_control / proportional_control (drone_env.py:609-679) on the reference's own outputs."""
    fx = H.load("controllers.npz")
    tags = [k[4:] for k in fx.files if k.startswith("pos_")]
    assert len(tags) == 4
    for tag in tags:
        n, g = tag.split("_")
        o = Oracle(int(n), [float(g), float(g)], 1, np.ones(int(n)), True)
        H.assert_close(o.gradient_control(fx[f"pos_{tag}"]), fx[f"grad_{tag}"], f"grad {tag}", rtol=1e-10, atol=1e-10)
        H.assert_close(o.proportional_control(fx[f"pos_{tag}"]), fx[f"prop_{tag}"], f"prop {tag}", **F64)
        assert np.abs(fx[f"grad_{tag}"]).max() <= 1.0 and (np.abs(fx[f"grad_{tag}"]) == 1.0).any()   # clipped at u_max


def test_returns_and_advantage_match_reference_episode():
    """MC returns as SA2CAgents.benchmark_cirtic returns them and the actor-loss weight of train_NN
    (SAC_agents.py:304-307, 333-351) on the C1 episode."""
    from oracle.oracle import mc_returns, neighbour_advantage
    fx = H.load("episode_n5.npz")
    gamma = float(fx["discount"])
    G = mc_returns(fx["reward"][:, None, :], gamma)
    H.assert_close(G[:, 0], fx["mc_return"], "G", rtol=1e-12, atol=1e-12)
    w = neighbour_advantage(G, fx["critic_value"][:, None, :], fx["nbr_idx_pre"][:, None], gamma)
    H.assert_close(w[:, 0], fx["adv_weight"], "adv", rtol=1e-10, atol=1e-10)
    # hand-checkable: T=3, gamma=0.5, rewards 1,2,4 -> G = 1+0.5*(2+0.5*4), 2+0.5*4, 4
    assert mc_returns(np.array([1., 2., 4.]).reshape(3, 1, 1), 0.5).ravel().tolist() == [3.0, 4.0, 4.0]
    assert mc_returns(np.array([1., 2., 4.]).reshape(3, 1, 1), 0.5, done=np.array([[0], [1], [0]])).ravel().tolist() == [2.0, 2.0, 4.0]
