"""The launcher path on the one GPU a box has: several ranks of the product path started the way the
driver starts `bench.py --gpus N` (python -m torch.distributed.run, one process per rank), each owning a shard of the
env axis, plus one RCCL collective in a world of one so that the "nccl" backend is exercised at all."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _launch(nproc, script, args, extra_env=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script] + [str(a) for a in args]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("N,G,E,T,W", [(64, 28.0, 96, 70, 2), (5, 5.0, 250, 70, 2), (64, 28.0, 99, 70, 4)])
def test_ranks_own_their_shards_of_the_gpu_outputs(tmp_path, N, G, E, T, W):
    """W ranks x their shard of the envs (ragged at 99 over 4) == 1 rank x all envs, bit for bit (states, observations,
    rewards, episode records), and the all-gathered statistic of the W-rank run equals the 1-rank one."""
    worker = os.path.join("tests", "launch_worker.py")
    two, one = tmp_path / "two", tmp_path / "one"
    two.mkdir(); one.mkdir()
    r = _launch(W, worker, [two, N, G, E, T])
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, worker, str(one), str(N), str(G), str(E), str(T)], cwd=ROOT,
                       env=dict(os.environ, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    full = np.load(one / "rank0.npz")
    parts = [np.load(two / f"rank{k}.npz") for k in range(W)]
    bounds = [(int(p["lo"]), int(p["hi"])) for p in parts]
    assert bounds[0][0] == 0 and bounds[-1][1] == E and all(bounds[i][1] == bounds[i + 1][0] for i in range(W - 1))
    if W == 2:
        assert bounds == [(0, E // 2), (E // 2, E)]
    for name in ("pos", "z", "nbr", "acc", "last_reward"):
        assert np.array_equal(np.concatenate([p[name] for p in parts]), full[name]), name
    for name in ("reward", "done"):
        assert np.array_equal(np.concatenate([p[name] for p in parts], axis=1), full[name]), name
    assert full["done"].sum() == E and (full["acc"].view(np.int32)[:, 6] == 1).all()      # one episode ended per env
    want = dict(zip(full["summary_keys"].tolist(), full["summary_vals"].tolist()))
    for p in parts:                                            # every rank holds the same global figures
        got = dict(zip(p["summary_keys"].tolist(), p["summary_vals"].tolist()))
        assert got["world_size"] == W and want["world_size"] == 1 and got["episodes"] == want["episodes"]
        for k in ("mean_episode_reward", "mean_episode_true_reward", "mean_episode_collisions", "mean_episode_len",
                  "mean_reward", "agent_steps"):
            assert got[k] == pytest.approx(want[k], rel=1e-12), k


def test_bench_py_runs_with_two_ranks_on_one_device():
    """bench.py through the driver's launcher (2 ranks, gloo, one device): barrier, max-over-ranks timing, the
    exchange inside the timed region, one JSON line from rank 0 with whole-job figures."""
    r = _launch(2, "bench.py", ["--gpus", 2, "--steps", 30, "--warmup", 5, "--envs-per-gpu", 512, "--min-seconds", 0.05,
                                "--no-cpu-baseline"], extra_env=dict(BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["n_envs_total"] == 1024 and out["config"]["parallelism"] == "env-shard x2"
    assert out["steps"] == 30 and out["repeats"] >= 1 and out["timed_steps"] == 30 * out["graph_copies"] * out["repeats"]
    assert out["value"] == pytest.approx(64 * 1024 * out["timed_steps"] / out["timed_seconds"], rel=1e-5)
    st = out["episode_end_stats"]
    assert st["world_size"] == 2 and st["agent_steps"] > 0
    assert st["mean_reward"] < 0 and out["roofline"]["frac"] > 0
    # the exchange: one in-capture reduction per 200 steps, one collective per replay, and the line says a real one ran
    ex = out["exchange"]
    assert ex["real_collective_ran"] is True and ex["collectives_in_timed_region"] == out["repeats"]
    assert ex["reduce_every_steps"] == 200 and ex["reductions_in_timed_region"] == out["repeats"] * (out["timed_steps"] // out["repeats"] // 200)
    assert ex["collective_latency_us"] > 0
    # every launch this job issued is accounted for by the device-side episode records
    lc = out["launch_check"]
    assert lc["ok"] is True and lc["env_steps_recorded_all_ranks"] == lc["env_steps_issued_all_ranks"] >= lc["launches_rank0"] * 512
    # the N = 1 figure of the same workload, timed by rank 0 alone ahead of the 2-rank region, and the ratio the script computes
    assert out["n1"]["value"] > 0 and out["n1"]["ratio"] == pytest.approx(out["value"] / out["n1"]["value"], rel=1e-3)
    assert out["n1"]["efficiency"] == pytest.approx(out["n1"]["ratio"] / 2, rel=1e-3)
    assert len(lines[0]) < 8000                                 # the driver keeps an 8 KB tail of stdout
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and all(r["step_kernel_ms"] > 0 and r["envs"] == 512 for r in out["per_rank"])


def test_bench_py_runs_with_eight_ranks_on_one_device():
    """The shape of the driver's first SCALE run, on the one GPU a box has: `bench.py --gpus 8` as 8 ranks (gloo, all on
    device 0, 100 envs each): one JSON line from rank 0 with whole-job figures, `per_rank` for all 8 ranks, the exchange
    inside the timed region with its `rccl_ranks` / `backend` fields, every launch accounted for by the device records.
    (Ragged shards over 8 ranks: the next test.)"""
    r = _launch(8, "bench.py", ["--gpus", 8, "--steps", 20, "--warmup", 2, "--envs-per-gpu", 100, "--min-seconds", 0.03,
                                "--no-cpu-baseline"],
                # (eight processes time-slice one GPU: the secondary measurements are shrunk, the graded region is not)
                extra_env=dict(BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1", BENCH_EAGER_MIN_STEPS="40", BENCH_KERNEL_SAMPLE_EPISODES="1",
                               BENCH_GRAPH_LAUNCHES="400"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["n_envs_total"] == 800 and out["config"]["parallelism"] == "env-shard x8"
    assert out["scaling"] == "weak" and out["value"] == pytest.approx(64 * 800 * out["timed_steps"] / out["timed_seconds"], rel=1e-5)
    ex = out["exchange"]
    assert ex["real_collective_ran"] is True and ex["backend"] == "gloo" and "rccl_ranks" in ex and ex["rccl_ranks"] == 0
    assert ex["collectives_in_timed_region"] == out["repeats"] and ex["collective_latency_us"] > 0
    assert [q["rank"] for q in out["per_rank"]] == list(range(8))
    assert all(q["envs"] == 100 and q["step_kernel_ms"] > 0 and q["timed_seconds"] > 0 for q in out["per_rank"])
    assert out["launch_check"]["ok"] is True and out["episode_end_stats"]["world_size"] == 8
    assert out["launch_check"]["env_steps_recorded_all_ranks"] == out["launch_check"]["env_steps_issued_all_ranks"]
    assert out["n1"]["value"] > 0 and len(lines[0]) < 8000


def test_bench_py_c5_with_its_policy_runs_with_two_ranks_on_one_device():
    """The shape of a SCALE run of BASELINE configs[4]: `bench.py --gpus 2 --workload c5 --policy gaussian
    --policy-precision f16x2` (weak: 512 envs of 256 agents per rank, the Gaussian policy of utils.py:55-117 evaluated on
    the observation every step) as 2 ranks on one device: one line, `per_rank`, every launch accounted for, the N = 1
    figure of the same loop next to `value`."""
    r = _launch(2, "bench.py", ["--gpus", 2, "--workload", "c5", "--policy", "gaussian", "--policy-precision", "f16x2",
                                "--steps", 20, "--warmup", 2, "--min-seconds", 0.05, "--no-cpu-baseline"],
                extra_env=dict(BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1", BENCH_EAGER_MIN_STEPS="20", BENCH_KERNEL_SAMPLE_EPISODES="1",
                               BENCH_GRAPH_LAUNCHES="200"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8000
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["n_agents"] == 256
    assert out["config"]["envs_per_gpu"] == 512 and out["config"]["n_envs_total"] == 1024
    assert out["config"]["actions"] == "gaussian policy f16x2"
    assert out["value"] == pytest.approx(256 * 1024 * out["timed_steps"] / out["timed_seconds"], rel=1e-5)
    assert [q["rank"] for q in out["per_rank"]] == [0, 1] and all(q["envs"] == 512 and q["step_kernel_ms"] > 0 for q in out["per_rank"])
    lc = out["launch_check"]
    assert lc["ok"] is True and lc["env_steps_recorded_all_ranks"] == lc["env_steps_issued_all_ranks"] > 0
    assert out["n1"]["value"] > 0 and 0 < out["n1"]["efficiency"] <= 1.5
    assert out["exchange"]["real_collective_ran"] is True


def test_ragged_shards_over_eight_ranks_reproduce_one_rank(tmp_path):
    """99 envs over 8 ranks (shards of 13 / 12 envs): the product path per rank, bit-identical to one rank."""
    worker = os.path.join("tests", "launch_worker.py")
    many, one = tmp_path / "many", tmp_path / "one"
    many.mkdir(); one.mkdir()
    N, G, E, T, W = 5, 5.0, 99, 60, 8
    r = _launch(W, worker, [many, N, G, E, T], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, worker, str(one), str(N), str(G), str(E), str(T)], cwd=ROOT,
                       env=dict(os.environ, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    full = np.load(one / "rank0.npz")
    parts = [np.load(many / f"rank{k}.npz") for k in range(W)]
    sizes = [int(p["hi"]) - int(p["lo"]) for p in parts]
    assert sum(sizes) == E and max(sizes) - min(sizes) == 1 and int(parts[0]["lo"]) == 0 and int(parts[-1]["hi"]) == E
    for name in ("pos", "z", "nbr", "acc", "last_reward"):
        assert np.array_equal(np.concatenate([p[name] for p in parts]), full[name]), name
    for name in ("reward", "done"):
        assert np.array_equal(np.concatenate([p[name] for p in parts], axis=1), full[name]), name


def test_bench_py_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with NO launcher and NO WORLD_SIZE in the environment (how the driver's scaling run
    may call it): bench.py re-runs itself as 2 ranks, and the line is a 2-rank line -- never a silent one-rank run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "512",
                        "--min-seconds", "0.05", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["n_envs_total"] == 1024
    assert out["exchange"]["real_collective_ran"] is True and out["exchange"]["backend"] == "gloo"
    assert out["exchange"]["rccl_ranks"] == 0                  # gloo here; = world size under backend nccl
    assert out["exchange"]["collective_latency_us"] > 0
    assert out["launch_check"]["ok"] is True
    assert [q["rank"] for q in out["per_rank"]] == [0, 1]


def test_bench_py_refuses_more_gpus_than_the_box_has():
    """--gpus N beyond the visible devices (and no BENCH_ONE_DEVICE): non-zero exit and a message, no JSON line."""
    import torch
    n = torch.cuda.device_count() + 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "5", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "refusing" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_bench_py_strong_scaling_splits_one_job():
    """--scaling strong: the workload's 8-GPU job (8 x envs-per-gpu) is ONE job split over the ranks that run."""
    r = _launch(2, "bench.py", ["--gpus", 2, "--steps", 20, "--warmup", 2, "--envs-per-gpu", 64, "--min-seconds", 0.05,
                                "--no-cpu-baseline", "--scaling", "strong"], extra_env=dict(BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["n_envs_total"] == 512 and out["config"]["envs_per_gpu"] == 256
    assert out["value"] == pytest.approx(64 * 512 * out["timed_steps"] / out["timed_seconds"], rel=1e-5)
    assert out["launch_check"]["ok"] is True


def test_rccl_all_gather_in_a_world_of_one():
    """backend "nccl" (= RCCL on ROCm) initialised once on this GPU: the exchange's collective on the float64 vector."""
    import torch
    import torch.distributed as dist
    from scalable_collision_avoidance_rl_amd import drones
    from scalable_collision_avoidance_rl_amd.sharding import all_gather_stats, summarize_episodes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        env = drones(5, 0, [5, 5], "O", deltas=np.ones(5), simplify_zstate=True, n_envs=64, batched=True, device=dev,
                     seed=3, auto_reset=True)
        env.t.fill_(190)
        env.rollout_random(20)
        tot = env.episode_totals()
        g = all_gather_stats(tot, force_collective=True)       # all_gather_into_tensor over RCCL
        g2, work = all_gather_stats(tot, async_op=True, force_collective=True)
        work.wait(); torch.cuda.synchronize()
        assert g.is_cuda and g.shape == (1, 8) and torch.equal(g[0], tot) and torch.equal(g2[0], tot)
        s = summarize_episodes(g, 5)
        assert s["episodes"] == 64 and s["mean_episode_len"] == 10 and dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()
