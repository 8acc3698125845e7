#!/usr/bin/env python3
"""Benchmark of the drone_env hot path on MI355X:  env agent-steps/s = N * E * steps / s.

A "step" is one env.step() of the batched environment: one launch of the fused HIP kernel
over all E envs of this rank (integrate -> all-pairs distance -> Delta mask -> reward ->
k-nearest localized state -> done), every API output written, the per-episode statistic the
rollout loop logs (train_problem.py:98-100) accumulated on EVERY step by the kernel itself, and
envs whose episode ends (200 steps, drone_env.py:30) reset + re-observed inside the same launch
(train_problem.py:132).  Actions are synthetic U(-1,1)^2 (RandomAgent, SAC_agents.py:22),
pre-generated and resident in HBM before the timed region.

Timing: the K requested steps are captured as ONE hipGraph (whatever K is; a short request several times
over, ~4000 launches per graph, because a replay costs 0.1-0.3 ms of GPU idle time whatever it holds) and the
timed region replays that graph `repeats` times -- enough for >= 0.5 s -- between the two barriers, so a
short `--steps 20` run measures the same thing as a long one: value = N * E * timed_steps / elapsed and
ms_per_step = elapsed / timed_steps, timed_steps = K * graph_copies * repeats.  After every replay (every ~200
steps when a replay is shorter than an episode) the path's only exchange runs inside the timed region: one fixed-order reduction of the per-env episode records + one all-gather (RCCL).

Default workload = BASELINE.json configs[2] (N=64 x E=4096 per GPU, Delta=1.0, G=28): the
configuration the headline target (>= 1e7 agent-steps/s on 1 GPU) is quoted on; weak scaling:
every extra GPU adds another 4096 envs (configs[3] at 8 GPUs).  --workload c2|c5 select the
other single-GPU-sized configs.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 2000 --warmup 200
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_AGENT_STEP = 76        # reads pos 8 + act 8; writes pos 8, vel 8, r 4, true_r 4, z 24, nbr 12
BYTES_PER_ENV_STEP = 13          # n_coll 4 + done 1 + t read/write 8
BYTES_PER_ENV_STEP_RECORD = 64   # episode layer: the hot 32 bytes of the env's DroneEpisodeAcc record, read + written

WORKLOADS = {
    #        N    E/GPU  G      Delta  label
    "c2": (5, 1024, 5.0, 1.0, "C2: n=5 x 1024 envs/GPU, Delta=1.0, G=5, random actions"),
    "c3": (64, 4096, 28.0, 1.0, "C3: n=64 x 4096 envs/GPU, Delta=1.0, G=28, random actions"),
    "c5": (256, 512, 256.0, 2.5, "C5-shard: n=256 x 512 envs/GPU, Delta=2.5, G=256, random actions"),
}


def pmc_traffic(workload):
    """HBM-side bytes per step launch from the latest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of this workload (tools/pmc_run.py + tools/pmc_parse.py; counters cannot be read from inside
    the benchmarked process).  None when no such profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        return float(json.load(open(files[-1]))["traffic_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (OSError, KeyError, ValueError):
        return None, None


def reference_record(workload):
    """The reference's OWN `drones.step()` timing at this shape (tools/time_reference.py, measured in the build
    container where /root/reference exists; it cannot travel to the GPU box) -- attached beside the live figure
    of the C port so that both CPU numbers stand next to the GPU one."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu.json")))
        sh = rec["shapes"][workload]
        return {"kind": "reference", "source": "profiles/reference_cpu.json (tools/time_reference.py)",
                "host": rec["host"]["cpu"], "where": rec["host"]["where"],
                "one_core_agent_steps_per_s": sh["one_core"]["agent_steps_per_s"],
                "one_core_ms_per_step": sh["one_core"]["ms_per_step"],
                "whole_host_agent_steps_per_s": sh["whole_host"]["agent_steps_per_s"],
                "whole_host_cores": sh["whole_host"]["cores"]}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(N, G, delta, budget_s=12.0):
    """Oracle (C port of the reference path) timed on this host: a bounded sample of the same workload."""
    from oracle.oracle import Oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                            # honour a cgroup CPU quota if the box sets one
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    orc = Oracle(N, [G, G], 2, np.ones(N) * delta, True, threads=cores)
    E = max(cores * 16, 64)
    pos, vel, t, _, _ = orc.reset(E, 1234)
    rng = np.random.default_rng(0)
    act = rng.uniform(-1, 1, (E, N, 2))
    orc.step(pos, vel, t, act)                      # warm-up
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for _ in range(5):
            orc.step(pos, vel, t, act)
        steps += 5
    el = time.perf_counter() - t0
    return {"value": N * E * steps / el, "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/drone_oracle.c (float64 C restatement), {E} envs x {steps} steps, "
                      f"{cores} OpenMP threads, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of hipGraph replay")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="minimum length of the timed region (graph replays)")
    ap.add_argument("--no-episode-layer", action="store_true",
                    help="plain dronesim_step launches (no per-step statistic, explicit reset kernel every 200 steps)")
    ap.add_argument("--policy", default="random", choices=["random", "softmax16", "gaussian"],
                    help="action source: pre-generated U(-1,1) actions (the graded workload) or a batched per-agent "
                         "policy evaluated on the env's observation every step (BASELINE configs[4] uses 'gaussian')")
    ap.add_argument("--policy-precision", default="f32", choices=["f32", "bf16x3", "f16x2", "bf16"],
                    help="matrix-core arithmetic of the batched policy (f32 = exact; bf16x3 / f16x2 = three-part bf16 / "
                         "two-part float16 split, float32-accurate; bf16 = opt-in fast path, ~1e-2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from scalable_collision_avoidance_rl_amd import drones, max_time_steps
    from scalable_collision_avoidance_rl_amd.sharding import reduce_episode_records, summarize_episodes, all_gather_stats

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_BACKEND=gloo BENCH_ONE_DEVICE=1 lets the launcher path (ranks, barrier, max-over-ranks, gather) be
    # exercised with several ranks on a single GPU; the graded runs use RCCL with one GPU per rank
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    N, e_gpu, G, delta, label = WORKLOADS[args.workload]
    if args.envs_per_gpu:
        e_gpu = args.envs_per_gpu
    E_global = e_gpu * world
    layer = not args.no_episode_layer
    env = drones(N, 0, [G, G], "O", k_closest=2, deltas=np.ones(N) * delta, simplify_zstate=True,
                 n_envs=E_global, device=dev, seed=1234, rank=rank, world_size=world, batched=True,
                 auto_reset=layer, track_episodes=True)
    E = env.n_envs
    T_ep = max_time_steps

    # synthetic actions, resident in HBM: one episode's worth, reused every episode
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = torch.rand(T_ep, E, N, 2, device=dev, generator=g) * 2 - 1

    policy = None
    if args.policy != "random":                     # random-init per-agent networks of the reference's shapes
        from scalable_collision_avoidance_rl_amd.policies import BatchedMLP
        gp = torch.Generator().manual_seed(4321)
        rw = lambda *sh: (torch.rand(*sh, generator=gp) * 2 - 1) * 0.2
        h, nout, ok, sk = (300, 16, 1, 1) if args.policy == "softmax16" else (400, 4, 2, 2)   # utils.py:255-302 / 55-108
        policy = BatchedMLP(rw(N, 6, h), rw(N, h), rw(N, h, h), rw(N, h), rw(N, h, nout), rw(N, nout), ok, sk,
                            device=dev, seed=1234, precision=args.policy_precision)

    def one_step(s):
        if policy is None:
            env.step(pool[s % T_ep])
        else:                                       # obs -> sample_action -> step (SAC_agents.py:170-180, train_problem.py:91-94)
            act, _ = policy.sample_action(env.z, env=env)
            env.step(act)
        if not layer and (s + 1) % T_ep == 0:       # plain path: explicit reset kernel + observe (train_problem.py:132)
            env.reset(renew_obstacles=False)

    # the path's only exchange (train_problem.py:118-121: what is logged per episode), off the per-step path:
    # local fixed-order reduction of the episode records on the step stream, then ONE all-gather of 8 doubles
    exchanges = []

    def exchange():
        tot = env.episode_totals()                  # one launch on the current stream; device tensor [8]
        exchanges.append(all_gather_stats(tot, async_op=True))   # RCCL on its own stream; the rollout is not held up

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_no = 0
    for _ in range(args.warmup):
        one_step(step_no); step_no += 1
    torch.cuda.synchronize()

    # secondary figure: the same K steps launched eagerly from Python (host launch latency included)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(step_no); step_no += 1
    barrier()
    eager_elapsed = time.perf_counter() - t0

    # the K steps as ONE hipGraph.  All episode state (t, episode counters, records, RNG stream position) lives on
    # the device, so every replay continues the rollout: episodes end, are logged and restart inside the replays.
    graph = None
    K = args.steps
    if not args.no_graph:
        if not layer:
            while step_no % T_ep:                   # plain path resets by step index: align the capture to an episode
                one_step(step_no); step_no += 1
        torch.cuda.synchronize()
        # a replay costs ~0.1-0.3 ms of GPU idle time whatever the graph holds (tools/replay_probe.py: 6.01 us per step
        # with 250 launches per graph, 5.73 with 1000, 5.62 with 4000): the request is captured several times over into
        # ONE graph of ~4000 launches (the episode layer keeps every counter on the device, so the copies simply
        # continue the rollout), and the per-step figure of `--steps 20` is that of `--steps 2000`
        copies = max(1, -(-4000 // K)) if layer else 1
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for s in range(K * copies):
                one_step(s)
        for _ in range(2):                          # untimed replays: instantiate + clocks
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize()
        one = max(time.perf_counter() - t0, 1e-6)
        if world > 1:                               # every rank must replay (and exchange) the same number of times
            o = torch.tensor([one], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(o, op=dist.ReduceOp.MIN)
            one = float(o.item())
        repeats = max(1, int(np.ceil(1.2 * args.min_seconds / one)))
    else:
        repeats, copies = 1, 1
    L = K * copies                                       # launches per replay
    log_every = max(1, T_ep // L) if L < T_ep else 1     # replays between two exchanges (~ once per episode)

    barrier()
    t0 = time.perf_counter()
    if graph is not None:
        for r in range(repeats):
            graph.replay()
            if (r + 1) % log_every == 0:
                exchange()
    else:
        for s in range(K):
            one_step(step_no); step_no += 1
        exchange()
    for _, work in exchanges:                       # every exchange of the timed region has completed
        if work is not None:
            work.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    total_steps = L * repeats
    # final exchange: global per-episode figures (the same reduction + all-gather as inside the timed region)
    summary = reduce_episode_records(env)
    summary["exchanges_in_timed_region"] = len(exchanges)
    last = exchanges[-1][0].to(dev) if exchanges else None
    summary["last_timed_exchange_episodes"] = None if last is None else float(last.double().sum(0)[4])

    # Duration of the dominant kernel (drone_kernel<step>) per launch, by HIP events on the launch stream
    # (torch's current stream = the stream handed to dronesim_step): events bracket a hipGraph holding
    # ONLY `n_samp` back-to-back step launches, so (t1 - t0) / n_samp is the kernel's duration including the
    # ~0.2 us dependent-launch boundary and excluding host launch latency.  rocprofv3 --kernel-trace --stats
    # of this command reports the same kernel's average duration (profiles/).
    # With the episode layer the graph holds twenty episodes (the in-kernel resets fire inside it, as in the timed
    # region); the ~0.1 ms a graph replay costs is then < 0.5 % of the bracketed time.
    n_samp = 20 * T_ep if layer else T_ep
    kgraph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(kgraph):
        for s in range(n_samp):
            env.step(pool[s % T_ep])
    kgraph.replay(); torch.cuda.synchronize()
    samples = []
    for _ in range(10):
        env.reset(renew_obstacles=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); kgraph.replay(); e1.record(); torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) / n_samp)
    kern_ms = float(np.median(samples))

    # secondary figure: the same workload through dronesim_rollout (200 steps fused in ONE launch, actions
    # known up front -- RandomAgent rollouts); every per-step output except the per-step state is written
    ro_us = None
    try:
        env.reset(renew_obstacles=False)
        out = env.rollout(pool); torch.cuda.synchronize(); del out
        rs = []
        for _ in range(3):
            env.reset(renew_obstacles=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = env.rollout(pool); e1.record(); torch.cuda.synchronize(); del out
            rs.append(e0.elapsed_time(e1) * 1e3 / T_ep)
        ro_us = float(np.median(rs))
    except RuntimeError:                               # e.g. not enough memory for the [T, ...] outputs
        ro_us = None
    # and with the actions drawn inside the kernel (dronesim_rollout_random: RandomAgent.forward, SAC_agents.py:22,
    # from a counter-based stream; no action pool, 44 B/agent-step), episode layer on: runs across episode ends
    rr_us = None
    try:
        env.reset(renew_obstacles=False)
        out = env.rollout_random(T_ep); torch.cuda.synchronize(); del out
        rs = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = env.rollout_random(T_ep); e1.record(); torch.cuda.synchronize(); del out
            rs.append(e0.elapsed_time(e1) * 1e3 / T_ep)
        rr_us = float(np.median(rs))
    except RuntimeError:
        rr_us = None

    el = torch.tensor([elapsed, eager_elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed, eager_elapsed = float(el[0].item()), float(el[1].item())

    if rank == 0:
        agent_steps = N * E_global * total_steps
        value = agent_steps / elapsed
        bytes_launch = BYTES_PER_AGENT_STEP * N * E + (BYTES_PER_ENV_STEP + (BYTES_PER_ENV_STEP_RECORD if layer else 0)) * E
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload) if e_gpu == WORKLOADS[args.workload][1] else (None, None)
        out = {
            "metric": "env agent-steps/sec (n_agents x n_envs x steps)",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "repeats": repeats, "graph_copies": copies, "timed_steps": total_steps, "timed_seconds": elapsed,
            "ms_per_step": elapsed / total_steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "n_agents": N, "envs_per_gpu": e_gpu, "n_envs_total": E_global,
                       "grid": G, "delta": delta, "k_closest": 2, "simplify_zstate": True,
                       "launch": ((f"hipGraph of the {K} requested steps" + (f", captured {copies}x over ({L} launches)" if copies > 1 else "") +
                                   f", replayed {repeats}x in the timed region") if graph is not None else "eager"),
                       "episode_layer": ("per-step statistic + in-kernel auto-reset (dronesim_step_ex)" if layer else
                                         "plain dronesim_step + reset kernel every 200 steps"),
                       "actions": "pre-generated U(-1,1)^2, resident in HBM" if policy is None else
                                  f"batched per-agent {args.policy} policy (random-init, {args.policy_precision} MFMA) on the observation",
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "drone_kernel<K=2,FAR=0,step,%s>" % ("episode layer" if layer else "plain"), "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": bytes_launch},
            "episode_end_stats": summary,
            "eager": {"value": N * E_global * args.steps / eager_elapsed, "ms_per_step": eager_elapsed / args.steps * 1e3,
                      "note": "the same K steps launched one by one from Python (host launch latency included)"},
            "fused_rollout": None if ro_us is None else {
                "us_per_step_per_gpu": ro_us, "agent_steps_per_s_per_gpu": N * E / ro_us * 1e6,
                "note": "dronesim_rollout: 200 steps per launch, 52 B/agent-step (no per-step state write-back)",
                "random_actions_in_kernel": None if rr_us is None else {
                    "us_per_step_per_gpu": rr_us, "agent_steps_per_s_per_gpu": N * E / rr_us * 1e6,
                    "note": "dronesim_rollout_random: actions drawn in the kernel (no action pool, 44 B/agent-step), "
                            "episode records + in-kernel reset on"}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(N, G, delta, args.cpu_budget)
            out["cpu_baseline"]["reference"] = reference_record(args.workload)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
