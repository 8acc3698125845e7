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
ms_per_step = elapsed / timed_steps, timed_steps = K * graph_copies * repeats.

The path's only exchange (train_problem.py:118-121: the per-episode log) runs inside the timed region at the
reference's cadence: every 200 steps (one episode, drone_env.py:30) the capture holds one fixed-order reduction of
the per-env episode records into the next slot of a small device ring, and after every replay ONE all-gather (RCCL
over xGMI when world_size > 1) ships the ring's slots.  The line's `exchange` block says how many reductions and
collectives the timed region held and whether a real collective ran.

Default workload = BASELINE.json configs[2] (N=64 x E=4096 per GPU, Delta=1.0, G=28): the
configuration the headline target (>= 1e7 agent-steps/s on 1 GPU) is quoted on.  --scaling weak (default):
every extra GPU adds another 4096 envs (configs[3] at 8 GPUs); --scaling strong: configs[3]'s 32768 envs are ONE
job split over the ranks.  --workload c2|c5 select the other single-GPU-sized configs; the default run also times
them (c2, the c5 shard, the c5 shard with the float32 Gaussian policy in the loop) after the graded region and
appends them as `other_workloads`.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python bench.py --gpus 8 --steps 2000 --warmup 200        # starts its own 8 ranks (torch.distributed.run)
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
# launcher tests with many ranks on ONE device shrink the secondary measurements (the graded region is untouched by these):
EAGER_MIN_STEPS = int(os.environ.get("BENCH_EAGER_MIN_STEPS", "1000"))      # steps of the eager side figure
KERNEL_SAMPLE_EPISODES = int(os.environ.get("BENCH_KERNEL_SAMPLE_EPISODES", "20"))   # episodes in the step-kernel timing graph
GRAPH_LAUNCHES = int(os.environ.get("BENCH_GRAPH_LAUNCHES", "4000"))   # launches captured into the one replayed graph
#                                  (profiles/r4_graph_size_probe.log: 2000 -> 5.20 us/step, 4000 -> 5.14, 16000 -> 5.31)

WORKLOADS = {
    #        N    E/GPU  G      Delta  label
    "c2": (5, 1024, 5.0, 1.0, "C2: n=5 x 1024 envs/GPU, Delta=1.0, G=5, random actions"),
    "c3": (64, 4096, 28.0, 1.0, "C3: n=64 x 4096 envs/GPU, Delta=1.0, G=28, random actions"),
    "c5": (256, 512, 256.0, 2.5, "C5-shard: n=256 x 512 envs/GPU, Delta=2.5, G=256, random actions"),
}
# side lines only (`other_workloads`): the shapes that carry the "boundary amortised" part of the roofline argument
SIDE_WORKLOADS = {
    "c3_8192": (64, 8192, 28.0, 1.0, "n=64 x 8192 envs on one GPU (2 x C3), Delta=1.0, G=28, random actions"),
    "c4_one_gpu": (64, 32768, 28.0, 1.0, "C4 as ONE job on one GPU: n=64 x 32768 envs (BASELINE configs[3] on one rank), Delta=1.0, G=28, random actions"),
    "c5_full": (256, 4096, 256.0, 2.5, "C5 as ONE job on one GPU: n=256 x 4096 envs (BASELINE configs[4]'s env axis on one rank), Delta=2.5, G=256, random actions"),
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


def make_policy(torch, kind, precision, N, dev):
    """Random-init per-agent networks of the reference's shapes (utils.py:255-302 / 55-108)."""
    from scalable_collision_avoidance_rl_amd.policies import BatchedMLP
    gp = torch.Generator().manual_seed(4321)
    rw = lambda *sh: (torch.rand(*sh, generator=gp) * 2 - 1) * 0.2
    h, nout, ok, sk = (300, 16, 1, 1) if kind == "softmax16" else (400, 4, 2, 2)
    return BatchedMLP(rw(N, 6, h), rw(N, h), rw(N, h, h), rw(N, h), rw(N, h, nout), rw(N, nout), ok, sk,
                      device=dev, seed=1234, precision=precision)


def algorithmic_bytes(N, E, layer):
    return BYTES_PER_AGENT_STEP * N * E + (BYTES_PER_ENV_STEP + (BYTES_PER_ENV_STEP_RECORD if layer else 0)) * E


def step_kernel_ms(torch, env, pool, n_samp, reps=10):
    """Duration of the step kernel per launch by HIP events on the launch stream (torch's current stream = the stream
    handed to dronesim_step): events bracket a hipGraph holding ONLY `n_samp` back-to-back step launches."""
    T_ep = pool.shape[0]
    kgraph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(kgraph):
        for s in range(n_samp):
            env.step(pool[s % T_ep])
    kgraph.replay(); torch.cuda.synchronize()
    samples = []
    for _ in range(reps):
        env.reset(renew_obstacles=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); kgraph.replay(); e1.record(); torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) / n_samp)
    return float(np.median(samples))


def side_workload(torch, dev, name, policy_kind=None, min_seconds=0.25, precision="f32", default_construction=False):
    """One of the other BASELINE configs, timed in this process after the graded region: the same step path
    (episode layer, hipGraph replay), single GPU.  Returns the block appended under `other_workloads`.
    ``default_construction``: the reference's own defaults `drones(n, 0, grid, "O")` -- deltas=None (Delta = d_hat),
    simplify_zstate=False (15-column observation: 112 B per agent-step) -- drone_env.py:55, 85-87, 184."""
    from scalable_collision_avoidance_rl_amd import drones, max_time_steps
    N, E, G, delta, label = WORKLOADS[name] if name in WORKLOADS else SIDE_WORKLOADS[name]
    if default_construction:
        env = drones(N, 0, [G, G], "O", n_envs=E, device=dev, seed=1234, batched=True, auto_reset=True, track_episodes=True)
        label = (f"reference defaults drones({N}, 0, [{G:g}, {G:g}], 'O'): deltas=None, simplify_zstate=False, k_closest=2; "
                 f"{E} envs/GPU, random actions")
    else:
        env = drones(N, 0, [G, G], "O", k_closest=2, deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, device=dev,
                     seed=1234, batched=True, auto_reset=True, track_episodes=True)
    T_ep = max_time_steps
    g = torch.Generator(device=dev).manual_seed(99)
    pool = torch.rand(T_ep, E, N, 2, device=dev, generator=g) * 2 - 1
    policy = make_policy(torch, policy_kind, precision, N, dev) if policy_kind else None

    def one_step(s):
        if policy is None:
            env.step(pool[s % T_ep])
        else:
            act, _ = policy.sample_action(env.z, env=env)
            env.step(act)

    for s in range(10):
        one_step(s)
    torch.cuda.synchronize()
    L = 2000 if policy is None else 100                # launches per graph (a policy step is ~100x an env step)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for s in range(L):
            one_step(s)
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-6)
    repeats = max(1, int(np.ceil(min_seconds / one)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(repeats):
        graph.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    steps = L * repeats
    kern_ms = step_kernel_ms(torch, env, pool, 5 * T_ep, reps=5)
    byt = algorithmic_bytes(N, E, True) + (36 * N * E if default_construction else 0)      # c = 5 rows: 60 B of z instead of 24
    cfg_note = "BASELINE configs[4], one shard" if (name == "c5" and policy_kind == "gaussian") else f"the reference's {policy_kind} networks (utils.py:255-309 / 55-117) at this shape"
    arith = {"f32": "exact float32 (v_mfma_f32_32x32x2_f32)", "bf16x3": "float32-accurate three-part bf16 split (6 x v_mfma_f32_32x32x16_bf16 per 16 k)",
             "f16x2": "float32-accurate two-part float16 split (3 x v_mfma_f32_32x32x16_f16 per 16 k; |activations| < 65504)"}.get(precision, precision)
    out = {"workload": label + (f" + {policy_kind} policy in the loop, {arith} ({cfg_note})" if policy else ""),
           "value": N * E * steps / el, "unit": "agent-steps/s", "ms_per_step": el / steps * 1e3, "timed_steps": steps,
           "timed_seconds": el, "step_kernel_ms": kern_ms,
           "roofline": {"bound": "hbm", "achieved": byt / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": byt / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byt,
                        "kernel": "drone_kernel<K=2,FAR=%d,step,episode layer>" % (1 if default_construction else 0)}}
    if policy is not None:                             # the policy kernel alone (exact-f32 MFMA), same events-around-a-graph method
        pg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(pg):
            for _ in range(20):
                policy.sample_action(env.z, env=env)
        pg.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):                             # median of five replays (one alone still carries the clock ramp:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # it came out ABOVE the
            e0.record(); pg.replay(); e1.record(); torch.cuda.synchronize()                       # in-the-loop step time)
            ts.append(e0.elapsed_time(e1) / 20)
        pol_ms = float(np.median(ts))
        flops = 2.0 * E * N * (policy.d_in * policy.h1 + policy.h1 * policy.h2 + policy.h2 * policy.nout)
        out["policy_kernel_ms"] = pol_ms
        if precision == "f32":
            out["policy_roofline"] = {"bound": "mfma", "achieved": flops / (pol_ms * 1e-3) / 1e12, "peak": 157.3,
                                      "unit": "TFLOP/s", "frac": flops / (pol_ms * 1e-3) / 1e12 / 157.3,
                                      "kernel": "mlp3_kernel (v_mfma_f32_32x32x2_f32, layer 3 on v_mfma_f32_16x16x4_f32; exact float32)"}
        else:                                          # six bf16 (three f16) products per float32-equivalent product on the 2.5 PFLOP/s pipe
            nprod = 6.0 if precision == "bf16x3" else 3.0
            mf = nprod * flops
            out["policy_roofline"] = {"bound": "mfma", "achieved": mf / (pol_ms * 1e-3) / 1e12, "peak": 2500.0,
                                      "unit": "TFLOP/s", "frac": mf / (pol_ms * 1e-3) / 1e12 / 2500.0,
                                      "float32_equivalent_tflops": flops / (pol_ms * 1e-3) / 1e12,
                                      "kernel": "mlp3_split_kernel<%s> (16-bit matrix flops actually issued: %d per float32 product)"
                                                % ("SchemeBf16x3" if precision == "bf16x3" else "SchemeF16x2", int(nprod))}
    del graph, env, pool
    torch.cuda.empty_cache()
    return out


def graph_time_us(torch, fn, calls, reps=5):
    """Median duration of one `fn()` call in microseconds: HIP events (on torch's current stream = the stream the library
    launches on) around a hipGraph that holds `calls` back-to-back calls."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / calls)
    del g
    return float(np.median(ts))


def aux_kernels(torch, dev):
    """The kernels around the step launch, at the C3 rollout shape (T = 200 steps x 4096 envs x 64 agents), each with its
    own HBM roofline on ALGORITHMIC bytes: the learner-side scans over a stored rollout (SAC_agents.py:304-307, 333-351),
    env.reset() = reset kernel + first observation (drone_env.py:98-102, 171-212) and the classical controllers
    (drone_env.py:609-679)."""
    from scalable_collision_avoidance_rl_amd import drones
    from scalable_collision_avoidance_rl_amd.rollout_buffer import mc_returns, neighbour_advantage
    T, E, N = 200, 4096, 64
    out = {}

    def line(what, us, byt, kernel, note):
        gbs = byt / (us * 1e-6) / 1e9
        return {"workload": what, "us_per_call": us, "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                                 "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byt,
                                                                 "kernel": kernel}, "note": note}
    g = torch.Generator(device=dev).manual_seed(5)
    r = torch.randn(T, E, N, device=dev, generator=g)
    done = torch.zeros(T, E, dtype=torch.uint8, device=dev); done[-1] = 1
    done[torch.randint(0, T, (E,), device=dev, generator=g), torch.arange(E, device=dev)] = 1   # one episode end per env somewhere
    V = torch.randn(T, E, N, device=dev, generator=g)
    nbr = torch.randint(-1, N, (T, E, N, 3), device=dev, dtype=torch.int32, generator=g)
    nbr[..., 0] = torch.arange(N, device=dev, dtype=torch.int32)
    us = graph_time_us(torch, lambda: mc_returns(r, 0.97, done), 8)
    out["returns"] = line(f"Monte-Carlo returns over a stored rollout [T={T}, E={E}, N={N}] with episode ends (SAC_agents.py:304-307)",
                          us, T * E * N * 8 + T * E, "returns_kernel", "reads reward 4 B + writes G 4 B per (step, agent), done 1 B per (step, env)")
    G = mc_returns(r, 0.97, done)
    us = graph_time_us(torch, lambda: neighbour_advantage(G, V, nbr, 0.97, done), 8)
    out["advantage"] = line(f"neighbour-summed advantage weights [T={T}, E={E}, N={N}, k+1=3] (SAC_agents.py:333-351)",
                            us, T * E * N * 24 + T * E, "advantage_kernel<3>",
                            "reads G 4 + V 4 + nbr_idx 12 B, writes w 4 B per (step, agent); the neighbour gathers of G fall into the row just read (cache)")
    del r, V, nbr, G, done
    torch.cuda.empty_cache()
    env = drones(N, 0, [28.0, 28.0], "O", k_closest=2, deltas=np.ones(N), simplify_zstate=True, n_envs=E, device=dev, seed=1234,
                 batched=True)
    us = graph_time_us(torch, lambda: env.reset(renew_obstacles=False), 20)
    out["reset"] = line(f"env.reset() of all {E} envs of C3: lattice draw without replacement (reset_kernel) + first observation "
                        "(drone_kernel<observe>) (drone_env.py:98-102, 171-212)", us, E * N * (16 + 16 + 36) + E * 12,
                        "reset_kernel + drone_kernel<K=2,FAR=0,observe,plain>",
                        "two launches: writes pos 8 + vel 8; reads them back 16, writes z 24 + nbr_idx 12 per agent; t / episode per env")
    for kind, lines in (("proportional", "drone_env.py:652-679"), ("gradient", "drone_env.py:609-650")):
        us = graph_time_us(torch, lambda: env.control(kind), 40)
        out[f"control_{kind}"] = line(f"{kind}_control for all agents of C3 ({lines})", us, E * N * 16,
                                      "control_kernel", "reads pos 8 B, writes act 8 B per agent: a 4 MB launch (launch-floor-bound)")
    del env
    torch.cuda.empty_cache()
    return out


RCCL_PROBE = r"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29653")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
src = torch.arange(8, dtype=torch.float64, device="cuda").view(1, 8); out = torch.empty(1, 8, dtype=torch.float64, device="cuda")
for _ in range(20): dist.all_gather_into_tensor(out, src)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): dist.all_gather_into_tensor(out, src)
torch.cuda.synchronize(); print("RCCL_US", (time.perf_counter() - t0) / 200 * 1e6, bool(torch.equal(out, src)))
dist.destroy_process_group()
"""


def rccl_probe():
    """Latency of the exchange's collective through RCCL in a world of ONE (a separate process with its own 60 s
    limit, so that a stuck rendezvous cannot take the benchmark with it).  None when it did not run."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", RCCL_PROBE], capture_output=True, text=True, timeout=90)
        for line in r.stdout.splitlines():
            if line.startswith("RCCL_US"):
                _, us, ok = line.split()
                return {"world_size": 1, "all_gather_into_tensor_8_doubles_us": float(us), "result_correct": ok == "True",
                        "note": "RCCL collective of the exchange, timed back to back in a world of one (separate process)"}
    except (subprocess.SubprocessError, OSError, ValueError):
        pass
    return None


def visible_gpus():
    """Number of GPUs this process could use, WITHOUT creating a HIP context in the parent of the ranks."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except (subprocess.SubprocessError, OSError, ValueError, IndexError):
        return 0


def spawn_ranks(n):
    """Re-run this command as `n` ranks of one node through torch.distributed.run (one process per GPU, RCCL over xGMI
    under backend "nccl"; rendezvous on 127.0.0.1 and a free port).  Returns the launcher's exit code.  With fewer than
    `n` visible GPUs the run is REFUSED (exit code 2) unless BENCH_ONE_DEVICE=1 asks for all ranks on device 0 (the
    launcher tests on a one-GPU box, backend gloo)."""
    import socket
    import subprocess
    if not os.environ.get("BENCH_ONE_DEVICE"):
        have = visible_gpus()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} GPUs, {have} visible: refusing to run (a one-rank run must not be "
                  f"reported as a {n}-GPU figure; BENCH_ONE_DEVICE=1 BENCH_BACKEND=gloo puts all ranks on device 0 "
                  "for launcher tests)", file=sys.stderr)
            return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_span(workload):
    """The step launches as the DEVICE sees them: first wave in -> last wave out (`span_us`) and last wave out -> the next
    launch's first wave (`boundary_us`) inside a hipGraph replay, from the waves' own entry / exit times on the chip-wide
    100 MHz clock.  Measured in a separate process on the -DDRONESIM_TRACE_SPAN build of the same source
    (libdronesim_span.so: product code + one s_memrealtime pair and one 8-byte store per wave; ~2 % slower), so that the
    benchmarked process keeps the product library.  None when that build is not there."""
    import subprocess
    lib = os.path.join(ROOT, "scalable_collision_avoidance_rl_amd", "libdronesim_span.so")
    if not os.path.exists(lib):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_span.py"), workload, "64"], capture_output=True,
                           text=True, timeout=240, env=dict(os.environ, DRONESIM_LIB=lib))
        for line in r.stdout.splitlines():
            if line.startswith("SPAN_JSON "):
                d = json.loads(line[10:])
                d["how"] = ("tools/trace_span.py on libdronesim_span.so (-DDRONESIM_TRACE_SPAN): 64 traced step launches in one "
                            "hipGraph, medians; 10 ns clock")
                return d
    except (subprocess.SubprocessError, OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: envs-per-gpu fixed (4096 at c3: configs[3] at 8 GPUs); strong: the c3 shape's 8-GPU job "
                         "(32768 envs = BASELINE configs[3]) is ONE job split over however many ranks run")
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
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the c2 / c5 side measurements appended to the default single-GPU run")
    ap.add_argument("--no-rccl-probe", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it (how the driver's scaling run may call it): start the N
        # ranks ourselves, one process per GPU, and let rank 0 print the line.  Never a silent one-rank run.
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from scalable_collision_avoidance_rl_amd import drones, max_time_steps
    from scalable_collision_avoidance_rl_amd.sharding import summarize_episodes, all_gather_stats

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_BACKEND=gloo BENCH_ONE_DEVICE=1 lets the launcher path (ranks, barrier, max-over-ranks, gather) be
    # exercised with several ranks on a single GPU; the graded runs use RCCL with one GPU per rank
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_ONE_DEVICE"):
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU (local rank {local_rank}, {torch.cuda.device_count()} visible device(s))",
              file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)", file=sys.stderr)
        sys.exit(2)
    cdev = dev if backend == "nccl" else "cpu"           # where small control tensors of the collectives live

    N, e_gpu, G, delta, label = WORKLOADS[args.workload]
    if args.envs_per_gpu:
        e_gpu = args.envs_per_gpu
    if args.scaling == "strong":
        E_global = e_gpu * 8                             # the 8-GPU job of the workload (c3: configs[3], 32768 envs)
        label += f" -- strong scaling: {E_global} envs in all (BASELINE configs[3] when c3), split over {world} rank(s)"
    else:
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

    policy = make_policy(torch, args.policy, args.policy_precision, N, dev) if args.policy != "random" else None
    launches = 0                                         # step launches issued on this rank (checked against the device records)

    def one_step(s):
        nonlocal launches
        if policy is None:
            env.step(pool[s % T_ep])
        else:                                       # obs -> sample_action -> step (SAC_agents.py:170-180, train_problem.py:91-94)
            act, _ = policy.sample_action(env.z, env=env)
            env.step(act)
        launches += 1
        if not layer and (s + 1) % T_ep == 0:       # plain path: explicit reset kernel + observe (train_problem.py:132)
            env.reset(renew_obstacles=False)

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
    eager_steps = max(args.steps, EAGER_MIN_STEPS) if policy is None else args.steps     # (a 20-step loop would mostly time its two syncs)
    barrier()
    t0 = time.perf_counter()
    for s in range(eager_steps):
        one_step(step_no); step_no += 1
    barrier()
    eager_elapsed = time.perf_counter() - t0

    # the K steps as ONE hipGraph.  All episode state (t, episode counters, records, RNG stream position) lives on
    # the device, so every replay continues the rollout: episodes end, are logged and restart inside the replays.
    graph = None
    K = args.steps
    ring = None                                          # [slots, 8] float64: one in-graph reduction per episode length
    if not args.no_graph:
        if not layer:
            while step_no % T_ep:                   # plain path resets by step index: align the capture to an episode
                one_step(step_no); step_no += 1
        torch.cuda.synchronize()
        # a replay costs ~0.1-0.3 ms of GPU idle time whatever the graph holds (tools/replay_probe.py: 6.01 us per step
        # with 250 launches per graph, 5.73 with 1000, 5.62 with 4000): the request is captured several times over into
        # ONE graph of ~4000 launches (the episode layer keeps every counter on the device, so the copies simply
        # continue the rollout), and the per-step figure of `--steps 20` is that of `--steps 2000`
        copies = max(1, -(-GRAPH_LAUNCHES // K)) if layer else 1
        L = K * copies
        # the exchange's local half at the reference's cadence, INSIDE the capture: after every T_ep-th launch one
        # fixed-order reduction of the episode records (train_problem.py:118-121 logs once per episode) into its own
        # ring slot; a capture shorter than an episode gets one at its end
        slots = max(1, L // T_ep)
        ring = torch.zeros(slots, 8, dtype=torch.float64, device=dev)
        launches_before = launches
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for s in range(L):
                one_step(s)
                if (s + 1) % T_ep == 0 and (s + 1) // T_ep <= slots and L >= T_ep:
                    env.episode_totals(out=ring[(s + 1) // T_ep - 1])
            if L < T_ep:
                env.episode_totals(out=ring[0])
        launches = launches_before                       # capturing enqueues nothing
        for _ in range(2):                          # untimed replays: instantiate + clocks
            graph.replay(); launches += L
        torch.cuda.synchronize()
        t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize(); launches += L
        one = max(time.perf_counter() - t0, 1e-6)
        if world > 1:                               # every rank must replay (and exchange) the same number of times
            o = torch.tensor([one], dtype=torch.float64, device=cdev)
            dist.all_reduce(o, op=dist.ReduceOp.MIN)
            one = float(o.item())
        repeats = max(1, int(np.ceil(1.2 * args.min_seconds / one)))
    else:
        repeats, copies, L, slots = 1, 1, K, 1
        ring = torch.zeros(1, 8, dtype=torch.float64, device=dev)

    # the exchange's collective half: ONE all-gather per replay of the ring's slots (64 B per slot and rank), issued
    # async on RCCL's own stream behind the replay that filled them; the rollout's next replay is not held up
    exchanges = []

    def exchange():
        exchanges.append(all_gather_stats(ring.view(-1), async_op=True))

    barrier()
    t0 = time.perf_counter()
    if graph is not None:
        for r in range(repeats):
            graph.replay(); launches += L
            exchange()
    else:
        for s in range(K):
            one_step(step_no); step_no += 1
            if (s + 1) % T_ep == 0 or s == K - 1:
                env.episode_totals(out=ring[0]); exchange()
    for _, work in exchanges:                       # every exchange of the timed region has completed
        if work is not None:
            work.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    total_steps = L * repeats
    reductions = (slots * repeats) if graph is not None else len(exchanges)
    # final exchange: global per-episode figures from one more reduction + all-gather of the same kind
    final = all_gather_stats(env.episode_totals())
    summary = summarize_episodes(final, N)
    last = exchanges[-1][0].to(dev).view(-1, slots, 8) if exchanges else None      # [world, slots, 8]
    summary["last_timed_exchange_episodes"] = None if last is None else float(last[:, -1, 4].double().sum())
    # the device-side records must account for every launch this process issued: env_steps = launches x envs
    if layer:
        own = env.episode_totals().cpu()
        recorded = float(own[3] + own[7])
        assert recorded == float(launches) * E, f"episode records hold {recorded} env-steps, {launches} launches x {E} envs issued"
    launch_check = {"launches_issued_per_rank": launches, "envs_per_rank": E,
                    "env_steps_recorded_all_ranks": summary["env_steps"],
                    "ok": (not layer) or summary["env_steps"] == float(launches) * E_global}

    # measured latency of the exchange's collective on this job's process group (world > 1), outside the timed region
    allgather_us = None
    if world > 1:
        src = ring.view(-1)
        for _ in range(5):
            all_gather_stats(src)
        barrier(); t1 = time.perf_counter()
        for _ in range(50):
            all_gather_stats(src)
        torch.cuda.synchronize()
        allgather_us = (time.perf_counter() - t1) / 50 * 1e6

    # Duration of the dominant kernel (drone_kernel<step>) per launch, by HIP events on the launch stream: events
    # bracket a hipGraph holding ONLY `n_samp` back-to-back step launches, so (t1 - t0) / n_samp is the kernel's
    # duration including the dependent-launch boundary and excluding host launch latency.  rocprofv3 --kernel-trace
    # --stats of this command reports the same kernel's average duration (profiles/).  With the episode layer the graph
    # holds twenty episodes (the in-kernel resets fire inside it, as in the timed region).
    kern_ms = step_kernel_ms(torch, env, pool, KERNEL_SAMPLE_EPISODES * T_ep if layer else T_ep)

    # secondary figure: the same workload through dronesim_rollout (200 steps fused in ONE launch, actions
    # known up front -- RandomAgent rollouts); every per-step output except the per-step state is written
    ro_us = None
    try:
        env.reset(renew_obstacles=False)
        out = env.rollout(pool); torch.cuda.synchronize(); del out
        # (four launches per bracket: the host-side cost of a call -- output allocation, argument marshalling, ~50 us --
        # is hidden behind the previous launch instead of being charged to a 0.6 ms kernel)
        rs = []
        for _ in range(3):
            env.reset(renew_obstacles=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(4):
                out = env.rollout(pool); del out
            e1.record(); torch.cuda.synchronize()
            rs.append(e0.elapsed_time(e1) * 1e3 / (4 * T_ep))
        ro_us = float(np.median(rs))
    except RuntimeError:                               # e.g. not enough memory for the [T, ...] outputs
        ro_us = None
    # the plain entry point (dronesim_rollout: no episode records, no in-kernel reset) on an env of the same shape
    rp_us = None
    if ro_us is not None and args.scaling != "strong":
        try:
            penv = drones(N, 0, [G, G], "O", k_closest=2, deltas=np.ones(N) * delta, simplify_zstate=True,
                          n_envs=E_global, device=dev, seed=1234, rank=rank, world_size=world, batched=True)
            out = penv.rollout(pool); torch.cuda.synchronize(); del out
            rs = []
            for _ in range(3):
                penv.reset(renew_obstacles=False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(4):
                    out = penv.rollout(pool); del out
                e1.record(); torch.cuda.synchronize()
                rs.append(e0.elapsed_time(e1) * 1e3 / (4 * T_ep))
            rp_us = float(np.median(rs))
            del penv
        except RuntimeError:
            rp_us = None
    # and with the actions drawn inside the kernel (dronesim_rollout_random: RandomAgent.forward, SAC_agents.py:22,
    # from a counter-based stream; no action pool, 44 B/agent-step), episode layer on: runs across episode ends
    rr_us = None
    try:
        env.reset(renew_obstacles=False)
        out = env.rollout_random(T_ep); torch.cuda.synchronize(); del out
        rs = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(4):
                out = env.rollout_random(T_ep); del out
            e1.record(); torch.cuda.synchronize()
            rs.append(e0.elapsed_time(e1) * 1e3 / (4 * T_ep))
        rr_us = float(np.median(rs))
    except RuntimeError:
        rr_us = None

    el = torch.tensor([elapsed, eager_elapsed], dtype=torch.float64, device=cdev)
    per_rank = None
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        mine = torch.tensor([kern_ms, elapsed, float(E)], dtype=torch.float64, device=cdev)
        allr = torch.empty(world, 3, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine.view(1, 3))
        per_rank = [{"rank": r, "step_kernel_ms": float(allr[r, 0]), "timed_seconds": float(allr[r, 1]), "envs": int(allr[r, 2])}
                    for r in range(world)]
    elapsed, eager_elapsed = float(el[0].item()), float(el[1].item())

    if rank == 0:
        agent_steps = N * E_global * total_steps
        value = agent_steps / elapsed
        bytes_launch = algorithmic_bytes(N, E, layer)
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload) if (e_gpu == WORKLOADS[args.workload][1] and E == e_gpu) else (None, None)
        out = {
            "metric": "env agent-steps/sec (n_agents x n_envs x steps)",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "repeats": repeats, "graph_copies": copies, "timed_steps": total_steps, "timed_seconds": elapsed,
            "ms_per_step": elapsed / total_steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "n_agents": N, "envs_per_gpu": E, "n_envs_total": E_global,
                       "grid": G, "delta": delta, "k_closest": 2, "simplify_zstate": True,
                       "launch": ((f"hipGraph of the {K} requested steps" + (f", captured {copies}x over ({L} launches)" if copies > 1 else "") +
                                   f", replayed {repeats}x in the timed region") if graph is not None else "eager"),
                       "episode_layer": ("per-step statistic + in-kernel auto-reset (dronesim_step_ex)" if layer else
                                         "plain dronesim_step + reset kernel every 200 steps"),
                       "actions": "pre-generated U(-1,1)^2, resident in HBM" if policy is None else
                                  f"batched per-agent {args.policy} policy (random-init, {args.policy_precision} MFMA) on the observation",
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # SURVEY.md 8(d) bytes only (76 B per agent-step + 13 B per env-step, no episode-record bytes)
                         "frac_survey_bytes": algorithmic_bytes(N, E, False) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "survey_bytes_per_launch": algorithmic_bytes(N, E, False),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "drone_kernel<K=2,FAR=0,step,%s>" % ("episode layer" if layer else "plain"), "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": bytes_launch},
            "exchange": {"what": "per-episode log of train_problem.py:118-121: fixed-order reduction of the per-env episode "
                                 "records (dronesim_episode_reduce, in the capture) + all-gather of the reduced 8-double vectors",
                         "reduce_every_steps": T_ep if L >= T_ep else L,
                         "reductions_in_timed_region": reductions,
                         "collectives_in_timed_region": len(exchanges),
                         "doubles_per_collective_and_rank": int(ring.numel()),
                         "collective": ("all_gather_into_tensor over %s, world_size %d, async on the collective's stream"
                                        % ("RCCL (backend nccl)" if backend == "nccl" else backend, world)) if world > 1 else
                                       "none: world_size 1, the reduced vectors are already everywhere (see rccl_probe)",
                         "real_collective_ran": bool(world > 1),
                         "backend": backend if world > 1 else None,
                         "rccl_ranks": (dist.get_world_size() if (world > 1 and backend == "nccl") else 0),
                         "collective_latency_us": allgather_us},
            "launch_check": launch_check,
            "episode_end_stats": summary,
            "eager": {"value": N * E_global * eager_steps / eager_elapsed, "ms_per_step": eager_elapsed / eager_steps * 1e3,
                      "steps": eager_steps,
                      "note": f"max(K, {EAGER_MIN_STEPS}) steps launched one by one from Python, no hipGraph (host launch latency included)"},
            "fused_rollout": None if ro_us is None else {
                "us_per_step_per_gpu": ro_us, "agent_steps_per_s_per_gpu": N * E / ro_us * 1e6,
                "roofline_frac_52B": 52.0 * N * E / (ro_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "note": "dronesim_rollout_ex on the bench's env (episode records + in-kernel reset on): 200 steps per launch, "
                        "52 B/agent-step (no per-step state write-back); four launches per timing bracket",
                "plain_entry_point": None if rp_us is None else {
                    "us_per_step_per_gpu": rp_us, "agent_steps_per_s_per_gpu": N * E / rp_us * 1e6,
                    "roofline_frac_52B": 52.0 * N * E / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "note": "dronesim_rollout (no episode layer) on an env of the same shape"},
                "random_actions_in_kernel": None if rr_us is None else {
                    "us_per_step_per_gpu": rr_us, "agent_steps_per_s_per_gpu": N * E / rr_us * 1e6,
                    "note": "dronesim_rollout_random: actions drawn in the kernel (no action pool, 44 B/agent-step), "
                            "episode records + in-kernel reset on"}},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(N, G, delta, args.cpu_budget)
            out["cpu_baseline"]["reference"] = reference_record(args.workload)
        else:
            out["cpu_baseline"] = None
        # the other BASELINE configs, timed by THIS process after the graded region (single-GPU default run only)
        if (world == 1 and not args.no_other_workloads and args.workload == "c3" and policy is None and layer
                and args.scaling == "weak" and not args.envs_per_gpu):
            del pool
            torch.cuda.empty_cache()
            other = {}
            for key, wl, pk, pr in (("c2", "c2", None, "f32"), ("c5_env", "c5", None, "f32"),
                                    ("c3_8192", "c3_8192", None, "f32"), ("c4_one_gpu", "c4_one_gpu", None, "f32"),
                                    ("c5_full", "c5_full", None, "f32"),
                                    ("default_construction", "c3", None, "f32"),
                                    ("c5_gaussian_f32", "c5", "gaussian", "f32"), ("c5_gaussian_bf16x3", "c5", "gaussian", "bf16x3"),
                                    ("c5_gaussian_f16x2", "c5", "gaussian", "f16x2"),
                                    ("c3_softmax16_f32", "c3", "softmax16", "f32")):
                try:
                    other[key] = side_workload(torch, dev, wl, pk, precision=pr, default_construction=(key == "default_construction"))
                except Exception as ex:                 # a side measurement must never cost the headline line
                    other[key] = {"error": f"{type(ex).__name__}: {ex}"}
            try:                                        # learner-side scans, reset, controllers
                other["aux_kernels"] = aux_kernels(torch, dev)
            except Exception as ex:
                other["aux_kernels"] = {"error": f"{type(ex).__name__}: {ex}"}
            out["other_workloads"] = other
        if world == 1 and not args.no_other_workloads and args.workload in ("c2", "c3", "c5") and policy is None and layer and not args.envs_per_gpu:
            torch.cuda.empty_cache()
            ds = device_span(args.workload)
            if ds is not None:                           # bytes moved inside the launch's own execution window
                ds["frac_inside_span"] = bytes_launch / (ds["span_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["device_clock"] = ds
        if world == 1 and not args.no_rccl_probe:
            out["exchange"]["rccl_probe"] = rccl_probe()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
