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

The JSON line is kept under 8 KB (the driver keeps an 8 KB tail of stdout): numbers with 5-6 significant digits and short
keys, no prose -- what every key means, and the workload behind every side line, is in DESIGN.md section 7.

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
    "c2": (5, 1024, 5.0, 1.0, "C2 n=5x1024"),
    "c3": (64, 4096, 28.0, 1.0, "C3 n=64x4096"),
    "c5": (256, 512, 256.0, 2.5, "C5-shard n=256x512"),
}
# side lines only (`other_workloads`): the shapes that carry the "boundary amortised" part of the roofline argument
SIDE_WORKLOADS = {
    "c3_8192": (64, 8192, 28.0, 1.0, "n=64x8192"),
    "c4_one_gpu": (64, 32768, 28.0, 1.0, "C4 n=64x32768 on one rank"),
    "c5_full": (256, 4096, 256.0, 2.5, "C5 n=256x4096 on one rank"),
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
    container where /root/reference exists; it cannot travel to the GPU box: profiles/reference_cpu.json) -- attached
    beside the live figure of the C port so that both CPU numbers stand next to the GPU one."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu.json")))
        sh = rec["shapes"][workload]
        return {"src": "profiles/reference_cpu.json", "core1": sh["one_core"]["agent_steps_per_s"],
                "core1_ms_per_step": sh["one_core"]["ms_per_step"],
                "host": sh["whole_host"]["agent_steps_per_s"], "host_cores": sh["whole_host"]["cores"]}
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
            "sample": f"oracle/drone_oracle.c f64, {E} envs x {steps} steps, {cores} threads, {el:.1f} s"}


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
    ms = el / steps * 1e3
    out = {"v": N * E * steps / el, "ms_per_step": ms, "steps": steps, "step_kernel_ms": kern_ms,
           "frac": byt / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "B": byt}
    if policy is not None:
        # the policy launch alone (events around a graph of 20 back-to-back calls, median of five replays) ...
        pg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(pg):
            for _ in range(20):
                policy.sample_action(env.z, env=env)
        pg.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); pg.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        alone_ms = float(np.median(ts))
        # ... and IN THE LOOP: what the policy launch adds to a step of the timed loop = loop time per step minus the step
        # launch's own time (the figure the roofline below is priced on; it cannot exceed ms_per_step)
        pol_ms = ms - kern_ms
        flops = 2.0 * E * N * (policy.d_in * policy.h1 + policy.h1 * policy.h2 + policy.h2 * policy.nout)
        nprod = {"f32": 1.0, "bf16x3": 6.0, "f16x2": 3.0}.get(precision, 1.0)   # matrix products issued per float32 product
        peak = 157.3 if precision == "f32" else 2500.0
        out.update({"policy_kernel_ms": pol_ms, "policy_alone_ms": alone_ms, "pol_tf": nprod * flops / (pol_ms * 1e-3) / 1e12,
                    "pol_peak": peak, "pol_frac": nprod * flops / (pol_ms * 1e-3) / 1e12 / peak,
                    "f32_eq_tf": flops / (pol_ms * 1e-3) / 1e12})
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

    def line(what, us, byt, kernel, note):      # (what / kernel / note: documentation of the call sites)
        return {"us": us, "frac": byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "B": byt}
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
                        "(drone_kernel<observe>) (drone_env.py:98-102, 171-212)", us, E * N * (16 + 36) + E * 12,
                        "drone_kernel<K=2,FAR=0,observe,plain> with the in-kernel draw (dronesim_reset_observe)",
                        "ONE launch since round 6: writes pos 8 + vel 8 + z 24 + nbr_idx 12 per agent; t / episode per env "
                        "(the two launches of rounds 1-5 also read pos / vel back: 68 B per agent)")
    for kind, lines in (("proportional", "drone_env.py:652-679"), ("gradient", "drone_env.py:609-650")):
        us = graph_time_us(torch, lambda: env.control(kind), 40)
        out[f"control_{kind}"] = line(f"{kind}_control for all agents of C3 ({lines})", us, E * N * 16,
                                      "control_kernel", "reads pos 8 B, writes act 8 B per agent: a 4 MB launch")
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
                return {"world": 1, "all_gather_8_doubles_us": float(us), "ok": ok == "True"}
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


def write_ceiling():
    """What the chip sustains for the step kernel's dominant traffic -- a pure WRITE stream in the kernel's own shape (16 B per lane,
    non-temporal) -- measured on THIS box by tools/stream_bw (tools/micro/stream_bw.hip, built by __graft_entry__.build()): sustained
    GB/s of one launch over 2 GiB, and us per launch of 4000 dependent launches (one hipGraph) writing 15.7 MB each (= what one C3 step launch writes:
    the floor of any one-launch-per-step kernel with this output volume).  60 of the path's 76 B per agent-step are WRITTEN; HBM3E on
    this part sustains ~4.3 TB/s of writes against ~7.1 TB/s of reads, so the 8 TB/s spec is not reachable for this mix.  None when the tool is not built."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "stream_bw")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        out = {}
        for line in r.stdout.splitlines():
            w = line.split()
            if line.startswith("write one launch"):
                out["write_GBps"] = float(w[w.index("->") + 1])
            elif line.startswith("write 4000 dependent"):
                out["write_15p7MB_launch_us"] = float(w[w.index("us") - 1])
            elif line.startswith("read  one launch"):
                out["read_GBps"] = float(w[w.index("->") + 1])
            elif line.startswith("copy  one launch"):
                out["copy_GBps"] = float(w[w.index("->") + 1])
        return out or None
    except (subprocess.SubprocessError, OSError, ValueError):
        return None


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
                return d
    except (subprocess.SubprocessError, OSError, ValueError):
        pass
    return None


def compact(o, sig=5):
    """Numbers of the JSON line at `sig` significant digits (floats that are whole numbers print as integers)."""
    if isinstance(o, dict):
        return {k: compact(v, 7 if k in ("value", "timed_seconds", "v") else sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v, sig) for v in o]
    if isinstance(o, (float, np.floating)):
        f = float(o)
        if f != f or f in (float("inf"), float("-inf")):
            return None
        if f == int(f) and abs(f) < 1e15:
            return int(f)
        return float(f"{f:.{sig}g}")
    if isinstance(o, np.integer):
        return int(o)
    return o


def c1_compat(torch, dev, episodes=3):
    """BASELINE configs[0]'s shape (n = 5, ONE env, 200-step episodes) through the E = 1 drop-in face: the loop of
    train_problem.py:82-107 -- list-of-ndarray actions in, the reference's 6-tuple of host objects out, env.reset() between
    episodes -- in ms per step() call, next to the reference's own step() on one CPU core (profiles/reference_cpu.json).
    Every call pays one host-to-device copy, one launch and one device-to-host copy + sync: a compatibility face, never `value`."""
    from scalable_collision_avoidance_rl_amd import drones, max_time_steps
    N = 5
    env = drones(N, 0, [5.0, 5.0], "O", k_closest=2, deltas=np.ones(N), simplify_zstate=True, device=dev, seed=7)
    rng = np.random.default_rng(0)
    acts = [[rng.uniform(-1, 1, 2) for _ in range(N)] for _ in range(max_time_steps)]
    for a in acts[:20]:
        env.step(a)
    steps, t_step = 0, 0.0
    for _ in range(episodes):
        env.reset()
        for a in acts:
            t0 = time.perf_counter()
            _, _, _, _, finished, _ = env.step(a)
            t_step += time.perf_counter() - t0
            steps += 1
            if finished:
                break
    ref = reference_record("c2")                       # same n = 5, G = 5 shape, E = 1: the reference's step() on one core
    ms = t_step / steps * 1e3
    return {"ms_per_step": ms, "steps": steps, "v": N / (ms * 1e-3), "ref_ms_per_step": None if ref is None else ref["core1_ms_per_step"],
            "vs_ref": None if ref is None else ref["core1_ms_per_step"] / ms}


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

    # world > 1: the figure the N-rank value is to be compared with -- the SAME graph replayed by rank 0 ALONE (the other
    # ranks wait at the barrier below), i.e. this workload on one GPU with nothing else running on the node.  Computed by
    # the script so that no reader has to pair lines of different runs.  (Untimed for the N-rank figure: it precedes it.)
    n1 = None
    if world > 1 and graph is not None and args.scaling == "weak":     # (strong: a rank's shard is not the N = 1 job)
        barrier()
        if rank == 0:
            r1 = max(1, int(np.ceil(0.6 * args.min_seconds / one)))
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(r1):
                graph.replay(); launches += L
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            n1 = {"value": N * E * L * r1 / e1, "ms": e1 / (L * r1) * 1e3}
        barrier()

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
    # ... and over all ranks: sum of launches x envs of every rank (rank 0 also issued the N = 1 comparison replays)
    issued = torch.tensor([float(launches) * E], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(issued, op=dist.ReduceOp.SUM)
    launch_check = {"launches_rank0": launches, "envs_per_rank": E,
                    "env_steps_issued_all_ranks": float(issued.item()),
                    "env_steps_recorded_all_ranks": summary["env_steps"],
                    "ok": (not layer) or summary["env_steps"] == float(issued.item())}

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
        fr52 = lambda us: 52.0 * N * E / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": "env agent-steps/sec (n_agents x n_envs x steps)",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "repeats": repeats, "graph_copies": copies, "timed_steps": total_steps, "timed_seconds": elapsed,
            "ms_per_step": elapsed / total_steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "n_agents": N, "envs_per_gpu": E, "n_envs_total": E_global,
                       "grid": G, "delta": delta, "k_closest": 2, "simplify_zstate": True,
                       "launch": (f"hipGraph {K}x{copies} launches, {repeats} replays" if graph is not None else "eager"),
                       "episode_layer": bool(layer),
                       "actions": "pool U(-1,1)^2 in HBM" if policy is None else f"{args.policy} policy {args.policy_precision}",
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # SURVEY.md 8(d) bytes only (76 B per agent-step + 13 B per env-step, no episode-record bytes)
                         "frac_survey_bytes": algorithmic_bytes(N, E, False) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "survey_bytes_per_launch": algorithmic_bytes(N, E, False),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "drone_kernel<K=2,FAR=0,step,%s>" % ("epi" if layer else "plain"), "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": bytes_launch},
            # the path's only exchange (train_problem.py:118-121): in-capture fixed-order reductions + one all-gather per replay
            "exchange": {"reduce_every_steps": T_ep if L >= T_ep else L,
                         "reductions_in_timed_region": reductions,
                         "collectives_in_timed_region": len(exchanges),
                         "doubles_per_collective_and_rank": int(ring.numel()),
                         "real_collective_ran": bool(world > 1),
                         "backend": backend if world > 1 else None,
                         "rccl_ranks": (dist.get_world_size() if (world > 1 and backend == "nccl") else 0),
                         "collective_latency_us": allgather_us},
            "launch_check": launch_check,
            "episode_end_stats": summary,
            "eager": {"ms_per_step": eager_elapsed / eager_steps * 1e3, "steps": eager_steps},
            # dronesim_rollout_ex (episode layer) / dronesim_rollout (plain) / dronesim_rollout_random: us per step, 52 B per agent-step
            "fused_rollout": None if ro_us is None else {
                "epi_us": ro_us, "epi_frac52": fr52(ro_us),
                "plain_us": rp_us, "plain_frac52": None if rp_us is None else fr52(rp_us),
                "rand_us": rr_us},
        }
        if n1 is not None:          # the N = 1 figure of the SAME workload / episode layer, timed on rank 0 alone ahead of the N-rank region
            out["n1"] = {"value": n1["value"], "ms_per_step": n1["ms"], "ratio": value / n1["value"],
                         "efficiency": value / n1["value"] / world}
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
                                    ("c5_full_gaussian_f16x2", "c5_full", "gaussian", "f16x2"),   # configs[4]'s whole env axis WITH its policy, one rank
                                    ("c3_softmax16_f32", "c3", "softmax16", "f32"), ("c3_softmax16_f16x2", "c3", "softmax16", "f16x2")):
                try:
                    other[key] = side_workload(torch, dev, wl, pk, precision=pr, default_construction=(key == "default_construction"))
                except Exception as ex:                 # a side measurement must never cost the headline line
                    other[key] = {"error": f"{type(ex).__name__}: {ex}"}
            try:                                        # learner-side scans, reset, controllers
                other["aux_kernels"] = aux_kernels(torch, dev)
            except Exception as ex:
                other["aux_kernels"] = {"error": f"{type(ex).__name__}: {ex}"}
            try:                                        # BASELINE configs[0] shape through the E = 1 drop-in face
                other["c1_compat"] = c1_compat(torch, dev)
            except Exception as ex:
                other["c1_compat"] = {"error": f"{type(ex).__name__}: {ex}"}
            out["other_workloads"] = other
        if world == 1 and not args.no_other_workloads and args.workload in ("c2", "c3", "c5") and policy is None and layer and not args.envs_per_gpu:
            torch.cuda.empty_cache()
            ds = device_span(args.workload)
            if ds is not None:                           # bytes moved inside the launch's own execution window
                ds["frac_inside_span"] = bytes_launch / (ds["span_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["device_clock"] = ds
            wc = write_ceiling()
            if wc is not None and "write_GBps" in wc:    # bytes WRITTEN per launch (60 B per agent-step + per-env words) against the measured write rate
                written = 60 * N * E + (9 + (32 if layer else 0)) * E
                wc["written_bytes_per_launch"] = written
                wc["frac"] = written / (kern_ms * 1e-3) / 1e9 / wc["write_GBps"]
            out["roofline"]["write_ceiling"] = wc
        if world == 1 and not args.no_rccl_probe:
            out["exchange"]["rccl_probe"] = rccl_probe()
        line = json.dumps(compact(out), separators=(",", ":"))
        if len(line) > 7900:                            # the driver keeps 8 KB of stdout: never let the head fall off
            print(f"bench.py: JSON line is {len(line)} bytes (> 7900): dropping episode_end_stats / per-rank detail", file=sys.stderr)
            out.pop("episode_end_stats", None)
            line = json.dumps(compact(out), separators=(",", ":"))
        print(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
