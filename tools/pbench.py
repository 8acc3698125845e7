#!/usr/bin/env python3
"""Developer benchmark: batched policy forward (csrc/policy.hip) alone and inside the rollout loop
obs -> sample_action -> env.step (hipGraph replay)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import drones
from scalable_collision_avoidance_rl_amd.policies import BatchedMLP
from tools.kbench import PRESETS


def rnd_policy(kind, N, d, dev, precision="f32"):
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * 0.2
    if kind == "softmax16":
        h1 = h2 = 300; nout = 16; ok, sk = 1, 1
    elif kind == "gaussian":
        h1 = h2 = 400; nout = 4; ok, sk = 2, 2
    else:
        h1 = h2 = 200; nout = 1; ok, sk = 0, 0
    return BatchedMLP(r(N, d, h1), r(N, h1), r(N, h1, h2), r(N, h2), r(N, h2, nout), r(N, nout), ok, sk, device=dev, precision=precision,
                      pack_w2={"0": False, "1": True}.get(os.environ.get("PB_PACK", "1"), os.environ.get("PB_PACK")),
                      split_kernel=bool(os.environ.get("PB_SPLIT"))), (h1, h2, nout)   # PB_SPLIT=1: f16x2 on the split kernel of rounds 2-5
    # PB_PACK=0: the plain [N, h1, h2] layer 2; PB_PACK=fragments: the fragment-packed layer 2 of rounds 3-5 (default: the row-tile stream)


def timeit(fn, steps=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(steps):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / steps * 1e3)
    return float(np.median(ts))


if __name__ == "__main__":
    for spec in (sys.argv[1:] or ["c3", "c2", "c5"]):
        N, E, G, delta = PRESETS[spec]
        env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
        for kind, prec in [(k, p) for p in (os.environ.get("PB_PREC", "f32,bf16x3,f16x2,bf16").split(",")) for k in os.environ.get("PB_KINDS", "softmax16,gaussian,critic").split(",")]:
            pol, (h1, h2, nout) = rnd_policy(kind, N, 6, env.device, prec)
            z = env.z
            flops = 2.0 * E * N * (6 * h1 + h1 * h2 + h2 * nout)
            us = timeit((lambda: pol.sample_action(z)) if pol.sample_kind else (lambda: pol.forward(z)))
            line = f"{spec} {kind:>9} {prec:>6}: policy {us:8.1f} us/step = {flops/us/1e6:6.1f} TFLOP/s"
            if pol.sample_kind:
                def loop():
                    act, _ = pol.sample_action(env.z, env=env)
                    env.step(act)
                us2 = timeit(loop)
                line += f" | obs->policy->step {us2:8.1f} us/step = {N*E/us2*1e6:.3e} agent-steps/s"
            print(line, flush=True)
