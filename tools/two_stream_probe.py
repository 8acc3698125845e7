#!/usr/bin/env python3
"""Developer probe: does splitting the C3 batch into S env shards, each stepped by its own chain of launches on its own
HIP stream (one graph per shard, replayed concurrently), hide the dependent-launch boundary and the lockstep of the
load / compute / store phases?   usage: python tools/two_stream_probe.py [S ...]   (default 1 2 4)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import drones

N, E, G, DELTA = 64, 4096, 28.0, 1.0


def run(S, steps=200, reps=15):
    Es = E // S
    envs, pools, streams, graphs = [], [], [], []
    g = torch.Generator(device="cuda").manual_seed(0)
    for s in range(S):
        envs.append(drones(N, 0, [G, G], "O", deltas=np.ones(N) * DELTA, simplify_zstate=True, n_envs=E, batched=True,
                           seed=1, rank=s, world_size=S))
        pools.append(torch.rand(steps, Es, N, 2, device="cuda", generator=g) * 2 - 1)
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    for s in range(S):
        with torch.cuda.stream(streams[s]):
            for k in range(5):
                envs[s].step(pools[s][k])
    torch.cuda.synchronize()
    for s in range(S):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=streams[s]):
            for k in range(steps):
                envs[s].step(pools[s][k])
        graphs.append(gr)
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        for e in envs:
            e.reset(renew_obstacles=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for s in range(S):
            streams[s].wait_event(a)
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                graphs[s].replay()
        for s in range(S):
            torch.cuda.current_stream().wait_stream(streams[s])
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / steps * 1e3)
    us = float(np.median(times))
    byt = 76 * N * E + 13 * E
    print(f"S={S}: {Es} envs per shard  {us:7.2f} us per full step (min {min(times):.2f})  {N*E/us*1e6:.3e} agent-steps/s  "
          f"{byt/us/1e3/8000*100:.1f}% of 8 TB/s aggregate", flush=True)


if __name__ == "__main__":
    for S in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
        run(S)
