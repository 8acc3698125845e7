#!/usr/bin/env python3
"""Developer tool (needs DRONESIM_LIB=build/libdronesim_trace.so, a -DDRONESIM_TRACE build): the LAST launch of a graph
of L steps from a fresh reset, per-wave stamps -- L = 199: an ordinary launch, L = 200: the launch in which every env
finishes its episode (in-kernel reset).  usage: trace_fire.py [c3]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_collision_avoidance_rl_amd import _native, drones
from tools.kbench import PRESETS

spec = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1, auto_reset=True)
lib = _native.lib()
lib.dronesim_debug_set_trace.argtypes = [C.c_void_p]
waves = E * max(1, (N + 63) // 64) if N > 64 else (E + (64 // N) - 1) // (64 // N)
g = torch.Generator(device="cuda").manual_seed(0)
pool = torch.rand(200, E, N, 2, device="cuda", generator=g) * 2 - 1
for s in range(5):
    env.step(pool[s])
torch.cuda.synchronize()
for L in (199, 200, 201):
    trace = torch.zeros(waves, 8, dtype=torch.int64, device="cuda")
    env.reset(renew_obstacles=False)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for s in range(L - 1):
            env.step(pool[s % 200])
        lib.dronesim_debug_set_trace(trace.data_ptr())           # only the last launch of the graph carries the pointer
        env.step(pool[(L - 1) % 200])
        lib.dronesim_debug_set_trace(None)
    for _ in range(3):
        env.reset(renew_obstacles=False); torch.cuda.synchronize()
        gr.replay(); torch.cuda.synchronize()
    t = trace.cpu().numpy()
    rt = t[:, 7]
    ent = (rt & 0xffffffff).astype(np.int64); ext = ((rt >> 32) & 0xffffffff).astype(np.int64)
    e = (ent - ent.min()) * 0.01; x = (ext - ent.min()) * 0.01
    print(f"{spec} last launch of a {L}-step graph: waves enter p50 {np.median(e):.2f} max {e.max():.2f} us; exit p5 {np.percentile(x, 5):.2f} "
          f"p50 {np.median(x):.2f} p95 {np.percentile(x, 95):.2f} max {x.max():.2f} us; wave lifetime p50 {np.median(x - e):.2f} max {(x - e).max():.2f} us", flush=True)
