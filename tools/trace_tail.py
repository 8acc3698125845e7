#!/usr/bin/env python3
"""Developer tool (needs DRONESIM_LIB=build/libdronesim_trace.so): which waves end an ordinary step launch INSIDE a graph
replay?  Last launch of a 150-step graph from a fresh reset: exit-time percentiles and the phases of the slowest waves."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_collision_avoidance_rl_amd import _native, drones
from tools.kbench import PRESETS

spec = sys.argv[1] if len(sys.argv) > 1 else "c3"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1,
             auto_reset=bool(os.environ.get("AUTO", "1") != "0"))
lib = _native.lib()
lib.dronesim_debug_set_trace.argtypes = [C.c_void_p]
waves = E * max(1, (N + 63) // 64) if N > 64 else (E + (64 // N) - 1) // (64 // N)
g = torch.Generator(device="cuda").manual_seed(0)
pool = torch.rand(200, E, N, 2, device="cuda", generator=g) * 2 - 1
for s in range(5):
    env.step(pool[s])
torch.cuda.synchronize()
trace = torch.zeros(waves, 8, dtype=torch.int64, device="cuda")
env.reset(renew_obstacles=False)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for s in range(L - 1):
        env.step(pool[s % 200])
    lib.dronesim_debug_set_trace(trace.data_ptr())
    env.step(pool[(L - 1) % 200])
    lib.dronesim_debug_set_trace(None)
for rep in range(3):
    env.reset(renew_obstacles=False); torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    t = trace.cpu().numpy()
    rt = t[:, 7]
    ent = (rt & 0xffffffff).astype(np.int64); ext = ((rt >> 32) & 0xffffffff).astype(np.int64)
    e = (ent - ent.min()) * 0.01; x = (ext - ent.min()) * 0.01
    ph = np.diff(t[:, :7].astype(np.float64), axis=1) / 2100.0        # us at ~2.1 GHz
    q = [50, 90, 95, 99, 99.9, 100]
    print(f"{spec} replay {rep}: exit percentiles (us) " + " ".join(f"p{p}={np.percentile(x, p):.2f}" for p in q) +
          f" | entry p50 {np.median(e):.2f} p99 {np.percentile(e, 99):.2f} max {e.max():.2f}")
    order = np.argsort(-x)[:12]
    names = ["load", "sync", "pairs", "epilogue", "copy-out", "ack"]
    print("   all waves   : life %.2f  " % np.median(x - e) + " ".join(f"{n} {np.median(ph[:, k]):.2f}" for k, n in enumerate(names)))
    for w in order:
        print(f"   wave {w:5d} (WG {w // 4:4d}, XCD {(w // 4) % 8}): entry {e[w]:.2f} exit {x[w]:.2f}  " + " ".join(f"{n} {ph[w, k]:.2f}" for k, n in enumerate(names)))
