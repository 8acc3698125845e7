#!/usr/bin/env python3
"""Developer tool: per-wave phase timestamps of the step kernel (needs a -DDRONESIM_TRACE build,
selected with DRONESIM_LIB=build/libdronesim_trace.so)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import _native, drones
from tools.kbench import PRESETS

spec = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
lib = _native.lib()
waves = E * max(1, (N + 63) // 64) if N > 64 else (E + (64 // N) - 1) // (64 // N)
trace = torch.zeros(waves, 8, dtype=torch.int64, device="cuda")
act = torch.rand(E, N, 2, device="cuda") * 2 - 1
for _ in range(5):
    env.step(act)
torch.cuda.synchronize()
lib.dronesim_debug_set_trace.argtypes = [C.c_void_p]
lib.dronesim_debug_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.step(act); e1.record()
torch.cuda.synchronize()
lib.dronesim_debug_set_trace(None)
t = trace.cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
names = ["entry", "loaded+LDS written", "after barrier", "pairs done", "stores issued", "end", "stores acked"]
print(f"{spec}: {waves} waves, event time {e0.elapsed_time(e1)*1e3:.1f} us; timestamps in ticks since first wave entry")
for k, nm in enumerate(names):
    col = t[:, k] - t0
    print(f"  {nm:>20}: min {col.min():9.0f}  median {np.median(col):9.0f}  max {col.max():9.0f}")
d = np.diff(t[:, :7], axis=1)
for k in range(6):
    print(f"  phase {names[k]:>20} -> {names[k+1]:<20}: median {np.median(d[:, k]):8.0f}  p95 {np.percentile(d[:, k], 95):8.0f}")
xcc = t[:, 7].astype(int)
for x in sorted(set(xcc)):
    sel = xcc == x
    e = t[sel, 0]; f = t[sel, 6]
    print(f"  XCC {x}: {sel.sum():5d} waves; entry spread {e.max()-e.min():8.0f}; first entry -> last ack {f.max()-e.min():8.0f} ticks")
span = (t[:, 6].max() - t0)
print(f"  total span {span:.0f} ticks; if 100 MHz ticks -> {span/100:.2f} us; if 2.4 GHz -> {span/2400:.2f} us")
