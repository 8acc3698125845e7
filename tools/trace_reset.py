#!/usr/bin/env python3
"""Developer tool: per-wave phase timestamps of env.reset() as one launch (dronesim_reset_observe); needs a -DDRONESIM_TRACE build,
selected with DRONESIM_LIB=...  Phases: entry -> draw done (lattice nodes settled, state written) -> tables synced -> pairs done ->
stores issued -> end -> stores acknowledged."""
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
for _ in range(5):
    env.reset(renew_obstacles=False)
torch.cuda.synchronize()
lib.dronesim_debug_set_trace.argtypes = [C.c_void_p]
lib.dronesim_debug_set_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.reset(renew_obstacles=False); e1.record()
torch.cuda.synchronize()
lib.dronesim_debug_set_trace(None)
t = trace.cpu().numpy().astype(np.float64)
names = ["entry", "draw done", "tables synced", "pairs done", "stores issued", "end", "stores acked"]
print(f"{spec}: {waves} waves, event time {e0.elapsed_time(e1)*1e3:.1f} us; phase lengths in s_memtime ticks (per-wave differences)")
d = np.diff(t[:, :7], axis=1)
for k in range(6):
    print(f"  phase {names[k]:>14} -> {names[k+1]:<14}: median {np.median(d[:, k]):8.0f}  p95 {np.percentile(d[:, k], 95):8.0f}  max {d[:, k].max():8.0f}")
rt = trace[:, 7].cpu().numpy()
ent = (rt & 0xffffffff).astype(np.int64); ext = ((rt >> 32) & 0xffffffff).astype(np.int64)
e = (ent - ent.min()) * 0.01; x = (ext - ent.min()) * 0.01
print(f"  global clock (us since the first wave entered): entry p50 {np.median(e):.2f} p95 {np.percentile(e, 95):.2f} max {e.max():.2f}; "
      f"exit p5 {np.percentile(x, 5):.2f} p50 {np.median(x):.2f} max {x.max():.2f}")
