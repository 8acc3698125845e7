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
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1, track_episodes=bool(os.environ.get("TRACK")), auto_reset=bool(os.environ.get("AUTO")))
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
print(f"{spec}: {waves} waves, event time {e0.elapsed_time(e1)*1e3:.1f} us; phase lengths in s_memtime ticks (per-wave differences; the counter is not synchronised across XCCs)")
d = np.diff(t[:, :7], axis=1)
for k in range(6):
    print(f"  phase {names[k]:>20} -> {names[k+1]:<20}: median {np.median(d[:, k]):8.0f}  p95 {np.percentile(d[:, k], 95):8.0f}")
rt = trace[:, 7].cpu().numpy()                    # s_memrealtime (100 MHz, global): entry in the low word, exit in the high word
ent = (rt & 0xffffffff).astype(np.int64); ext = ((rt >> 32) & 0xffffffff).astype(np.int64)
e = (ent - ent.min()) * 0.01; x = (ext - ent.min()) * 0.01
print(f"  global clock (us since the first wave entered): entry p50 {np.median(e):.2f} p95 {np.percentile(e, 95):.2f} max {e.max():.2f}; "
      f"exit p5 {np.percentile(x, 5):.2f} p50 {np.median(x):.2f} max {x.max():.2f}")
late = x >= np.percentile(x, 98)                  # what makes the last 2 % of the waves late?
ph = d / 2400.0                                   # phases in us at ~2.4 GHz
print(f"  last 2 % of the waves vs all (us): entry {e[late].mean():.2f} vs {e.mean():.2f}" +
      "".join(f"; {nm} {ph[late, k].mean():.2f} vs {ph[:, k].mean():.2f}"
              for k, nm in enumerate(["load", "sync", "pairs", "epilogue", "copy-out", "ack"])))
