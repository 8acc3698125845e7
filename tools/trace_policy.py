#!/usr/bin/env python3
"""Developer tool: per-wave phase timestamps of the bf16 policy kernel (needs a -DDRONESIM_TRACE build,
selected with DRONESIM_LIB=build/libdronesim_trace.so).  usage: trace_policy.py [softmax16|gaussian|critic] [c3] [bf16|f32]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if len(sys.argv) > 3 and sys.argv[3] == "f32":                       # the stamps live in the LDS-staged f32 kernel (the row-tile kernels: trace_rt16.py)
    os.environ.setdefault("PB_PACK", "fragments")
from scalable_collision_avoidance_rl_amd import _native, drones
from tools.kbench import PRESETS
from tools.pbench import rnd_policy

kind = sys.argv[1] if len(sys.argv) > 1 else "softmax16"
spec = sys.argv[2] if len(sys.argv) > 2 else "c3"
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
pol, shape = rnd_policy(kind, N, 6, env.device, prec)
run = (lambda: pol.sample_action(env.z)) if pol.sample_kind else (lambda: pol.forward(env.z))
for _ in range(3):
    run()
torch.cuda.synchronize()
blocks = ((E + (63 if prec == "bf16" else 31)) // (64 if prec == "bf16" else 32)) * N
trace = torch.zeros(blocks, 4, 8, dtype=torch.int64, device="cuda")
lib = _native.lib()
lib.dronesim_debug_set_policy_trace.argtypes = [C.c_void_p]
lib.dronesim_debug_set_policy_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
lib.dronesim_debug_set_policy_trace(None)
t = trace.cpu().numpy().astype(np.float64)
names = ["entry", "prologue + barrier", "layer 1 done", "barrier", "h1 tile in registers", "layers 2+3 done",
         "partials + barriers", "finish"]
if prec == "f32":
    names = ["entry", "x tile + barrier", "layer 1 done", "barrier", "layers 2+3 done", "partials + barrier", "finish"]
print(f"{kind} {spec} {shape}: {blocks} workgroups, event time {e0.elapsed_time(e1)*1e3:.1f} us; ticks ~ 100 MHz or core clock, see ratio")
last = 6 if prec == "bf16" else 5
life = t[:, :, last] - t[:, :, 0]
print(f"  wave lifetime entry -> partials: median {np.median(life):.0f} p95 {np.percentile(life, 95):.0f} ticks")
for k in range(1, last + 1):
    d = t[:, :, k] - t[:, :, k - 1]
    per = " ".join(f"w{w}:{np.median(d[:, w]):7.0f}" for w in range(4))
    print(f"  {names[k]:>24}: median {np.median(d):8.0f}  p95 {np.percentile(d, 95):8.0f}   {per}")
if prec == "f32":                                   # stamp 7: layer 1's operands (W1, b1, x tile from LDS) have arrived
    d = (t[:, :, 7] - t[:, :, 1])[t[:, :, 7] > 0]
    if d.size:
        print(f"  {'of layer 1: operand wait':>24}: median {np.median(d):8.0f}  p95 {np.percentile(d, 95):8.0f}")
if prec == "f32":                                   # waves 2, 3: slot 6 = the wave's lifetime on the 100 MHz clock
    d = (t[:, :2, last + 1] - t[:, :2, last])[t[:, :2, last + 1] > 0]
    rt = t[:, 2:, 6]
    mhz = (t[:, 2:, 5] - t[:, 2:, 0])[rt > 0] / rt[rt > 0] * 100.0
    print(f"  shader clock over the waves' lifetimes: median {np.median(mhz):.0f} MHz  p5 {np.percentile(mhz, 5):.0f}  p95 {np.percentile(mhz, 95):.0f}"
          f"   (lifetime median {np.median(rt[rt > 0]) / 100.0:.1f} us)")
else:
    d = (t[:, :, last + 1] - t[:, :, last])[t[:, :, last + 1] > 0]
print(f"  {names[last + 1]:>24}: median {np.median(d):8.0f}  p95 {np.percentile(d, 95):8.0f}")
span = t.max() - t[:, :, 0].min()
print(f"  kernel span {span:.0f} ticks -> {span / (e0.elapsed_time(e1) * 1e3):.1f} ticks/us")
