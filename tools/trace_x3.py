#!/usr/bin/env python3
"""Developer tool: shader clock and per-wave stream time of the bf16x3 policy kernel (needs a -DDRONESIM_TRACE build,
selected with DRONESIM_LIB=build/libdronesim_trace.so).  usage: trace_x3.py [softmax16|gaussian|critic] [c3]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import _native, drones
from tools.kbench import PRESETS
from tools.pbench import rnd_policy

kind = sys.argv[1] if len(sys.argv) > 1 else "gaussian"
spec = sys.argv[2] if len(sys.argv) > 2 else "c3"
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
pol, shape = rnd_policy(kind, N, 6, env.device, "bf16x3")
run = (lambda: pol.sample_action(env.z)) if pol.sample_kind else (lambda: pol.forward(env.z))
for _ in range(3):
    run()
torch.cuda.synchronize()
blocks = ((E + 63) // 64) * N
trace = torch.zeros(blocks, 4, 8, dtype=torch.int64, device="cuda")
lib = _native.lib()
lib.dronesim_debug_set_policy_trace.argtypes = [C.c_void_p]
lib.dronesim_debug_set_policy_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
lib.dronesim_debug_set_policy_trace(None)
t = trace.cpu().numpy().astype(np.float64)
core, real = t[:, :, 2] - t[:, :, 0], t[:, :, 3] - t[:, :, 1]
print(f"{kind} {spec} {shape}: {blocks} workgroups, event time {e0.elapsed_time(e1)*1e3:.1f} us")
print(f"  shader clock over the waves' streams: {core.sum() / real.sum() * 100:.0f} MHz (s_memtime ticks per 100 MHz tick)")
for w in range(4):
    print(f"  wave {w}: stream time median {np.median(real[:, w]) / 100:.2f} us  p95 {np.percentile(real[:, w], 95) / 100:.2f} us")
span = (t[:, :, 3].max() - t[:, :, 1].min()) / 100
print(f"  first entry -> last stream end: {span:.1f} us")
