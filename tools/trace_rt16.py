#!/usr/bin/env python3
"""Developer tool: per-wave phase timestamps (s_memtime) of the float16 row-tile policy kernel (mlp3_rt16_kernel; needs a
-DDRONESIM_TRACE build: DRONESIM_LIB=abl/libdronesim_trace.so).  usage: trace_rt16.py [gaussian|critic] [c5|c3]"""
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
spec = sys.argv[2] if len(sys.argv) > 2 else "c5"
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
pol, shape = rnd_policy(kind, N, 6, env.device, "f16x2")
run = (lambda: pol.sample_action(env.z)) if pol.sample_kind else (lambda: pol.forward(env.z))
for _ in range(3):
    run()
torch.cuda.synchronize()
blocks = ((E + 127) // 128) * N
trace = torch.zeros(blocks, 4, 64, dtype=torch.int64, device="cuda")
lib = _native.lib()
lib.dronesim_debug_set_policy_trace.argtypes = [C.c_void_p]
lib.dronesim_debug_set_policy_trace(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
lib.dronesim_debug_set_policy_trace(None)
t = trace.cpu().numpy().astype(np.float64)
us = e0.elapsed_time(e1) * 1e3
rt = t[:, :, 33] - t[:, :, 32]                                     # wave lifetimes on the constant 100 MHz clock
mhz = (t[:, :, 31] - t[:, :, 0]) / rt * 100.0
span = (t[:, :, 33].max() - t[:, :, 32].min()) / 100.0
print(f"{kind} {spec} {shape}: {blocks} workgroups, event time {us:.1f} us; first entry -> last finish {span:.1f} us; "
      f"shader clock over the waves' lifetimes: median {np.median(mhz):.0f} MHz (p5 {np.percentile(mhz, 5):.0f}, p95 {np.percentile(mhz, 95):.0f}), "
      f"lifetime median {np.median(rt) / 100.0:.1f} us")
life = t[:, :, 31] - t[:, :, 0]
print(f"  wave lifetime: median {np.median(life):.0f} p5 {np.percentile(life, 5):.0f} p95 {np.percentile(life, 95):.0f} ticks")
first = t[:, 0, 32] - t[:, :, 32].min()
order = np.argsort(first)
rounds = {"first round": order[:512], "later": order[512:]}
names = {1: "prologue + barrier", 15: "pass 0: layer 3", 29: "pass 1: layer 3", 30: "drain + barrier", 31: "finish"}
for c in range(13):
    names[2 + c] = f"pass 0: in-chunk {c}"
    names[16 + c] = f"pass 1: in-chunk {c}"
for rname, idx in rounds.items():
    if len(idx) == 0:
        continue
    tt = t[idx]
    print(f" {rname} ({len(idx)} workgroups; entry at {np.median(first[idx]):.0f} ticks, lifetime {np.median(tt[:, :, 31] - tt[:, :, 0]):.0f}):")
    prev = tt[:, :, 0]
    for k in range(1, 32):
        if not (tt[:, :, k] > 0).all():
            continue
        d = tt[:, :, k] - prev
        prev = tt[:, :, k]
        print(f"  {names.get(k, str(k)):>22}: median {np.median(d):8.0f}  p5 {np.percentile(d, 5):8.0f}  p95 {np.percentile(d, 95):8.0f}")

# where the workgroups ran (stamp 34 = HW_ID | XCC_ID << 32 of each wave)
from collections import Counter, defaultdict
hw = t[:, :, 34].astype(np.int64)
simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15
cukey = (xcc * 10000 + se * 1000 + sh * 100 + cu)[:, 0]
by_cu = defaultdict(list)
for b in order[:512]:
    by_cu[cukey[b]].append(int(b))
gaps = Counter(abs((v[0] >> 3) - (v[1] >> 3)) for v in by_cu.values() if len(v) == 2)
print(f" first round: {len(by_cu)} CUs hold {sum(len(v) for v in by_cu.values())} workgroups; co-resident pairs differ in blockIdx >> 3 by {dict(gaps)}; "
      f"a workgroup's four waves sit on {Counter(len(set(simd[b])) for b in order[:512]).most_common(1)[0][0]} different SIMDs")
