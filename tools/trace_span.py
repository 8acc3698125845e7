#!/usr/bin/env python3
"""Device-side duration of the step launches INSIDE a hipGraph replay, from the waves' own entry / exit times on the chip-wide
100 MHz clock (needs a -DDRONESIM_TRACE_SPAN build, selected with DRONESIM_LIB): per launch, first wave in -> last wave out
(`span`), first wave in -> the next launch's first wave in (`period` = what a step costs the device) and the idle gap between
consecutive launches.  The same replay is timed with HIP events for comparison.  This is the kernel-duration evidence that is
consistent with the live bench run: rocprofv3 serialises the launches and reports averages ABOVE the whole per-step time.

    make -C scalable_collision_avoidance_rl_amd/csrc -j8 EXTRA=-DDRONESIM_TRACE_SPAN OUT=../../abl/span.so OBJDIR=../../build/obj_span
    DRONESIM_LIB=abl/span.so python tools/trace_span.py [c3|c5|c2] [launches]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scalable_collision_avoidance_rl_amd import _native, drones
from tools.kbench import PRESETS

spec = sys.argv[1] if len(sys.argv) > 1 else "c3"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, E, G, delta = PRESETS[spec]
layer = not os.environ.get("PLAIN")
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1,
             auto_reset=layer, track_episodes=layer)
lib = _native.lib()
lib.dronesim_debug_set_trace.argtypes = [C.c_void_p]
waves = E * ((N + 63) // 64) if N > 64 else (E + (64 // N) - 1) // (64 // N)
buf = torch.zeros(L, waves, dtype=torch.int64, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
pool = torch.rand(L, E, N, 2, device="cuda", generator=g) * 2 - 1
for s in range(5):
    env.step(pool[s])
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for s in range(L):
        lib.dronesim_debug_set_trace(buf[s].data_ptr())      # (read by the library when the launch is captured)
        env.step(pool[s])
lib.dronesim_debug_set_trace(None)
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
ev = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
    ev.append(a.elapsed_time(b) / L * 1e3)
t = buf.cpu().numpy()
ent = (t & 0xffffffff).astype(np.int64); ext = ((t >> 32) & 0xffffffff).astype(np.int64)
first_in, last_out = ent.min(1) * 0.01, ext.max(1) * 0.01                   # us
med_out = np.median(ext, axis=1) * 0.01
span, period, gap = last_out - first_in, np.diff(first_in), first_in[1:] - last_out[:-1]
q = lambda x: f"median {np.median(x):.2f} (p10 {np.percentile(x, 10):.2f}, p90 {np.percentile(x, 90):.2f})"
print(f"{spec}{' with the episode layer' if layer else ' plain'}: {L} step launches in one hipGraph, {waves} waves each; us on the 100 MHz chip clock (10 ns resolution)")
print(f"  span   first wave in -> last wave out : {q(span[1:])}")
print(f"  median wave out - first wave in       : {q((med_out - first_in)[1:])}")
print(f"  gap    last wave out -> next first in : {q(gap)}")
print(f"  period first wave in -> next first in : {q(period)}    <- the device-side cost of a step")
print(f"  HIP events around the same replay     : {np.median(ev):.2f} us per step")
import json
print("SPAN_JSON " + json.dumps({"spec": spec, "episode_layer": bool(layer), "launches": L, "waves_per_launch": waves,
                                 "span_us": float(np.median(span[1:])), "median_wave_out_us": float(np.median((med_out - first_in)[1:])),
                                 "boundary_us": float(np.median(gap)), "period_us": float(np.median(period)),
                                 "hip_event_us_per_step": float(np.median(ev))}))
