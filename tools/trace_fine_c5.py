import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from scalable_collision_avoidance_rl_amd import _native, drones
from tools.kbench import PRESETS
spec = sys.argv[1]
N, E, G, delta = PRESETS[spec]
env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1)
lib = _native.lib()
waves = E * ((N + 63) // 64)
trace = torch.zeros(waves, 8, dtype=torch.int64, device="cuda")
act = torch.rand(E, N, 2, device="cuda") * 2 - 1
for _ in range(5): env.step(act)
torch.cuda.synchronize()
lib.dronesim_debug_set_trace.argtypes = [C.c_void_p]
g = torch.cuda.CUDAGraph()
lib.dronesim_debug_set_trace(trace.data_ptr())
with torch.cuda.graph(g):
    for _ in range(20): env.step(act)
lib.dronesim_debug_set_trace(None)
g.replay(); g.replay(); torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.float64)
order = [0, 1, 4, 5, 2, 3, 6]
names = ["entry", "tables built", "masks combined", "first candidates tested", "all candidates tested", "pairs done", "stores acked"]
for a, b in zip(range(6), range(1, 7)):
    d = t[:, order[b]] - t[:, order[a]]
    print(f"  {names[a]:>26} -> {names[b]:<26}: median {np.median(d):7.0f}  p95 {np.percentile(d, 95):7.0f}")
