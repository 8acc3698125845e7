"""On-device rollout storage reductions: the learner-side scans the reference runs
per agent in Python over its `ExperienceBuffers` deques (utils.py:232-253), here over the `[T,E,N]`
tensors that `drones.rollout()` / T calls of `drones.step()` leave on the device.

  mc_returns           SAC_agents.py:304-307   G[t] = r[t] + gamma * G[t+1]
  neighbour_advantage  SAC_agents.py:333-351   gamma^t / N * sum_{j in Ni[t]} (G_j[t] - V_i[t])

Both run in the HIP library (dronesim_returns / dronesim_advantage); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C


def _prep(x, dtype, shape=None):
    import torch
    if not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError("rollout_buffer works on ROCm device tensors only (no CPU fallback)")
    x = x.to(dtype).contiguous()
    if shape is not None and tuple(x.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(x.shape)}")
    return x


def mc_returns(reward, gamma: float, done=None):
    """Monte-Carlo returns of every (env, agent) column of ``reward [T,E,N]``; ``done [T,E]`` (optional)
    marks steps that ended an episode (the scan restarts there)."""
    import torch
    from . import _native
    lib = _native.lib()
    reward = _prep(reward, torch.float32)
    T, E, N = reward.shape
    d = None if done is None else _prep(done, torch.uint8, (T, E))
    G = torch.empty_like(reward)
    with torch.cuda.device(reward.device):
        rc = lib.dronesim_returns(reward.data_ptr(), None if d is None else d.data_ptr(), float(gamma),
                                  G.data_ptr(), T, E, N, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _native.check(rc, "dronesim_returns")
    return G


def neighbour_advantage(G, V, nbr_idx, gamma: float, done=None):
    """Actor-loss weight ``gamma^t / N * sum_{j in Ni} (G_j - V_i)`` for ``G, V [T,E,N]`` and the neighbour
    lists ``nbr_idx [T,E,N,k+1]`` the actions were based on (the PRE-step observation: `env.nbr_idx`
    before each step, i.e. ``rollout()['nbr_idx_pre']``)."""
    import torch
    from . import _native
    lib = _native.lib()
    G = _prep(G, torch.float32)
    T, E, N = G.shape
    V = _prep(V, torch.float32, (T, E, N))
    nbr_idx = _prep(nbr_idx, torch.int32)
    if nbr_idx.dim() != 4 or tuple(nbr_idx.shape[:3]) != (T, E, N):
        raise ValueError(f"nbr_idx must be [T,E,N,k+1], got {tuple(nbr_idx.shape)}")
    d = None if done is None else _prep(done, torch.uint8, (T, E))
    w = torch.empty_like(G)
    with torch.cuda.device(G.device):
        rc = lib.dronesim_advantage(G.data_ptr(), V.data_ptr(), nbr_idx.data_ptr(), None if d is None else d.data_ptr(),
                                    float(gamma), w.data_ptr(), T, E, N, int(nbr_idx.shape[3]),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _native.check(rc, "dronesim_advantage")
    return w
