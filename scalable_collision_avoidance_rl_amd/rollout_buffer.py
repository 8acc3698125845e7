"""On-device experience storage and its reductions.

The reference appends one namedtuple per agent and step to Python deques (`ExperienceBuffers`, utils.py:232-253,
called at train_problem.py:96) and scans them per agent at the end of the episode.  Here:

  RolloutStorage       utils.py:232-253        `[T, E, N, ...]` device tensors that the STEP KERNEL ITSELF fills:
                                               `drones.step(act, into=(storage, t))` hands slot t's addresses to
                                               `dronesim_step_ex`, so storing a transition costs no launch and no copy
  mc_returns           SAC_agents.py:304-307   G[t] = r[t] + gamma * G[t+1]
  neighbour_advantage  SAC_agents.py:333-351   gamma^t / N * sum_{j in Ni[t]} (G_j[t] - V_i[t])

The reductions run in the HIP library (dronesim_returns / dronesim_advantage); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

# the reference's experience tuple (utils.py:241-242), field for field
Experience = namedtuple("experience", ["z_state", "action", "reward", "next_z", "Ni", "finished"])


class RolloutStorage:
    """T steps of experience of a batched `drones` env, on its device, written by the step launches themselves.

    What `ExperienceBuffers.append(z_states, actions, rewards, new_z, Ni, finished)` (utils.py:244-249) stores per
    agent and step, as tensors over (step, env, agent):

        z_state  -> ``z_pre[t]``      the observation the action was based on   [T,E,N,(k+1)c]
        action   -> ``actions[t]``                                               [T,E,N,2]
        reward   -> ``reward[t]``     (+ ``true_reward[t]``, ``n_coll[t]``)      [T,E,N]
        next_z   -> ``next_z()[t]``   the observation after the step            [T,E,N,(k+1)c]
        Ni       -> ``nbr_pre[t]``    neighbour ids of ``z_pre`` (slot 0 = i, -1 = none)   [T,E,N,k+1]
        finished -> ``done[t]``                                                  [T,E]

    Observations live in ONE ring of T+1 slots: the step into slot t writes its new observation at ring index t+1, so
    ``z_pre`` (= ring[:T]) and the raw post-step observation ``z`` (= ring[1:]) are views of the same memory and the
    pre-step observation is never copied.  With ``auto_reset`` the post-step observation of a step that ends an episode
    is the NEW episode's first one (which is the right ``z_pre`` of the next step); the finished episode's terminal
    observation -- the reference's ``new_z`` for that transition (drone_env.py:258) -- is written by the same launch
    to ``z_final[t]`` / ``nbr_final[t]`` and merged by ``next_z()``.

        storage = RolloutStorage(env, T)
        storage.begin()                                   # slot 0 of the ring <- the env's current observation
        #                                                   (optional: a step into slot t carries the env's current
        #                                                   observation into ring slot t itself when it lives elsewhere,
        #                                                   e.g. after a manual `env.reset(mask=...)` between steps)
        for t in range(T):
            policy.sample_action(env.z, env=env, act_out=storage.actions[t])
            env.step(storage.actions[t], into=(storage, t))
        G = storage.returns(gamma); w = storage.advantage(storage.values, gamma)
    """

    def __init__(self, env, T, actions=True, values=False):
        import torch
        if not getattr(env, "batched", False):
            raise ValueError("RolloutStorage belongs to the batched (tensor) API of `drones`")
        self.env, self.T = env, int(T)
        E, N, K1, c = env.n_envs, env.n_agents, env.k_closest + 1, env.c
        dev = env.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.zbuf = torch.zeros(self.T + 1, E, N, K1 * c, **f32)
        self.nbrbuf = torch.full((self.T + 1, E, N, K1), -1, dtype=torch.int32, device=dev)
        self.reward = torch.zeros(self.T, E, N, **f32)
        self.true_reward = torch.zeros(self.T, E, N, **f32)
        self.n_coll = torch.zeros(self.T, E, dtype=torch.int32, device=dev)
        self.done = torch.zeros(self.T, E, dtype=torch.uint8, device=dev)
        self.actions = torch.zeros(self.T, E, N, 2, **f32) if actions else None
        self.values = torch.zeros(self.T, E, N, **f32) if values else None
        final = bool(env.auto_reset)
        self.z_final = torch.zeros(self.T, E, N, K1 * c, **f32) if final else None
        self.nbr_final = torch.full((self.T, E, N, K1), -1, dtype=torch.int32, device=dev) if final else None
        self._slots = [None] * self.T
        self._generation = None

    # views over the observation ring
    z_pre = property(lambda self: self.zbuf[:self.T])
    nbr_pre = property(lambda self: self.nbrbuf[:self.T])
    z = property(lambda self: self.zbuf[1:])                 # raw post-step observation (new episode's first one after a reset)
    nbr_idx = property(lambda self: self.nbrbuf[1:])

    def begin(self):
        """Start (or restart) filling at slot 0: the env's current observation becomes ring slot 0 (ONE copy per T
        steps, none when the env is already bound there) and the env's observation attributes are bound to it."""
        env = self.env
        if env.z.data_ptr() != self.zbuf[0].data_ptr():
            self.zbuf[0].copy_(env.z)
            self.nbrbuf[0].copy_(env.nbr_idx)
        views = dict(reward=env.reward, true_reward=env.true_reward, z=self.zbuf[0], nbr_idx=self.nbrbuf[0],
                     n_coll=env.n_coll, done=env.done, z_final=env.z_final, nbr_final=env.nbr_final,
                     pos_final=env.pos_final)
        env._bind(views, home=False)
        return self

    def _slot(self, env, t):
        """(pre-marshalled DroneStepCall, attribute views) of slot t -- built once, reused by every later pass."""
        if env is not self.env:
            raise ValueError("this RolloutStorage belongs to another env")
        if not (0 <= t < self.T):
            raise IndexError(f"slot {t} outside a storage of {self.T} steps")
        gen = getattr(env, "_ctl_generation", 0)
        if gen != self._generation:                         # (load_state changed the seed the ctls carry)
            self._slots, self._generation = [None] * self.T, gen
        if self._slots[t] is None:
            zf = None if self.z_final is None else self.z_final[t]
            nf = None if self.nbr_final is None else self.nbr_final[t]
            views = dict(reward=self.reward[t], true_reward=self.true_reward[t], z=self.zbuf[t + 1],
                         nbr_idx=self.nbrbuf[t + 1], n_coll=self.n_coll[t], done=self.done[t],
                         z_final=zf, nbr_final=nf, pos_final=env._home["pos_final"],
                         actions=None if self.actions is None else self.actions[t],
                         _pre=(self.zbuf[t], self.nbrbuf[t], self.zbuf[t].data_ptr()))
            ctl = env._make_ctl(zf, nf, env._home["pos_final"]) if env._use_ctl else None
            env._params()
            self._slots[t] = (env._make_call(views, ctl), views)   # (the call keeps its ctl object alive)
        return self._slots[t]

    def next_z(self):
        """``new_z`` of every transition as the reference stores it (utils.py:244-249): the post-step observation,
        with the TERMINAL observation substituted where the step ended an episode under auto_reset.
        Returns ``(z [T,E,N,(k+1)c], nbr_idx [T,E,N,k+1])``."""
        import torch
        if self.z_final is None:
            return self.z, self.nbr_idx
        fin = self.done.bool()[:, :, None, None]
        return torch.where(fin, self.z_final, self.z), torch.where(fin, self.nbr_final, self.nbr_idx)

    def returns(self, gamma, true_rewards=False):
        """Monte-Carlo returns of the stored rewards, restarting at episode ends (`mc_returns`)."""
        return mc_returns(self.true_reward if true_rewards else self.reward, gamma, self.done)

    def advantage(self, V, gamma, G=None):
        """Actor-loss weights from the stored pre-step neighbour lists (`neighbour_advantage`)."""
        G = self.returns(gamma) if G is None else G
        return neighbour_advantage(G, V, self.nbr_pre, gamma, self.done)

    def experience(self, t, i, e=0):
        """The reference's namedtuple for (step t, agent i) of env e, host-side (tests / debugging): flattened float64
        z rows, action, reward, next_z, ``Ni`` as the reference's list (i first, ghost slots dropped), finished."""
        nz, _ = self.next_z()
        nb = self.nbr_pre[t, e, i].cpu().numpy()
        act = None if self.actions is None else self.actions[t, e, i].double().cpu().numpy()
        return Experience(self.z_pre[t, e, i].double().cpu().numpy(), act, float(self.reward[t, e, i]),
                          nz[t, e, i].double().cpu().numpy(), [int(j) for j in nb if j >= 0], bool(self.done[t, e]))

    def __len__(self):
        return self.T



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
