"""float64 verification variant of the env step on the GPU (include/dronesim_verify.h, libdronesim_verify.so:
dronesim_step_f64 / dronesim_observe_f64).

The reference is float64 throughout (drone_env.py:189) while the product kernels compute in float32.  `F64Env` runs
the same per-pair arithmetic (one scalar-type template in csrc/common.hpp, instantiated for double) and the same
epilogue semantics on float64 device buffers, slowly and simply (one workgroup per env, every ordered pair).  It is
TEST INFRASTRUCTURE for the float32 kernels -- golden vectors without a float32-state allowance, free-running
episodes against the float64 oracle, on-device float64 judgement of float32 outputs -- not a second product path:
no rollout, no episode layer, no reset kernel (initial states are injected with `set_state`)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .drone_env import DONE_RADIUS, GHOST_FACTOR, clip_deltas, dt, formation_O, max_time_steps


class F64Env:
    """Batched float64 mirror of `drones` for state injection + step()/observe() only."""

    def __init__(self, n_agents, grid, k_closest=2, deltas=None, simplify_zstate=False, *, n_envs=1, device=None,
                 collision_weight=0.2, drone_radius=None):
        import torch
        from . import _native
        self._torch, self._native, self._lib = torch, _native, _native.verify_lib()
        if not torch.cuda.is_available():
            raise RuntimeError("F64Env needs a ROCm GPU (there is no CPU path in this package)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n_agents, self.k_closest, self.n_envs = int(n_agents), int(k_closest), int(n_envs)
        self.c = 2 if simplify_zstate else 5
        self.collision_weight = float(collision_weight)
        N, E, K1 = self.n_agents, self.n_envs, self.k_closest + 1
        radius = np.full(N, 0.1) if drone_radius is None else np.asarray(drone_radius, np.float64)
        self.end_points, self.d_safety = formation_O(N, grid, radius)
        self.deltas = clip_deltas(deltas, self.d_safety, warn=False)
        f64 = dict(dtype=torch.float64, device=self.device)
        self._xF = torch.tensor(self.end_points.reshape(N, 2), **f64).contiguous()
        self._d_hat = torch.tensor(self.d_safety, **f64)
        self._delta = torch.tensor(np.asarray(self.deltas, np.float64), **f64)
        self._radius = torch.tensor(radius, **f64)
        self.pos = torch.zeros(E, N, 2, **f64); self.vel = torch.zeros(E, N, 2, **f64)
        self.t = torch.zeros(E, dtype=torch.int32, device=self.device)
        self.reward = torch.zeros(E, N, **f64); self.true_reward = torch.zeros(E, N, **f64)
        self.z = torch.zeros(E, N, K1 * self.c, **f64)
        self.nbr_idx = torch.full((E, N, K1), -1, dtype=torch.int32, device=self.device)
        self.n_coll = torch.zeros(E, dtype=torch.int32, device=self.device)
        self.done = torch.zeros(E, dtype=torch.uint8, device=self.device)

    def _params(self):
        p = self._native.DroneParamsF64()
        p.N, p.k, p.c, p.max_steps = self.n_agents, self.k_closest, self.c, max_time_steps
        p.dt, p.q, p.b = dt, 2 * dt, self.collision_weight * dt          # drone_env.py:269-270
        p.done_radius, p.ghost_factor = DONE_RADIUS, GHOST_FACTOR
        p.xF, p.d_hat = self._xF.data_ptr(), self._d_hat.data_ptr()
        p.delta, p.radius = self._delta.data_ptr(), self._radius.data_ptr()
        return p

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def set_state(self, pos, vel=None, t=None):
        """Inject a float64 state and refresh rewards / z / Ni on it (drones.rewards(), drone_env.py:260-293)."""
        torch = self._torch
        E, N = self.n_envs, self.n_agents
        self.pos.copy_(torch.as_tensor(np.asarray(pos, np.float64) if not torch.is_tensor(pos) else pos, dtype=torch.float64).reshape(E, N, 2))
        if vel is None:
            self.vel.zero_()
        else:
            self.vel.copy_(torch.as_tensor(np.asarray(vel, np.float64) if not torch.is_tensor(vel) else vel, dtype=torch.float64).reshape(E, N, 2))
        if t is not None:
            tt = torch.as_tensor(np.asarray(t, np.int32) if not torch.is_tensor(t) else t, dtype=torch.int32)
            self.t.copy_(tt.reshape(-1).expand(E) if tt.numel() == 1 else tt.reshape(E))
        self.observe()

    def observe(self):
        p = self._params()
        with self._torch.cuda.device(self.device):
            rc = self._lib.dronesim_observe_f64(C.byref(p), self.pos.data_ptr(), self.vel.data_ptr(), self.reward.data_ptr(),
                                                self.true_reward.data_ptr(), self.z.data_ptr(), self.nbr_idx.data_ptr(),
                                                self.n_coll.data_ptr(), self.n_envs, self._stream())
        self._native.check_verify(rc, "dronesim_observe_f64")

    def step(self, actions):
        """drones.step() in float64 (drone_env.py:214-258): ``actions [E,N,2]`` float64 device tensor."""
        torch = self._torch
        act = torch.as_tensor(actions, dtype=torch.float64, device=self.device).contiguous()
        if tuple(act.shape) != (self.n_envs, self.n_agents, 2):
            raise ValueError(f"actions must be [{self.n_envs},{self.n_agents},2], got {tuple(act.shape)}")
        p = self._params()
        with torch.cuda.device(self.device):
            rc = self._lib.dronesim_step_f64(C.byref(p), self.pos.data_ptr(), self.vel.data_ptr(), self.t.data_ptr(),
                                             act.data_ptr(), self.reward.data_ptr(), self.true_reward.data_ptr(),
                                             self.z.data_ptr(), self.nbr_idx.data_ptr(), self.n_coll.data_ptr(),
                                             self.done.data_ptr(), self.n_envs, self._stream())
        self._native.check_verify(rc, "dronesim_step_f64")
        return self
