"""MI355X-native batched `drones` environment: drop-in for the reference class.

Host-side mirror of `/root/reference/drone_env.py:53-401` (class `drones`) for the
step()/reset()/get_local_states() hot path.  Everything per-timestep runs in the
hand-written HIP kernels of csrc/drone_kernel.hpp (step / observe / rollout) and
csrc/dronesim.hip (reset, controllers, reductions) through the C ABI of include/dronesim.h; this file only owns the caller-facing surface:

* construction-time constants (goal ring, safety distance, Delta clip) computed once
  on the host exactly as the reference does (drone_env.py:83-91, 115-153),
* device buffers (torch tensors = HBM allocations) and the current HIP stream,
* the two faces of the API:
    - ``n_envs == 1`` (default): the reference's own Python types -- ``state`` a live
      ``[N,5]`` float64 ndarray, ``z_states`` a list of ``[k+1,c]`` ndarrays, ``Ni`` a
      list of int lists, ``step()`` returning the reference 6-tuple
      (drone_env.py:258) -- so `SAC_agents.py` / `train_problem.py:82-107` loops run
      unchanged;
    - ``n_envs > 1``: the same method names on ``[E, ...]`` device tensors with no
      host synchronisation per step.

There is no CPU fallback: constructing an env without the built HIP library or
without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import numpy as np

# module constants of the reference (drone_env.py:27-30)
dim = 2
dt = 0.05
max_time_steps = 200

DRONE_RADIUS = 0.1          # drone_env.py:75, 174
DONE_RADIUS = 0.2           # drone_env.py:251
GHOST_FACTOR = 1.1          # drone_env.py:386
LATTICE_PITCH = 2 * 1.1 * DRONE_RADIUS   # drone_env.py:193


# --------------------------------------------------------------------------------------
# host-side, once-per-env precompute (pure NumPy; testable without a GPU)

def formation_O(n_agents: int, grid, drone_radius=None):
    """Goal ring and safety distance of end_formation == "O" (drone_env.py:124-153).

    Returns ``(end_points [2N,1] column, d_safety [N])`` in float64 like the reference."""
    n = int(n_agents)
    gx, gy = float(grid[0]), float(grid[1])
    radius = np.full(n, DRONE_RADIUS) if drone_radius is None else np.asarray(drone_radius, np.float64)
    ang = np.arange(n) * (2 * np.pi / n)
    xF = np.stack([np.cos(ang) * 0.9 * gx / 2 + gx / 2, np.sin(ang) * 0.9 * gy / 2 + gy / 2], axis=1)
    diff = xF[:, None, :] - xF[None, :, :]
    gap = np.sqrt((diff ** 2).sum(-1)) - radius[:, None] - radius[None, :]
    np.fill_diagonal(gap, np.inf)
    d_safety = np.floor(gap.min(axis=1) * 100) / 100
    return xF.reshape(2 * n, 1), d_safety


def clip_deltas(deltas, d_safety, warn=True):
    """deltas=None -> d_safety; else min(deltas, d_safety) with the reference's warning (drone_env.py:85-91)."""
    if deltas is None:
        return d_safety
    deltas = np.asarray(deltas, np.float64)
    out = np.minimum(deltas, d_safety)
    if warn and not np.all(deltas <= d_safety):
        print("Some deltas are greater than the final minimum distance between end positions. Using minimum "
              "distance between end positions for those cases instead.", f"deltas = {out}")
    return out


def lattice_divisions(grid):
    """Nodes per axis of the initial-position lattice (drone_env.py:193-194)."""
    d = np.floor(np.array(grid, np.float64) / LATTICE_PITCH)
    return int(d[0]), int(d[1])


def shard_range(n_envs: int, rank: int, world_size: int):
    """Contiguous slice [lo, hi) of the env axis owned by `rank` (envs never interact)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(int(n_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


StepResult = namedtuple("StepResult", "state z_states rewards n_collisions finished true_rewards")


class DroneState:
    """Batched state view: ``pos``/``vel`` ``[E,N,2]`` device tensors (live), ``radius`` ``[N]``."""
    __slots__ = ("pos", "vel", "radius")

    def __init__(self, pos, vel, radius):
        self.pos, self.vel, self.radius = pos, vel, radius

    def tensor(self):
        """``[E,N,5]`` copy in the reference's column order x, y, vx, vy, l (drone_env.py:173)."""
        import torch
        E, N = self.pos.shape[:2]
        return torch.cat([self.pos, self.vel, self.radius.view(1, N, 1).expand(E, N, 1)], dim=2)

    def copy(self):
        return self.tensor()


class drones:
    """Batched multi-agent 2-D formation / collision-avoidance environment on MI355X.

    Signature and attributes follow the reference (drone_env.py:55); keyword-only
    additions: ``n_envs`` (E parallel, independent envs), ``device``, ``seed``,
    ``rank``/``world_size`` (shard the env axis across one-process-per-GPU ranks),
    ``batched`` (force tensor API at E == 1), ``track_episodes`` (keep the per-episode sums the
    rollout loop logs, train_problem.py:98-100, in per-env device records updated by the step kernel
    itself), ``auto_reset`` (envs whose ``done`` fires are reset and re-observed inside the same step
    launch -- what train_problem.py:132 does after the ``while not finished`` loop; implies
    ``track_episodes``), ``keep_final_obs`` (with ``auto_reset``: the finished episode's terminal observation and
    state -- what the reference's ``step()`` returns on its last call, drone_env.py:258, and its loop stores as
    ``new_z``, utils.py:244-249 -- are kept in ``z_final`` / ``nbr_final`` / ``pos_final`` instead of being lost to
    the new episode's first observation; rows are valid for the envs whose ``finished`` flag the step raised)."""

    def __init__(self, n_agents: int, n_obstacles: int, grid: list, end_formation: str, k_closest=2,
                 deltas: np.ndarray = None, simplify_zstate=False, *, n_envs: int = 1, device=None,
                 seed: int = None, rank: int = 0, world_size: int = 1, batched: bool = None,
                 track_episodes: bool = None, auto_reset: bool = False, keep_final_obs: bool = False) -> None:
        import torch
        from . import _native

        self._torch = torch
        self._native = _native
        self._lib = _native.lib()                     # raises ImportError if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("scalable_collision_avoidance_rl_amd.drones needs a ROCm GPU (MI355X); "
                               "there is no CPU fallback for the step()/reset() path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"device must be a ROCm GPU, got {self.device}")
        if self.device.index is None:                 # "cuda" -> "cuda:<current>": tensors always carry an index
            self.device = torch.device("cuda", torch.cuda.current_device())

        self.n_agents = int(n_agents)
        self.grid = grid
        self.goal = self.grid
        self.k_closest = int(k_closest)
        self.simplify_zstate = bool(simplify_zstate)
        self.internal_t = 0
        self.collision_weight = 0.2                   # live attribute, read at every step (drone_env.py:72, 270)
        self._alloc_done = False
        self._deltas = None
        self._set_const("_drone_radius", np.ones(self.n_agents) * DRONE_RADIUS, "_radius")
        self.auto_reset = bool(auto_reset)
        self.keep_final_obs = bool(keep_final_obs)
        if self.keep_final_obs and not self.auto_reset:
            raise ValueError("keep_final_obs only has a meaning with auto_reset (without it z_states IS the terminal "
                             "observation until the caller resets)")
        self.track_episodes = self.auto_reset if track_episodes is None else bool(track_episodes) or self.auto_reset
        self.A = np.eye(dim)
        self.B = np.eye(dim) * dt

        N, k = self.n_agents, self.k_closest
        if N < 2 or N > _native.MAX_AGENTS:
            raise ValueError(f"n_agents must be in 2..{_native.MAX_AGENTS}")
        if not (1 <= k <= min(N - 1, _native.MAX_K)):
            raise ValueError(f"k_closest must be in 1..min(n_agents-1, {_native.MAX_K}) "
                             "(the reference raises IndexError for k_closest >= n_agents)")
        if end_formation != "O":
            raise ValueError(str(end_formation) + " is Not a valid end formation identifier")

        self.obstacles = self.create_obstacles(n_obstacles)
        self.end_points, d_safety = formation_O(N, grid, self.drone_radius)
        self._set_const("_d_safety", d_safety, "_d_hat")
        if not np.all(self.d_safety > 0):
            raise ValueError(f"safety distance d_hat = {self.d_safety.min():.2f} <= 0: the goal ring does not fit "
                             f"{N} agents on grid {grid} (need 0.9*G*sin(pi/N) > 0.2); the reference's reward is "
                             "degenerate there")
        self._set_const("_deltas", self.d_safety if deltas is None else deltas, "_delta")   # clipped, with the reference's warning

        # sharding of the env axis
        self.n_envs_global = int(n_envs)
        self.rank, self.world_size = int(rank), int(world_size)
        self.env_lo, self.env_hi = shard_range(self.n_envs_global, self.rank, self.world_size)
        self.n_envs = self.env_hi - self.env_lo
        if self.n_envs < 1:
            raise ValueError("this rank owns no environments")
        self.batched = (self.n_envs_global > 1) if batched is None else bool(batched)
        if self.auto_reset and not self.batched:
            raise ValueError("auto_reset needs the batched (tensor) API: the reference-typed E = 1 face mirrors "
                             "the reference, whose caller resets explicitly (train_problem.py:132)")
        self.seed = int(np.random.SeedSequence().entropy & 0xFFFFFFFFFFFFFFFF) if seed is None else int(seed)

        self.global_state_space = N * (2 * dim + 1)
        self.c = dim if self.simplify_zstate else 2 * dim + 1
        self.local_state_space = self.c * (1 + k)     # drone_env.py:180-184
        self.global_action_space = N * dim
        self.local_action_space = dim

        # per-step host path: current device / raw stream handle without building Stream objects
        self._dev_index = self.device.index
        self._step_call = self._lib.dronesim_step_call
        self._cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._raw_stream = raw if raw is not None else (lambda i: torch.cuda.current_stream(i).cuda_stream)
        self._alloc()
        self.reset(renew_obstacles=False)

    # ------------------------------------------------------------------ live constants
    # The reference reads self.deltas / self.d_safety / self.drone_radius on every step (drone_env.py:242): assigning
    # them here re-uploads the device copies and takes effect at the next launch.
    def _set_const(self, name, value, tensor_name):
        # a scalar broadcasts over the agents; `deltas` is clipped against d_safety exactly as the constructor does
        # (drone_env.py:85-91); the stored array is READ-ONLY so that an in-place edit (`env.deltas[i] = x`), which the
        # reference would pick up on its next step but which cannot reach the device copy, raises instead of being lost
        arr = np.array(np.broadcast_to(np.asarray(value, np.float64), (self.n_agents,)))
        if name == "_deltas":
            arr = np.array(clip_deltas(arr, self.d_safety))
        arr.setflags(write=False)
        setattr(self, name, arr)
        if name == "_d_safety" and getattr(self, "_deltas", None) is not None and not np.all(self._deltas <= arr):
            self._set_const("_deltas", self._deltas, "_delta")       # keep Delta <= d_hat (the constructor's invariant)
        if self._alloc_done:
            t = getattr(self, tensor_name)
            t.copy_(self._torch.as_tensor(arr.copy(), dtype=self._torch.float32))
            self._params_cache = None

    deltas = property(lambda self: self._deltas, lambda self, v: self._set_const("_deltas", v, "_delta"))
    d_safety = property(lambda self: self._d_safety, lambda self, v: self._set_const("_d_safety", v, "_d_hat"))
    drone_radius = property(lambda self: self._drone_radius, lambda self, v: self._set_const("_drone_radius", v, "_radius"))

    # ------------------------------------------------------------------ buffers / params
    def _alloc(self):
        torch = self._torch
        E, N, K1, c = self.n_envs, self.n_agents, self.k_closest + 1, self.c
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self._xF = torch.tensor(self.end_points.reshape(N, 2), **f32).contiguous()
        # what float32 drops of the float64 goal ring: the kernels subtract it after the (exact) float32 difference
        self._xF_lo = torch.tensor(self.end_points.reshape(N, 2) - self._xF.double().cpu().numpy(), **f32).contiguous()
        self._d_hat = torch.tensor(self.d_safety, **f32)
        self._delta = torch.tensor(np.asarray(self.deltas, np.float64), **f32)
        self._radius = torch.tensor(self.drone_radius, **f32)
        self.pos = torch.zeros(E, N, 2, **f32)
        self.vel = torch.zeros(E, N, 2, **f32)
        self.t = torch.zeros(E, dtype=torch.int32, device=dev)
        self.episode = torch.zeros(E, dtype=torch.int32, device=dev)   # resets seen per env (RNG stream id)
        # per-step outputs (overwritten by every step) and current observation
        self.reward = torch.zeros(E, N, **f32)
        self.true_reward = torch.zeros(E, N, **f32)
        self.z = torch.zeros(E, N, K1 * c, **f32)
        self.nbr_idx = torch.full((E, N, K1), -1, dtype=torch.int32, device=dev)
        self.n_coll = torch.zeros(E, dtype=torch.int32, device=dev)
        self.done = torch.zeros(E, dtype=torch.uint8, device=dev)
        self._act = torch.zeros(E, N, 2, **f32)
        self._act_shape = self._act.shape
        # terminal observation / state of the envs an auto-reset launch finishes (DroneEpisodeCtl.z_final ...)
        self.z_final = torch.zeros(E, N, K1 * c, **f32) if self.keep_final_obs else None
        self.nbr_final = torch.full((E, N, K1), -1, dtype=torch.int32, device=dev) if self.keep_final_obs else None
        self.pos_final = torch.zeros(E, N, 2, **f32) if self.keep_final_obs else None
        # the env's own output buffers; step(..., into=(storage, t)) re-binds the attributes to storage slots
        self._home = dict(reward=self.reward, true_reward=self.true_reward, z=self.z, nbr_idx=self.nbr_idx,
                          n_coll=self.n_coll, done=self.done, z_final=self.z_final, nbr_final=self.nbr_final,
                          pos_final=self.pos_final)
        self._bound_home = True
        self._z_ptr = self._home["_zptr"] = self.z.data_ptr()
        self._params_cache = None
        self._step_args = None
        # episode bookkeeping (include/dronesim.h: DroneEpisodeAcc, one 64-byte record per env) -- off unless asked for
        self.episode_acc = torch.zeros(E, 8, dtype=torch.float64, device=dev) if self.track_episodes else None
        self._episode_totals = torch.zeros(8, dtype=torch.float64, device=dev) if self.track_episodes else None
        self._ctl_cache = None
        self._alloc_done = True
        self.state = DroneState(self.pos, self.vel, self._radius) if self.batched else None
        # step() hands back the same live tensors every call
        self._result = StepResult(self.state, self.z, self.reward, self.n_coll, self.done, self.true_reward)
        self._home["_result"] = self._result

    def _params(self):
        """DroneParams for this launch (collision_weight is live: train_problem.py:31)."""
        P = self._native.DroneParams
        key = float(self.collision_weight)
        if self._params_cache is None or self._params_cache[0] != key:
            p = P()
            p.N, p.k, p.c, p.max_steps = self.n_agents, self.k_closest, self.c, max_time_steps
            p.dt, p.q, p.b = dt, 2 * dt, key * dt                      # drone_env.py:269-270
            p.done_radius, p.ghost_factor = DONE_RADIUS, GHOST_FACTOR
            # float32 images of the arrays (what the kernel compares) bound the variant choice
            p.d_hat_min = float(self._d_hat.min().item())
            p.d_hat_max = float(self._d_hat.max().item())
            p.delta_min = float(self._delta.min().item())
            p.delta_max = float(self._delta.max().item())
            p.radius_min = float(self._radius.min().item())
            p.radius_max = float(self._radius.max().item())
            p.xF, p.d_hat = self._xF.data_ptr(), self._d_hat.data_ptr()
            p.xF_lo = self._xF_lo.data_ptr()
            p.delta, p.radius = self._delta.data_ptr(), self._radius.data_ptr()
            self._params_cache = (key, p)
            self._p_addr = C.addressof(p)             # what the pre-marshalled step calls point at
        return self._params_cache[1]

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _make_call(self, views, ctl):
        """dronesim_step_ex's arguments for one output binding, marshalled ONCE (include/dronesim.h: DroneStepCall):
        step() then makes a 3-argument call (call, actions, stream).  Returns ``[struct, address, ctl]`` -- the ctl
        object is kept alive next to the struct that points at it."""
        c = self._native.DroneStepCall()
        c.p = self._p_addr
        c.ctl = None if ctl is None else C.addressof(ctl)
        c.pos, c.vel, c.t = self.pos.data_ptr(), self.vel.data_ptr(), self.t.data_ptr()
        c.reward, c.true_reward = views["reward"].data_ptr(), views["true_reward"].data_ptr()
        c.z, c.nbr_idx = views["z"].data_ptr(), views["nbr_idx"].data_ptr()
        c.n_coll, c.done, c.E = views["n_coll"].data_ptr(), views["done"].data_ptr(), self.n_envs
        return [c, C.addressof(c), ctl]

    def _make_ctl(self, z_final=None, nbr_final=None, pos_final=None):
        """A DroneEpisodeCtl of this env with the given terminal-observation targets (device tensors or None)."""
        c = self._native.DroneEpisodeCtl()
        c.acc = self.episode_acc.data_ptr() if self.episode_acc is not None else None
        c.auto_reset = 1 if self.auto_reset else 0
        c.div_x, c.div_y = lattice_divisions(self.grid)
        c.pitch = float(LATTICE_PITCH)
        c.seed, c.env_base = self.seed, self.env_lo
        c.episode = self.episode.data_ptr()
        c.z_final = None if z_final is None else z_final.data_ptr()
        c.nbr_final = None if nbr_final is None else nbr_final.data_ptr()
        c.pos_final = None if pos_final is None else pos_final.data_ptr()
        return c

    def _ctl(self):
        """DroneEpisodeCtl of this env (None when neither bookkeeping nor auto-reset is on)."""
        if self._ctl_cache is None:
            h = self._home
            self._ctl_cache = self._make_ctl(h["z_final"], h["nbr_final"], h["pos_final"])
        return self._ctl_cache

    def _bind(self, views, home):
        """Point the observation / per-step output attributes at `views` (the env's own buffers or a storage slot)."""
        self.reward, self.true_reward, self.z, self.nbr_idx = views["reward"], views["true_reward"], views["z"], views["nbr_idx"]
        self.n_coll, self.done = views["n_coll"], views["done"]
        self.z_final, self.nbr_final, self.pos_final = views["z_final"], views["nbr_final"], views["pos_final"]
        self._result = views.get("_result") or StepResult(self.state, self.z, self.reward, self.n_coll, self.done,
                                                          self.true_reward)
        views["_result"] = self._result
        self._bound_home = home
        zp = views.get("_zptr")
        if zp is None:
            zp = views["_zptr"] = self.z.data_ptr()
        self._z_ptr = zp                                  # where the current observation lives (see step(into=...))

    def _rebind_home(self):
        """After step(into=(storage, t)) the observation attributes are views of a storage slot.  Everything that
        writes through them outside step() (reset, set_state, load_state, rollout) first returns to the env's own
        buffers, carrying the current observation over, so that stored experience is never clobbered.  The next
        step(into=(storage, t + 1)) copies the (re-observed) observation into the ring slot it is the ``z_pre`` of."""
        if self._bound_home:
            return
        cur = dict(z=self.z, nbr_idx=self.nbr_idx, reward=self.reward, true_reward=self.true_reward, n_coll=self.n_coll,
                   done=self.done)
        self._bind(self._home, home=True)
        for name, src in cur.items():
            self._home[name].copy_(src)

    @property
    def _use_ctl(self):
        return self.track_episodes or self.auto_reset

    # ------------------------------------------------------------------ reference API
    def create_obstacles(self, n_obstacles):
        """Cosmetic obstacles (never read by dynamics or reward) -- drone_env.py:155-169."""
        self.n_obstacles = n_obstacles
        max_size = 0.1 * np.max(self.grid)
        min_size = 0.05 * max_size
        obstacles = np.random.rand(n_obstacles, dim + 1)
        obstacles[:, 0] *= self.grid[0]
        obstacles[:, 1] *= self.grid[1]
        obstacles[:, dim] = obstacles[:, dim] * (max_size - min_size) + min_size
        return obstacles

    def reset(self, renew_obstacles=True, mask=None):
        """Re-sample initial states on the lattice, zero t, refresh z / Ni (drone_env.py:98-102, 171-212).

        ``mask`` (bool/uint8 ``[E]`` device tensor, batched mode) resets only the flagged envs."""
        torch = self._torch
        self._rebind_home()
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        p = self._params()
        # ONE launch (dronesim_reset_observe): the lattice draw of dronesim_reset[_ex] -- same stream, same nodes -- and the
        # first observation of the new state (drone_env.py:208-210), with the episode records of the envs it resets
        # retired when the env keeps them.  (Two launches through round 5: reset kernel, then observe.)
        with torch.cuda.device(self.device):
            rc = self._lib.dronesim_reset_observe(C.byref(p), C.byref(self._ctl()), None if m is None else m.data_ptr(),
                                                  self.pos.data_ptr(), self.vel.data_ptr(), self.t.data_ptr(), None,
                                                  self.z.data_ptr(), self.nbr_idx.data_ptr(), self.n_envs, self._stream())
            self._native.check(rc, "dronesim_reset_observe")
        if m is None:
            self.internal_t = 0
        if renew_obstacles:
            self.obstacles = self.create_obstacles(self.n_obstacles)
        self._sync_host_views()

    def _observe(self, mask=None, rewards=False):
        p = self._params()
        rc = self._lib.dronesim_observe(C.byref(p), self.pos.data_ptr(), self.vel.data_ptr(),
                                        self.reward.data_ptr() if rewards else None,
                                        self.true_reward.data_ptr() if rewards else None,
                                        self.z.data_ptr(), self.nbr_idx.data_ptr(),
                                        self.n_coll.data_ptr() if rewards else None,
                                        None if mask is None else mask.data_ptr(), self.n_envs, self._stream())
        self._native.check(rc, "dronesim_observe")

    def step(self, actions, copy=False, into=None):
        """One env.step() for every env (drone_env.py:214-258).

        ``into=(storage, t)`` (batched mode; a `rollout_buffer.RolloutStorage`): the launch writes every per-step
        output STRAIGHT into slot ``t`` of the storage -- reward / true_reward / n_coll / done at ``[t]``, the new
        observation at ``[t + 1]`` of its observation ring (so that ``storage.z_pre[t + 1]`` is already in place), the
        terminal observation of finishing envs at ``z_final[t]`` -- with no extra launch or copy; the env's observation
        attributes (``z``, ``nbr_idx``, ...) are re-bound to those slots.  T such calls with T distinct slots can be
        captured in one hipGraph.

        Compat mode (E == 1): ``actions`` is any indexable of N array-likes ``[2]`` (list or deque);
        returns ``(state, z_states, r_vec, n_collisions, finished, true_r_vec)`` in the reference's
        types.  Batched mode: ``actions`` is a ``[E,N,2]`` float32 tensor on this env's device;
        returns a `StepResult` of device tensors (no host sync).  NOTE: those are the env's LIVE buffers, the
        same objects on every call and overwritten by the next launch (the reference returns fresh arrays each
        step); pass ``copy=True`` -- or clone what you keep -- when storing results across steps.

        With ``auto_reset`` an env whose ``finished`` flag fires is re-sampled and re-observed inside this
        launch: rewards / n_collisions / finished describe the finished episode's last transition, state and
        z_states are the new episode's first ones."""
        torch = self._torch
        if self.batched:
            act = actions
            if not (type(act) is torch.Tensor and act.dtype is torch.float32 and act.is_cuda and act.is_contiguous()
                    and act.get_device() == self._dev_index):
                act = torch.as_tensor(np.asarray(act, np.float32) if not torch.is_tensor(act) else act,
                                      dtype=torch.float32, device=self.device).contiguous()
            if act.shape != self._act_shape:
                raise ValueError(f"actions must be [{self.n_envs},{self.n_agents},2], got {tuple(act.shape)}")
        else:
            self._push_host_state()
            a = np.asarray([np.asarray(actions[i], np.float64).reshape(2) for i in range(self.n_agents)],
                           np.float32)
            self._act.copy_(torch.from_numpy(a).view(1, self.n_agents, 2), non_blocking=False)
            act = self._act
        self._params()                                   # (collision_weight is live: refreshes self._p_addr when it changed)
        # host fast path: the buffer addresses never change, so dronesim_step_ex's arguments are marshalled once per output
        # binding (per storage slot when stepping into a RolloutStorage) into a DroneStepCall; a step is a 3-argument
        # call -- (call, actions, stream) as plain integers -- instead of 14 converted one by one
        if into is not None:
            if not self.batched:
                raise ValueError("step(into=...) needs the batched (tensor) API")
            storage, slot = into
            call, views = storage._slot(self, int(slot))
            pre = views["_pre"]                              # ring slot that serves as `z_pre[slot]` / `nbr_pre[slot]`
            if self._z_ptr != pre[2]:
                # the current observation is not where this slot's pre-step observation is read from: a reset(mask) /
                # set_state / load_state / rollout since the last step into the storage re-observed into the env's own
                # buffers, or the storage was never begun -- carry it over, so that the stored (z_pre, action) pairs
                # are the ones the policy acted on
                pre[0].copy_(self.z)
                pre[1].copy_(self.nbr_idx)
            av = views["actions"]
            if av is not None and act.data_ptr() != av.data_ptr():
                av.copy_(act)                                # (a policy writing into storage.actions[t] avoids this)
            self._bind(views, home=False)
        else:
            if not self._bound_home:
                self._bind(self._home, home=True)
            call = self._step_args
            if call is None:
                call = self._step_args = self._make_call(self._home, self._ctl() if self._use_ctl else None)
        if call[0].p != self._p_addr:
            call[0].p = self._p_addr
        if self._cur_device() == self._dev_index:
            rc = self._step_call(call[1], act.data_ptr(), self._raw_stream(self._dev_index))
        else:
            with torch.cuda.device(self.device):
                rc = self._step_call(call[1], act.data_ptr(), self._raw_stream(self._dev_index))
        if rc:
            self._native.check(rc, "dronesim_step")
        if self.batched:
            if copy:
                return StepResult(DroneState(self.pos.clone(), self.vel.clone(), self._radius), self.z.clone(),
                                  self.reward.clone(), self.n_coll.clone(), self.done.clone(), self.true_reward.clone())
            return self._result
        self._sync_host_views()
        self.internal_t += 1
        return (self.state, self.z_states, self._host_reward, self._host_n_coll, self._host_done,
                self._host_true_reward)

    def rollout_random(self, T, record_actions=False, with_pre=False):
        """T fused steps whose actions are drawn inside the kernel (RandomAgent.forward, SAC_agents.py:9-22):
        no action pool is read.  The stream is keyed by (seed, global env id, agent, t, episode), so results do
        not depend on the sharding or on how T is split into calls.  ``record_actions`` adds ``actions [T,E,N,2]``."""
        return self.rollout(None, with_pre=with_pre, _random=(int(T), bool(record_actions)))

    def rollout(self, actions, with_pre=False, _random=None):
        """T fused steps in one launch with the actions known up front (RandomAgent-style rollouts,
        SAC_agents.py:9-22 + train_problem.py:82-107).  ``actions``: ``[T,E,N,2]`` float32 device tensor.
        Returns a dict of ``[T, ...]`` tensors with every per-step output of step(); ``with_pre=True`` adds
        ``z_pre`` / ``nbr_idx_pre``, the observation each action was based on (what the reference stores as
        ``z_state`` / ``Ni`` in its experience tuples, utils.py:236-249).  With ``auto_reset`` the rollout runs
        across episode ends (see step()); with ``keep_final_obs`` the dict also holds ``z_final`` / ``nbr_final`` /
        ``pos_final`` ``[T, ...]``: at ``[s]``, the terminal observation / state of the envs whose ``done[s]`` fired."""
        torch = self._torch
        E, N, K1, c = self.n_envs, self.n_agents, self.k_closest + 1, self.c
        random_actions = _random is not None
        if random_actions:
            T = _random[0]
            act = torch.empty(T, E, N, 2, dtype=torch.float32, device=self.device) if _random[1] else None
        else:
            act = actions.to(device=self.device, dtype=torch.float32).contiguous()
            T = act.shape[0]
            if tuple(act.shape) != (T, E, N, 2):
                raise ValueError(f"actions must be [T,{E},{N},2], got {tuple(act.shape)}")
        f32 = dict(dtype=torch.float32, device=self.device)
        out = dict(reward=torch.empty(T, E, N, **f32), true_reward=torch.empty(T, E, N, **f32),
                   z=torch.empty(T, E, N, K1 * c, **f32),
                   nbr_idx=torch.empty(T, E, N, K1, dtype=torch.int32, device=self.device),
                   n_coll=torch.empty(T, E, dtype=torch.int32, device=self.device),
                   done=torch.empty(T, E, dtype=torch.uint8, device=self.device))
        if not self.batched:
            self._push_host_state()
        self._rebind_home()
        p = self._params()
        z0, nb0 = (self.z.clone(), self.nbr_idx.clone()) if with_pre else (None, None)
        # the rollout kernels address the terminal-observation buffers per STEP ([T][E][N]... like z): a ctl of this
        # call with [T, ...] buffers (returned in `out`), never the [E, ...] home buffers of step()
        if self.keep_final_obs:
            out["z_final"] = torch.zeros(T, E, N, K1 * c, **f32)
            out["nbr_final"] = torch.full((T, E, N, K1), -1, dtype=torch.int32, device=self.device)
            out["pos_final"] = torch.zeros(T, E, N, 2, **f32)
        ctl = self._make_ctl(out.get("z_final"), out.get("nbr_final"), out.get("pos_final"))
        with torch.cuda.device(self.device):
            if random_actions:
                rc = self._lib.dronesim_rollout_random(
                    C.byref(p), C.byref(ctl), self.pos.data_ptr(), self.vel.data_ptr(), self.t.data_ptr(),
                    None if act is None else act.data_ptr(), out["reward"].data_ptr(), out["true_reward"].data_ptr(),
                    out["z"].data_ptr(), out["nbr_idx"].data_ptr(), out["n_coll"].data_ptr(),
                    out["done"].data_ptr(), E, T, self._stream())
            else:
                rc = self._lib.dronesim_rollout_ex(
                    C.byref(p), C.byref(ctl) if self._use_ctl else None, self.pos.data_ptr(),
                    self.vel.data_ptr(), self.t.data_ptr(), act.data_ptr(), out["reward"].data_ptr(),
                    out["true_reward"].data_ptr(), out["z"].data_ptr(), out["nbr_idx"].data_ptr(),
                    out["n_coll"].data_ptr(), out["done"].data_ptr(), E, T, self._stream())
        self._native.check(rc, "dronesim_rollout")
        if random_actions and act is not None:
            out["actions"] = act
        if with_pre and T > 0:
            out["z_pre"] = torch.cat([z0.unsqueeze(0), out["z"][:-1]], dim=0)
            out["nbr_idx_pre"] = torch.cat([nb0.unsqueeze(0), out["nbr_idx"][:-1]], dim=0)
        if T > 0:
            self.z.copy_(out["z"][-1]); self.nbr_idx.copy_(out["nbr_idx"][-1])
            self.reward.copy_(out["reward"][-1]); self.true_reward.copy_(out["true_reward"][-1])
            self.n_coll.copy_(out["n_coll"][-1]); self.done.copy_(out["done"][-1])
            if self.keep_final_obs:                       # rows of the envs whose `finished` the LAST step raised, as after step()
                self.z_final.copy_(out["z_final"][-1]); self.nbr_final.copy_(out["nbr_final"][-1])
                self.pos_final.copy_(out["pos_final"][-1])
        if not self.batched:
            self.internal_t += T
            self._sync_host_views()
        return out

    def control(self, kind: str, u_max: float = 1.0, state=None):
        """Batched classical controllers evaluated on the CURRENT state (drone_env.py:609-679):
        ``kind`` = "proportional" or "gradient".  Returns actions ``[E,N,2]`` (device tensor) in batched
        mode, a list of N float64 row vectors in compat mode -- directly usable as ``step()`` input.
        ``state`` (compat mode, ``[N,5]``): evaluate on that state instead, leaving the env untouched."""
        torch = self._torch
        code = {"proportional": self._native.CONTROL_PROPORTIONAL, "gradient": self._native.CONTROL_GRADIENT}[kind]
        pos = self.pos
        if not self.batched:
            if state is None:
                self._push_host_state()
            else:                                    # a hypothetical state: temporary device copy, env.state untouched
                pos = torch.as_tensor(np.asarray(state, np.float64)[None, :, 0:2].astype(np.float32),
                                      device=self.device).contiguous()
        act = torch.empty_like(self._act)
        p = self._params()
        with torch.cuda.device(self.device):
            rc = self._lib.dronesim_control(C.byref(p), code, pos.data_ptr(), act.data_ptr(), float(u_max),
                                            self.n_envs, self._stream())
        self._native.check(rc, "dronesim_control")
        if self.batched:
            return act
        a = act[0].double().cpu().numpy()
        return [a[i] for i in range(self.n_agents)]

    def get_local_states(self):
        """Current localized observation.  Compat: ``(z_states, Ni)`` (the attributes
        train_problem.py:85-86 reads).  Batched: ``(z [E,N,(k+1)c], nbr_idx [E,N,k+1], nbr_cnt [E,N])``
        with ``nbr_cnt = len(Ni[i])`` (1..k+1)."""
        if not self.batched:
            return self.z_states, self.Ni
        return self.z, self.nbr_idx, (self.nbr_idx >= 0).sum(dim=2, dtype=self._torch.int32)

    # ------------------------------------------------------------------ state injection / checkpoint
    def set_state(self, pos, vel=None, t=None):
        """Inject a state (parity tests, checkpoints) and refresh the observation (rewards() path)."""
        torch = self._torch
        self._rebind_home()
        E, N = self.n_envs, self.n_agents
        self.pos.copy_(torch.as_tensor(np.asarray(pos, np.float32) if not torch.is_tensor(pos) else pos,
                                       dtype=torch.float32).reshape(E, N, 2))
        if vel is None:
            self.vel.zero_()
        else:
            self.vel.copy_(torch.as_tensor(np.asarray(vel, np.float32) if not torch.is_tensor(vel) else vel,
                                           dtype=torch.float32).reshape(E, N, 2))
        if t is not None:
            tt = torch.as_tensor(np.asarray(t, np.int32) if not torch.is_tensor(t) else t, dtype=torch.int32)
            self.t.copy_(tt.reshape(-1).expand(E) if tt.numel() == 1 else tt.reshape(E))
            if not self.batched:
                self.internal_t = int(self.t[0].item())
        with torch.cuda.device(self.device):
            self._observe(rewards=True)
        self._sync_host_views()

    def get_state(self):
        """``dict(pos, vel, t, episode, seed[, episode_acc])`` clones -- everything needed to resume the env."""
        st = dict(pos=self.pos.clone(), vel=self.vel.clone(), t=self.t.clone(),
                  episode=self.episode.clone(), seed=self.seed)
        if self.episode_acc is not None:
            st["episode_acc"] = self.episode_acc.clone()
        return st

    def load_state(self, state):
        """Restore a `get_state()` checkpoint: positions, velocities, step counters AND the random-stream
        position (``seed`` + per-env ``episode`` counters), so the resumed env draws the same initial states
        (and in-kernel actions) as the original would have; then refresh the observation."""
        torch = self._torch
        self.seed = int(state["seed"])
        self._ctl_cache = None
        self._step_args = None
        self._ctl_generation = getattr(self, "_ctl_generation", 0) + 1    # storages re-build their per-slot ctls
        self.episode.copy_(torch.as_tensor(state["episode"], dtype=torch.int32).reshape(self.n_envs))
        if self.episode_acc is not None and "episode_acc" in state:
            self.episode_acc.copy_(torch.as_tensor(state["episode_acc"], dtype=torch.float64).reshape(self.n_envs, 8))
        self.set_state(state["pos"], state["vel"], state["t"])

    # ------------------------------------------------------------------ episode bookkeeping (train_problem.py:98-121)
    def _acc_i32(self):
        return self.episode_acc.view(self._torch.int32)               # [E,16] int32 view of the records

    def episode_stats(self):
        """Per-env view of the device records (no host sync): the episode in progress -- ``ep_return`` /
        ``ep_true_return`` (sum over steps of the MEAN over agents, as train_problem.py:98-99 accumulates),
        ``ep_collisions``, ``ep_len`` -- and totals over the episodes each env has completed."""
        if self.episode_acc is None:
            raise RuntimeError("construct the env with track_episodes=True (or auto_reset=True)")
        a, ai, N = self.episode_acc, self._acc_i32(), self.n_agents
        al = a.view(self._torch.int64)
        return dict(ep_return=a[:, 0] / N, ep_true_return=a[:, 1] / N, ep_collisions=ai[:, 4], ep_len=ai[:, 5],
                    episodes=ai[:, 6], done_return=a[:, 4] / N, done_true_return=a[:, 5] / N,
                    done_collisions=al[:, 6], done_len=al[:, 7])

    def episode_totals(self, out=None):
        """One launch (`dronesim_episode_reduce`, fixed summation order): float64 ``[8]`` device tensor of this
        rank's sums (done_return, done_true_return, done_collisions, done_len, episodes, ep_return, ep_true_return,
        ep_len) with the returns summed over agents (divide by N for the reference's per-step mean).  This vector is
        what multi-GPU runs all-gather (`sharding.reduce_episode_records`).  ``out``: a float64 ``[8]`` device tensor
        to reduce into (e.g. one slot of a ring when several reductions are captured in one hipGraph)."""
        if self.episode_acc is None:
            raise RuntimeError("construct the env with track_episodes=True (or auto_reset=True)")
        dst = self._episode_totals if out is None else out
        if out is not None and not (out.dtype == self._torch.float64 and out.numel() == 8 and out.is_contiguous()
                                    and out.device.type == "cuda" and out.device.index == self.device.index):
            raise ValueError("out must be a contiguous float64 [8] tensor on the env's device")
        with self._torch.cuda.device(self.device):
            rc = self._lib.dronesim_episode_reduce(self.episode_acc.data_ptr(), self.n_envs,
                                                   dst.data_ptr(), self._stream())
        self._native.check(rc, "dronesim_episode_reduce")
        return dst

    # ------------------------------------------------------------------ compat-mode host views
    def _sync_host_views(self):
        """E == 1 compat: mirror device state/observation into the reference's Python types.
        Everything the reference returns per step travels in ONE device-to-host copy."""
        if self.batched:
            return
        torch = self._torch
        N, K1, c = self.n_agents, self.k_closest + 1, self.c
        flat = torch.cat([self.pos.view(-1), self.vel.view(-1), self.z.view(-1), self.reward.view(-1),
                          self.true_reward.view(-1), self.nbr_idx.view(-1).float(), self.n_coll.float(),
                          self.done.float()]).double().cpu().numpy()
        o = 0

        def take(n):
            nonlocal o
            out = flat[o:o + n]; o += n
            return out
        st = np.empty((N, 5))
        st[:, 0:2] = take(2 * N).reshape(N, 2)
        st[:, 2:4] = take(2 * N).reshape(N, 2)
        st[:, 4] = self.drone_radius
        if getattr(self, "state", None) is None or self.state.shape != st.shape:
            self.state = st
        else:
            self.state[...] = st                      # keep the same live array object (drone_env.py:258)
        self._state_pushed = self.state.copy()
        z = take(N * K1 * c).reshape(N, K1, c)
        self._host_reward = take(N).copy()
        self._host_true_reward = take(N).copy()
        nb = take(N * K1).astype(np.int64).reshape(N, K1)
        self._host_n_coll = np.int64(take(1)[0])
        self._host_done = bool(take(1)[0])
        self.z_states = [z[i].copy() for i in range(N)]
        self.Ni = [[int(i)] + [j for j in nb[i, 1:] if j >= 0] for i in range(N)]

    def _push_host_state(self):
        """Callers may write into env.state / env.internal_t between steps (the reference's state is a
        plain attribute); upload it if it changed."""
        torch = self._torch
        if self.batched:
            return
        if not np.array_equal(self.state[:, :4], self._state_pushed[:, :4]):
            self.pos.copy_(torch.from_numpy(self.state[None, :, 0:2].astype(np.float32)))
            self.vel.copy_(torch.from_numpy(self.state[None, :, 2:4].astype(np.float32)))
        self.t.fill_(int(self.internal_t))

    def __str__(self):
        """Same printout as the reference (drone_env.py:105-113); returns ""."""
        print("Grid size: [x_lim, y_lim]\n", self.grid)
        if self.batched:
            print(f"State: {self.n_envs} envs x {self.n_agents} agents on {self.device} (pos/vel tensors)")
        else:
            print("State: [x, y, vx, vy, r]\n", self.state)
        print(f"z_sattes for k_closest = {self.k_closest}: simplify? {self.simplify_zstate}")
        print("safety distance for each agent:\n", self.d_safety)
        print("Deltas disk radius for each agent: \n", self.deltas)
        print(f"Collision cost weight (per unit of time) = {self.collision_weight} ")
        return ""

    # ------------------------------------------------------------------ figures (compat.py; CPU, matplotlib)
    def show(self, state=None, not_animate=True):
        """Current (or given) ``[N,5]`` state on the grid (drone_env.py:404-434); returns the figure."""
        from . import compat
        return compat.show_state(self, None if not_animate else state, show=not_animate)

    def plot(self, trajectory, episode=None):
        """Paths + collision marks of one episode from a list of ``[N,5]`` states (drone_env.py:450-514)."""
        from . import compat
        return compat.plot_trajectory(self, trajectory, episode, show=True)[0]

    def animate(self, trajectory, z_trajectory, deltas, episode, name="test", format="gif"):
        """``videos/<name>.gif|mp4`` of one episode (drone_env.py:516-607)."""
        from . import compat
        print("\nSaving animation...")
        full = compat.animate_trajectory(self, trajectory, z_trajectory, deltas, episode, name, format)
        print(f"Animation saved as {full}")
        return full


def gradient_control(state, env, u_max=1):
    """Drop-in for the reference's module-level `gradient_control(state, env, u_max)` (drone_env.py:609-650).
    Pure like the reference's: a `state` other than the env's own is evaluated without touching the env."""
    return env.control("gradient", u_max, state=None if (env.batched or state is env.state) else state)


def proportional_control(state, env):
    """Drop-in for the reference's `proportional_control(state, env)` (drone_env.py:652-679); pure."""
    return env.control("proportional", 1.0, state=None if (env.batched or state is env.state) else state)
