"""Multi-GPU layout of the hot path: shard the env axis, exchange only the logged statistic.

Environments never interact (one `drones` object = one env in the reference), so rank g of W
owns envs [lo, hi) of the global env axis (`shard_range`) and steps them with no data-path
collective.  The only exchange is the global reward / collision statistic the rollout loop
logs per episode (train_problem.py:98-100, 118-120): each rank contributes one small float64
vector; one all-gather (RCCL over xGMI when the backend is "nccl", gloo in the CPU tests)
per episode -- or per K steps -- makes the global figures available on every rank.  The
message is tens of bytes: latency-bound, kept off the per-step path."""
from __future__ import annotations

import ctypes as C

from .drone_env import shard_range  # noqa: F401  (re-exported)

STAT_FIELDS = ("sum_reward", "sum_true_reward", "sum_collisions", "agent_steps", "env_steps")


class EpisodeStats:
    """Per-rank accumulators of the logged statistic, kept on the env's device (no host sync)."""

    def __init__(self, device):
        import torch
        self.vec = torch.zeros(len(STAT_FIELDS), dtype=torch.float64, device=device)
        self._scratch = None                            # device scratch of dronesim_episode_stats (CUDA tensors)

    def add_step(self, rewards, true_rewards, n_collisions):
        """rewards/true_rewards [E,N], n_collisions [E] of one step (what train_problem.py:98-100 sums).
        Device tensors: one launch of `dronesim_episode_stats` (fixed summation order, no host sync).
        Host tensors (the CPU tests of the exchange): plain torch sums."""
        import torch
        E, N = rewards.shape
        if rewards.is_cuda:
            from . import _native
            if self._scratch is None:
                self._scratch = torch.zeros(_native.STATS_SCRATCH_DOUBLES, dtype=torch.float64, device=self.vec.device)
            r, tr = rewards.contiguous(), true_rewards.contiguous()
            nc = n_collisions.to(torch.int32).contiguous()
            with torch.cuda.device(self.vec.device):
                rc = _native.lib().dronesim_episode_stats(r.data_ptr(), tr.data_ptr(), nc.data_ptr(), E, N,
                                                          self.vec.data_ptr(), self._scratch.data_ptr(),
                                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _native.check(rc, "dronesim_episode_stats")
            return
        self.vec += torch.stack([rewards.sum(dtype=self.vec.dtype), true_rewards.sum(dtype=self.vec.dtype),
                                 n_collisions.sum(dtype=self.vec.dtype),
                                 torch.tensor(float(E * N), dtype=self.vec.dtype), torch.tensor(float(E), dtype=self.vec.dtype)])

    def reduce(self, group=None):
        """All-gather every rank's vector and sum locally -> dict of global figures (same on all ranks)."""
        return summarize(all_gather_stats(self.vec, group))


EPISODE_FIELDS = ("done_return", "done_true_return", "done_collisions", "done_len", "episodes",
                  "ep_return", "ep_true_return", "ep_len")


def reduce_episode_records(env, group=None):
    """The path's only exchange, on the per-env episode records the step kernel keeps (include/dronesim.h:
    DroneEpisodeAcc): one launch sums this rank's records in a fixed order (`dronesim_episode_reduce`), ONE
    all-gather of the resulting 8-double vector (RCCL over xGMI under backend "nccl"; 64 B per rank) makes
    every rank's sums available everywhere, and the global figures the reference logs per episode
    (train_problem.py:118-121, 136-140) follow locally."""
    return reduce_episode_totals(env.episode_totals(), env.n_agents, group)


def reduce_episode_totals(totals, n_agents, group=None):
    """`reduce_episode_records` on an already reduced per-rank 8-vector (`drones.episode_totals()`; a host tensor in
    the gloo tests of the exchange)."""
    return summarize_episodes(all_gather_stats(totals, group), n_agents)


def summarize_episodes(gathered, n_agents):
    """Global per-episode means from the gathered [world, 8] episode totals (returns are summed over agents on
    the device; the reference accumulates the mean over agents per step: divide by N)."""
    tot = gathered.double().sum(0)
    eps = max(float(tot[4]), 1.0)
    return {"episodes": float(tot[4]),
            "mean_episode_reward": float(tot[0]) / n_agents / eps,
            "mean_episode_true_reward": float(tot[1]) / n_agents / eps,
            "mean_episode_collisions": float(tot[2]) / eps,
            "mean_episode_len": float(tot[3]) / eps,
            "env_steps": float(tot[3]) + float(tot[7]),
            "agent_steps": (float(tot[3]) + float(tot[7])) * n_agents,
            "mean_reward": (float(tot[0]) + float(tot[5])) / n_agents / max(float(tot[3]) + float(tot[7]), 1.0),
            "mean_true_reward": (float(tot[1]) + float(tot[6])) / n_agents / max(float(tot[3]) + float(tot[7]), 1.0),
            # collisions are only totalled for COMPLETED episodes (done_collisions / done_len): named accordingly
            "collisions_per_completed_env_step": float(tot[2]) / max(float(tot[3]), 1.0),
            "world_size": int(gathered.shape[0])}


def all_gather_stats(vec, group=None, async_op=False, force_collective=False):
    """[world, len(vec)] tensor holding every rank's statistic vector (identity when not distributed).
    ``async_op=True`` returns ``(out, work)`` without making the current stream wait for the collective (RCCL runs it
    on its own stream behind the producer of `vec`): the rollout's next launches are not held up; call
    ``work.wait()`` (or synchronise) before reading ``out``.  ``force_collective`` runs the collective even in a
    world of one (smoke test of the RCCL path on a single GPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        out = vec.view(1, -1).clone()
        return (out, None) if async_op else out
    world = dist.get_world_size(group)
    src = vec.view(1, -1).contiguous()
    if dist.get_backend(group) == "gloo" and src.is_cuda:      # gloo gathers host tensors (CPU tests, debugging)
        src = src.cpu()
    out = torch.empty(world, vec.numel(), dtype=vec.dtype, device=src.device)
    if async_op and src.is_cuda:
        return out, dist.all_gather_into_tensor(out, src.clone(), group=group, async_op=True)
    dist.all_gather_into_tensor(out, src, group=group)
    out = out.to(vec.device)
    return (out, None) if async_op else out


def summarize(gathered):
    """Global means from the gathered [world, 5] statistic."""
    tot = gathered.sum(0)
    agent_steps = max(float(tot[3]), 1.0)
    env_steps = max(float(tot[4]), 1.0)
    return {"mean_reward": float(tot[0]) / agent_steps, "mean_true_reward": float(tot[1]) / agent_steps,
            "collisions_per_env_step": float(tot[2]) / env_steps, "agent_steps": float(tot[3]),
            "env_steps": float(tot[4]), "world_size": int(gathered.shape[0])}
