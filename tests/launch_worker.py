"""Rank body of tests/test_gpu_launcher.py (test infrastructure): one process of a torch.distributed.run launch.

Every rank builds ITS shard of the same global batch on the one GPU the box has (backend gloo: RCCL cannot put two
ranks on one device), rolls it forward with the in-kernel RandomAgent stream and auto-reset, and writes its GPU
outputs plus the result of the path's only exchange (`reduce_episode_records`) to OUT_DIR/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def run(rank, world, out_dir, N, G, E, T):
    from scalable_collision_avoidance_rl_amd import drones
    from scalable_collision_avoidance_rl_amd.sharding import reduce_episode_records
    torch.cuda.set_device(0)
    env = drones(N, 0, [G, G], "O", deltas=np.ones(N), simplify_zstate=True, n_envs=E, batched=True, device="cuda:0",
                 seed=1234, rank=rank, world_size=world, auto_reset=True)
    env.t.fill_(150)                                  # the 200-step limit fires inside the rollout
    out = env.rollout_random(T)
    stepwise = env.step(torch.zeros(env.n_envs, N, 2, device="cuda:0"), copy=True)
    summary = reduce_episode_records(env)             # fixed-order local reduction + one all-gather
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=env.env_lo, hi=env.env_hi,
             pos=env.pos.cpu().numpy(), z=env.z.cpu().numpy(), nbr=env.nbr_idx.cpu().numpy(),
             reward=out["reward"].cpu().numpy(), done=out["done"].cpu().numpy(), acc=env.episode_acc.cpu().numpy(),
             last_reward=stepwise.rewards.cpu().numpy(),
             summary_keys=np.array(sorted(summary)), summary_vals=np.array([summary[k] for k in sorted(summary)]))


if __name__ == "__main__":
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = sys.argv[1]
    N, G, E, T = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    if world > 1:
        dist.init_process_group(backend="gloo")
    try:
        run(rank, world, out_dir, N, G, E, T)
    finally:
        if world > 1:
            dist.destroy_process_group()
