#!/usr/bin/env python3
"""The reference's rollout loop (train_problem.py:66-132) on the MI355X-native stack.

Part 1 -- drop-in, E = 1: the loop body of train_problem.py:82-107 verbatim against the new `drones`
          (reference Python types in and out), driven by the classical P-controller instead of a learner.
Part 2 -- the same loop batched: E envs x N agents stay on the device; a per-agent softmax-16 policy
          (DiscreteSoftmaxNN shapes, random init) is evaluated for all agents in one launch, the env
          steps in one launch and writes each transition straight into an on-device `RolloutStorage`
          (the batched `ExperienceBuffers`, utils.py:232-253), and the episode's Monte-Carlo returns /
          neighbour-summed advantage weights (SAC_agents.py:304-307, 333-351) are reduced on the device.

    python examples/rollout_loop.py [--envs 4096] [--agents 64]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import scalable_collision_avoidance_rl_amd.drone_env as drone_env          # was: import drone_env
from scalable_collision_avoidance_rl_amd.policies import BatchedMLP
from scalable_collision_avoidance_rl_amd.rollout_buffer import RolloutStorage, mc_returns, neighbour_advantage


def part1():
    n_agents = 5
    deltas = np.ones(n_agents) * 1.0
    env = drone_env.drones(n_agents=n_agents, n_obstacles=0, grid=[5, 5], end_formation="O", deltas=deltas,
                           simplify_zstate=True)
    env.collision_weight = 0.2
    total_episode_reward = total_true_episode_reward = total_episode_collisions = 0
    t_iter, finished = 0, False
    while not finished:                                            # train_problem.py:82
        state, z_states, Ni = env.state, env.z_states, env.Ni      # :84-86
        actions = drone_env.proportional_control(state, env)       # :90 (commented alternative in the reference)
        new_state, new_z, rewards, n_collisions, finished, true_rewards = env.step(actions)   # :94
        total_episode_reward += np.mean(rewards)                   # :98-100
        total_true_episode_reward += np.mean(true_rewards)
        total_episode_collisions += n_collisions
        t_iter += 1
    env.reset(renew_obstacles=False)                               # :132
    print(f"[E=1 drop-in] episode of {t_iter} steps: return {total_episode_reward:.2f}, "
          f"true return {total_true_episode_reward:.2f}, collisions {total_episode_collisions}")


def part2(E, N):
    """The batched loop twice: (a) the caller keeps the experience itself -- one torch copy per stored tensor and
    step, as in round 2; (b) `RolloutStorage`: the step launch, the policy and the critic write straight into slot t
    (`env.step(act, into=(storage, t))`, `act_out=`, `out=`), no copy launches; (c) = (b) captured as one hipGraph."""
    G = 28.0 if N == 64 else max(6.0, 0.45 * N)
    T = drone_env.max_time_steps
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * 0.2
    d = 6
    wa = [r(N, d, 300), r(N, 300), r(N, 300, 300), r(N, 300), r(N, 300, 16), r(N, 16)]
    wc = [r(N, d, 200), r(N, 200), r(N, 200, 200), r(N, 200), r(N, 200, 1), r(N, 1)]

    def build():
        env = drone_env.drones(N, 0, [G, G], "O", deltas=np.ones(N), simplify_zstate=True, n_envs=E, seed=0, auto_reset=True)
        return (env, BatchedMLP(*wa, 1, 1, device=env.device, seed=1, precision="bf16"),
                BatchedMLP(*wc, 0, 0, device=env.device, precision="bf16"))

    # (a) caller-side storage: clone / copy_ per tensor and step
    env, actor, critic = build()
    rew = torch.empty(T, E, N, device=env.device); val = torch.empty(T, E, N, device=env.device)
    nbr = torch.empty(T, E, N, env.k_closest + 1, dtype=torch.int32, device=env.device)
    zpre = torch.empty(T, E, N, 6, device=env.device); acts = torch.empty(T, E, N, 2, device=env.device)
    done = torch.empty(T, E, dtype=torch.uint8, device=env.device)

    def loop_a():
        for t in range(T):
            z, nbr_idx, _ = env.get_local_states()                     # the observation the action is based on
            zpre[t].copy_(z); nbr[t].copy_(nbr_idx)
            val[t].copy_(critic.forward(z).squeeze(-1))
            act, _ = actor.sample_action(z, env=env)                   # all N policies, one launch
            acts[t].copy_(act)
            res = env.step(act)                                        # all E envs, one launch
            rew[t].copy_(res.rewards); done[t].copy_(res.finished)
    loop_a(); torch.cuda.synchronize(); t0 = time.perf_counter(); loop_a(); torch.cuda.synchronize()
    dt_a = time.perf_counter() - t0
    w_a = neighbour_advantage(mc_returns(rew, 0.99, done), val, nbr, 0.99, done)

    # (b) on-device experience storage filled by the launches themselves
    env, actor, critic = build()
    st = RolloutStorage(env, T, actions=True, values=True)

    def loop_b():
        st.begin()
        for t in range(T):
            critic.forward(env.z, out=st.values[t])
            actor.sample_action(env.z, env=env, act_out=st.actions[t])
            env.step(st.actions[t], into=(st, t))
    loop_b(); torch.cuda.synchronize(); t0 = time.perf_counter(); loop_b(); torch.cuda.synchronize()
    dt_b = time.perf_counter() - t0
    G_t = st.returns(0.99)                                             # SAC_agents.py:304-307
    w = st.advantage(st.values, 0.99, G_t)                             # SAC_agents.py:333-351
    same = torch.equal(w, w_a) and torch.equal(st.z_pre, zpre) and torch.equal(st.actions, acts)

    # (c) the same window as one hipGraph (T distinct slot address sets in one capture)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loop_b()
    graph.replay(); torch.cuda.synchronize(); t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize()
    dt_c = time.perf_counter() - t0
    nz, _ = st.next_z()
    print(f"[batched] {E} envs x {N} agents x {T} steps (bf16 policy + critic in the loop):\n"
          f"   (a) caller-side copies   {dt_a*1e3:7.1f} ms = {E*N*T/dt_a:.3e} agent-steps/s\n"
          f"   (b) RolloutStorage       {dt_b*1e3:7.1f} ms = {E*N*T/dt_b:.3e} agent-steps/s   (same numbers as (a): {same})\n"
          f"   (c) (b) as one hipGraph  {dt_c*1e3:7.1f} ms = {E*N*T/dt_c:.3e} agent-steps/s\n"
          f"   mean return {float(G_t[0].mean()):.1f}, episodes ended in the window {int(st.done.sum())}, "
          f"advantage weight rms {float(w.pow(2).mean().sqrt()):.3f}, terminal observations kept: "
          f"{bool((nz != st.z).any()) if int(st.done.sum()) else 'n/a'}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=64)
    a = ap.parse_args()
    part1()
    part2(a.envs, a.agents)
