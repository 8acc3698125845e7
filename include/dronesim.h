/*
 * dronesim.h -- C ABI of the MI355X-native batched drone_env hot path.
 *
 * The reference (AndreuMatoses/scalable-collision-avoidance-RL) is pure Python
 * and has no FFI layer; its boundary for this path is the duck-typed surface of
 * class `drones` (drone_env.py:53-401).  Each entry point below replaces one
 * reference method, batched over E independent environments, and is what a
 * Python `ctypes` binding of that class loads (see INTEGRATION.md and
 * scalable_collision_avoidance_rl_amd/_native.py).
 *
 * Conventions
 *  - Every buffer is CALLER-OWNED DEVICE memory (hipMalloc / a torch tensor's
 *    data_ptr()), contiguous, env-major:
 *        pos, vel, act      float32 [E][N][2]
 *        reward, true_reward float32 [E][N]
 *        z                  float32 [E][N][k+1][c]   (c = 2 or 5)
 *        nbr_idx            int32   [E][N][k+1]      slot 0 = i, then the real
 *                                                    neighbours by ascending d_ij,
 *                                                    unused slots = -1
 *        n_coll             int32   [E]   ordered colliding pairs (always even)
 *        done               uint8   [E]
 *        t                  int32   [E]   internal_t of every env
 *  - The library allocates nothing persistent, never synchronises the host and
 *    enqueues all work on `stream` (a hipStream_t passed as void*; NULL = the
 *    null stream).  Pass the stream the neighbouring kernels run on.
 *  - Return value: 0 on success, a negative DRONESIM_E* code otherwise; nothing
 *    is thrown across the ABI.  dronesim_last_error() gives a thread-local
 *    description of the last failure.
 *  - Re-entrant; the only mutable state is that thread-local string and a per-device record of which kernels
 *    have been opted into > 48 KiB of dynamic LDS (hipFuncSetAttribute; guarded by a mutex).
 *  - One process drives one GPU; multi-GPU runs shard the E axis across
 *    processes (env_base keeps random streams independent of the sharding).
 */
#ifndef DRONESIM_H
#define DRONESIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRONESIM_VERSION 600           /* 0.6.0: dronesim_reset_observe (env.reset() as one launch); 0.5.0: the float64 verification entry
                                          points moved to libdronesim_verify.so (dronesim_verify.h), DroneMlpBf16.wscale */
#define DRONESIM_MAX_K 8               /* k_closest supported by the kernels */
#define DRONESIM_MAX_AGENTS 1024       /* one workgroup holds one env */

#define DRONESIM_OK 0
#define DRONESIM_EINVAL (-1)           /* null pointer / size out of range */
#define DRONESIM_EUNSUPPORTED (-2)     /* k or N beyond the compiled kernels */
#define DRONESIM_ELAUNCH (-3)          /* HIP launch error (see last_error) */

/* Constants of one `drones` object (shared by all E envs).
 * Scalars: drone_env.py:27-30 (dim, dt, max_time_steps), :72 (collision_weight),
 * :269-270 (q = 2*dt, b = collision_weight*dt), :251 (done radius 0.2), :386 (1.1).
 * Arrays (device pointers): goal ring xF and safety distance d_hat from
 * generate_formation (:115-153), deltas after the clip of :85-89, radii (:75). */
typedef struct DroneParams {
    int32_t N;              /* n_agents, 2..DRONESIM_MAX_AGENTS                     */
    int32_t k;              /* k_closest, 1..min(N-1, DRONESIM_MAX_K)               */
    int32_t c;              /* columns of a z row: 2 (simplify_zstate) or 5         */
    int32_t max_steps;      /* max_time_steps (200)                                 */
    float dt;               /* 0.05                                                 */
    float q;                /* formation weight, 2*dt                               */
    float b;                /* collision weight, collision_weight*dt                */
    float done_radius;      /* 0.2                                                  */
    float ghost_factor;     /* 1.1                                                  */
    /* host-known bounds of the arrays below (the library never reads device
       memory on the host): used to pick the kernel variant and the early-out
       radius; when min == max for all three, the kernels take the constants from
       here instead of reading the arrays.  0 < d_hat_min <= d_hat_max required.    */
    float d_hat_min;
    float d_hat_max;
    float delta_min;
    float delta_max;
    float radius_min;
    float radius_max;
    const float *xF;        /* [N][2] */
    const float *d_hat;     /* [N]    */
    const float *delta;     /* [N]    */
    const float *radius;    /* [N]    */
    /* optional (NULL = zeros): the part of the float64 goal ring that float32 drops, xF_lo = (float)(xF64 - (double)xF),
       [N][2].  The kernels form x - xF as (x - xF) - xF_lo: the first difference is exact near the goal, so the offset,
       the arrival test (:251) and the ghost direction (:383-386) keep float32 RELATIVE accuracy there instead of an
       absolute ulp32(G) -- without it a ghost row of an agent 0.04 from its goal is 2e-5 off the reference's at G = 31. */
    const float *xF_lo;
} DroneParams;

/* drones.step(actions)                                   drone_env.py:214-258
 * Integrates pos += dt*act, vel = act IN PLACE, then evaluates rewards(),
 * distance_data() and localized_states() on the new state (:260-401), the
 * termination test (:247-254) and t += 1 (:256).                                  */
int dronesim_step(const DroneParams *p, float *pos, float *vel, int32_t *t, const float *act,
                  float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                  int32_t *n_coll, uint8_t *done, int E, void *stream);

/* drones.rewards(state, ...) without integration         drone_env.py:260-293
 * (what init_agents runs to produce the first z_states / Ni, :208-210).
 * reward, true_reward, n_coll may be NULL (not written).  mask NULL = every env,
 * otherwise only envs with mask[e] != 0 are evaluated and written.                */
int dronesim_observe(const DroneParams *p, const float *pos, const float *vel,
                     float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                     int32_t *n_coll, const uint8_t *mask, int E, void *stream);

/* drones.reset() / init_agents(): state part             drone_env.py:98-102, 171-205
 * Draws N distinct nodes of the div_x x div_y lattice (node (a,b) -> (a*pitch,
 * b*pitch), pitch = 2*1.1*l) per env with a counter-based Philox4x32-10 stream
 * keyed by (seed, env_base + e, episode[e], agent, round); zeroes vel and t and
 * increments episode[e] (int32 [E], device, in/out: how many times env e has been
 * reset -- kept on the device so that a captured hipGraph replays fresh streams).
 * mask as above.  node_out (int32 [E][N], node = a*div_y + b) may be NULL.
 * Follow with dronesim_observe() to refresh z / nbr_idx (:208-210).               */
int dronesim_reset(const DroneParams *p, int div_x, int div_y, float pitch,
                   uint64_t seed, int64_t env_base, const uint8_t *mask,
                   float *pos, float *vel, int32_t *t, int32_t *episode, int32_t *node_out,
                   int E, void *stream);

/* T consecutive drones.step() calls in ONE launch (the `while not finished` loop
 * of train_problem.py:82-107 with the actions known up front, e.g. RandomAgent,
 * SAC_agents.py:9-22).  act is [T][E][N][2]; every per-step output of
 * dronesim_step is written for every step into [T][...] buffers laid out as T
 * consecutive copies of the per-step layout; pos/vel/t hold the final state.
 * Envs are NOT reset inside the rollout (t keeps counting, done stays set).
 * For N = 64 the far filter's verdicts (taken with radius reach + skin, skin = 0.4 reach) are kept in
 * registers and reused until some agent of the env has moved more than skin/2: bit-identical results. */
int dronesim_rollout(const DroneParams *p, float *pos, float *vel, int32_t *t, const float *act,
                     float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                     int32_t *n_coll, uint8_t *done, int E, int T, void *stream);

/* ---- episode bookkeeping on the device (round 2) -------------------------------------------------------
 * The rollout loop logs, per episode, the sums it accumulates on EVERY step (train_problem.py:72-74, 98-100,
 * 118-121):  total_episode_reward += mean(rewards), total_true_episode_reward += mean(true_rewards),
 * total_episode_collisions += n_collisions, t_iter += 1.  DroneEpisodeAcc is that record, one per env, kept in
 * device memory and updated by the step kernel itself (the per-env reward sums are one fixed-order wave reduction
 * in the kernel's epilogue: bit-reproducible, no extra launch, 48 B of traffic per env-step).
 *   ep_*    the episode in progress.  ep_return / ep_true_return hold sum_t sum_i r_i: DIVIDE BY N for the
 *           reference's figure (it adds the mean over agents each step).
 *   done_*  totals over the episodes this env has completed; `episodes` counts them.  An episode is retired
 *           (ep_* added into done_*, ep_* cleared) when the env is reset: by the step kernel itself under
 *           auto_reset, or by dronesim_reset_ex.
 * The caller zero-initialises the records once.  dronesim_episode_reduce sums them over the envs of a rank (fixed
 * order); multi-GPU runs all-gather that 8-double vector -- the path's only exchange.                         */
typedef struct DroneEpisodeAcc {
    double ep_return;           /* sum over steps of sum_i reward_i            train_problem.py:98  */
    double ep_true_return;      /* same for true_reward                         :99                  */
    int32_t ep_collisions;      /* sum of n_collisions                          :100                 */
    int32_t ep_len;             /* steps of the episode in progress (t_iter)    :110                 */
    int32_t episodes;           /* completed episodes                                                */
    int32_t reserved;
    double done_return;         /* totals over completed episodes               :118-121             */
    double done_true_return;
    int64_t done_collisions;
    int64_t done_len;
} DroneEpisodeAcc;              /* 64 bytes, one cache line per env */

/* Episode control of the *_ex entry points.  acc may be NULL (no bookkeeping).  auto_reset != 0: an env whose
 * `done` fires (drone_env.py:251) is re-sampled (exactly as dronesim_reset would: same Philox stream, so the
 * fresh states do not depend on which of the two did it), its record retired, its t zeroed and its observation
 * recomputed INSIDE the same launch -- what train_problem.py:132 does after the `while not finished` loop.
 * reward / true_reward / n_coll / done of that step still describe the finished episode's last transition;
 * pos / vel / t / z / nbr_idx hold the new episode's first state and observation.
 * The lattice (div_x, div_y, pitch), seed, env_base and episode[] have the meaning they have in
 * dronesim_reset and are needed when auto_reset or in-kernel random actions are used.                        */
typedef struct DroneEpisodeCtl {
    DroneEpisodeAcc *acc;       /* [E] device records, or NULL                                       */
    int32_t auto_reset;
    int32_t div_x;              /* lattice of dronesim_reset: nodes per axis ...              */
    int32_t div_y;
    float pitch;                /* ... and node spacing                                          */
    uint64_t seed;
    int64_t env_base;
    int32_t *episode;           /* [E] device, resets seen per env (in/out)                          */
    /* Terminal observation under auto_reset (all three optional, NULL = not wanted).  The reference's step() returns
     * the z-states and state of the FINAL state of an episode (drone_env.py:258) and its loop stores them as `new_z` of
     * the last transition (utils.py:244-249) before it resets (train_problem.py:132).  With auto_reset the launch that
     * ends an episode overwrites z / nbr_idx / pos with the NEW episode's first observation; when these pointers are
     * given, the finished env's terminal rows are kept here instead of being lost: written ONLY for envs whose `done`
     * fired in this launch (rows of other envs are left untouched), same layouts as z / nbr_idx / pos.
     * SIZE: dronesim_step_ex writes rows [e][i]; the fused rollouts (dronesim_rollout_ex / _rollout_random) write
     * the rows of an env that finishes at step s at [s][e][i], exactly like their z output: there all three buffers
     * must hold T x E x N rows.  Passing [E]-sized buffers to a rollout of T > 1 steps is a caller error the library
     * cannot detect (plain pointers).                                                                           */
    float *z_final;             /* step: [E][N][k+1][c];  rollout: [T][E][N][k+1][c]                 */
    int32_t *nbr_final;         /* step: [E][N][k+1];     rollout: [T][E][N][k+1]                    */
    float *pos_final;           /* step: [E][N][2];       rollout: [T][E][N][2]                      */
} DroneEpisodeCtl;

/* dronesim_step / dronesim_rollout with episode bookkeeping and optional in-kernel auto-reset.  ctl == NULL
 * behaves exactly like the plain entry points.                                                               */
int dronesim_step_ex(const DroneParams *p, const DroneEpisodeCtl *ctl, float *pos, float *vel, int32_t *t,
                     const float *act, float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                     int32_t *n_coll, uint8_t *done, int E, void *stream);
int dronesim_rollout_ex(const DroneParams *p, const DroneEpisodeCtl *ctl, float *pos, float *vel, int32_t *t,
                        const float *act, float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                        int32_t *n_coll, uint8_t *done, int E, int T, void *stream);

/* dronesim_step_ex with its arguments marshalled ONCE: a rollout loop calls step() with the same buffers every time
 * (only the actions and, possibly, the stream change), and through an FFI that converts arguments one by one
 * (ctypes, cgo, JNI) the 14-argument form costs more host time than the launch.  The host class keeps one
 * DroneStepCall per (env, output binding) -- per storage slot when stepping into a RolloutStorage -- and makes a
 * 3-argument call per step (drone_env.py:214, train_problem.py:94).  Same checks, same launch, same result.   */
typedef struct DroneStepCall {
    const DroneParams *p;
    const DroneEpisodeCtl *ctl;         /* NULL = plain dronesim_step                                         */
    float *pos;
    float *vel;
    int32_t *t;
    float *reward;
    float *true_reward;
    float *z;
    int32_t *nbr_idx;
    int32_t *n_coll;
    uint8_t *done;
    int32_t E;
    int32_t reserved;
} DroneStepCall;
int dronesim_step_call(const DroneStepCall *call, const float *act, void *stream);

/* T fused steps whose actions are drawn INSIDE the kernel: RandomAgent.forward, SAC_agents.py:9-22
 * (clip(-1 + 2 rand(2), -1, 1)) for every agent and step, from the counter-based stream
 *   philox4x32-10(ctr = (agent, env_base + e, t[e] >> 1, episode[e]); key = (seed.lo ^ 0x52414E44, seed.hi)),
 *   words 2 (t & 1), 2 (t & 1) + 1 -> a = -1 + (w >> 8) * 2^-23   (uniform on the 2^24-point grid of [-1, 1))
 * keyed by the env's own step and episode counters: no action pool is read (44 B per agent-step instead of 52),
 * results do not depend on how the env axis is sharded or on T.  ctl is required (seed, env_base, episode; acc
 * and auto_reset optional).  act_out ([T][E][N][2], may be NULL) records the actions drawn.  Other buffers as
 * dronesim_rollout; any of reward / true_reward / n_coll may be NULL.                                         */
int dronesim_rollout_random(const DroneParams *p, const DroneEpisodeCtl *ctl, float *pos, float *vel, int32_t *t,
                            float *act_out, float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                            int32_t *n_coll, uint8_t *done, int E, int T, void *stream);

/* dronesim_reset that also retires the episode records of the envs it resets (those with ep_len > 0).        */
int dronesim_reset_ex(const DroneParams *p, const DroneEpisodeCtl *ctl, const uint8_t *mask,
                      float *pos, float *vel, int32_t *t, int32_t *node_out, int E, void *stream);

/* env.reset() as ONE launch (drone_env.py:98-102 -> init_agents :171-212 -> rewards :208): draws the lattice nodes exactly
 * as dronesim_reset does (same Philox stream, same acceptance rule: node ids bit-identical), writes pos / vel = 0 /
 * t = 0 / episode += 1, retires the episode records of the envs it resets when ctl->acc is set (as dronesim_reset_ex), and
 * computes the first observation z / nbr_idx of the new state like dronesim_observe -- from registers and LDS, without a
 * second launch or a round trip of the state through HBM.  ctl supplies div_x, div_y, pitch, seed, env_base, episode
 * (required) and acc (optional); mask as in dronesim_reset; node_out ([E][N], may be NULL) records the nodes drawn.      */
int dronesim_reset_observe(const DroneParams *p, const DroneEpisodeCtl *ctl, const uint8_t *mask,
                           float *pos, float *vel, int32_t *t, int32_t *node_out,
                           float *z, int32_t *nbr_idx, int E, void *stream);

/* out[0..7] = sums over the E records of (done_return, done_true_return, done_collisions, done_len, episodes,
 * ep_return, ep_true_return, ep_len), float64, one launch, fixed summation order (bit-reproducible).          */
#define DRONESIM_EPISODE_REDUCE_DOUBLES 8
int dronesim_episode_reduce(const DroneEpisodeAcc *acc, int E, double *out, void *stream);

/* Classical controllers, batched (deterministic action sources for rollouts and tests):
 *   kind DRONESIM_CONTROL_PROPORTIONAL  proportional_control(state, env)      drone_env.py:652-679
 *        u = k_gain (xF - x), norm capped at u_max (reference: k_gain = 1, u_max = 1)
 *   kind DRONESIM_CONTROL_GRADIENT      gradient_control(state, env, u_max)   drone_env.py:609-650
 *        u = clip(-(2 (x - xF) - 0.1 sum_{j != i, d_ij <= dhat_i} (x_i - x_j) / (d_ij |x_i - x_j|)), +-u_max)
 * pos [E][N][2] in, act [E][N][2] out.  Reads p->N, xF, xF_lo, d_hat, radius; with d_hat_max / radius_max set (> 0 / >= 0)
 * the gradient controller of envs of >= 40 agents finds its partners through the step kernel's cell-mask far filter.  */
#define DRONESIM_CONTROL_PROPORTIONAL 0
#define DRONESIM_CONTROL_GRADIENT 1
int dronesim_control(const DroneParams *p, int kind, const float *pos, float *act, float u_max,
                     int E, void *stream);

/* The statistic the rollout loop logs per step (train_problem.py:98-100, 118-120), accumulated on the device:
 *   acc[0] += sum reward, acc[1] += sum true_reward, acc[2] += sum n_coll, acc[3] += E N, acc[4] += E   (float64)
 * reward / true_reward [E][N], n_coll [E] as written by dronesim_step.  One launch, fixed summation order
 * (bit-reproducible).  scratch: DRONESIM_STATS_SCRATCH_DOUBLES doubles of device memory, zero-initialised once by
 * the caller and owned by one accumulator.  Multi-GPU runs all-gather `acc` (the path's only exchange).            */
#define DRONESIM_STATS_SCRATCH_DOUBLES 769
int dronesim_episode_stats(const float *reward, const float *true_reward, const int32_t *n_coll, int E, int N,
                           double *acc, double *scratch, void *stream);

/* Learner-side reductions over a stored rollout, buffers laid out [T][E][N] like the
 * outputs of dronesim_rollout / T calls of dronesim_step:
 *   dronesim_returns    Monte-Carlo return  G[t] = r[t] + gamma G[t+1],  G[T-1] = r[T-1]
 *                       (SAC_agents.py:304-307); `done` ([T][E] uint8, may be NULL) restarts the scan:
 *                       G[t] = r[t] where done[t] != 0.
 *   dronesim_advantage  weight of the actor loss  w[t,i] = gamma^t / N * sum_{j in Ni[t]} (G[t,j] - V[t,i])
 *                       (SAC_agents.py:333-351); nbr_idx [T][E][N][K1] is the neighbour list the action was
 *                       based on (slot 0 = i, -1 = empty); with `done` the exponent restarts after every
 *                       episode end.                                                            */
int dronesim_returns(const float *reward, const uint8_t *done, float gamma, float *G,
                     int T, int E, int N, void *stream);
int dronesim_advantage(const float *G, const float *V, const int32_t *nbr_idx, const uint8_t *done,
                       float gamma, float *w, int T, int E, int N, int K1, void *stream);

/* Batched per-agent policy / critic forward: N independent 3-layer MLPs, one per agent,
 * evaluated on x[E][N][d_in] in one launch on the matrix cores in exact float32.
 *   DiscreteSoftmaxNN  utils.py:255-309   d_in -> 300 relu -> 300 relu -> n_actions softmax; sample_kind 1
 *                      draws the action index and returns the unit vector at angle 2 pi a / n_actions
 *   NormalActorNN      utils.py:55-117    d_in -> 400 relu -> (200 | 200) relu -> (tanh mu[2] | sigmoid var[2]):
 *                      pass the two heads concatenated as one 400-wide second layer and a block-diagonal
 *                      [400][4] output matrix; sample_kind 2 draws a ~ N(mu, sqrt(var))
 *   CriticNN           utils.py:14-53     d_in -> 200 relu -> 200 relu -> 1, out_kind 0
 * Weights are stacked per agent, "in x out" row-major: w1 [N][d_in][h1], b1 [N][h1], w2 [N][h1][h2],
 * b2 [N][h2], w3 [N][h2][nout], b3 [N][nout] (torch Linear stores [out][in]: transpose when importing).
 * out [E][N][nout] (post-activation, may be NULL), act [E][N][2] and act_idx [E][N] (may be NULL).
 * Random stream: Philox4x32-10 keyed by (seed, env_base + e, agent, counter.lo + t[e],
 * counter.hi + episode[e]); t / episode (int32 [E], device, may be NULL) are the env's own step and
 * episode counters, so a captured hipGraph draws fresh numbers on every replay.                      */
typedef struct DroneMlp {
    int32_t N, d_in, h1, h2, nout;
    int32_t out_kind;       /* 0 identity, 1 softmax, 2 tanh(first half) + sigmoid(second half).  Accuracy contract of kinds 1 / 2
                             * and of the sampling (round 5 on): the activations use the hardware's 1-ulp exp2 / rcp / log2 / sqrt /
                             * sin / cos -- ABSOLUTE error <= 1e-6 per output (tanh as 1 - 2 / (e^2y + 1): no relative accuracy
                             * for |y| << 1), so sampled actions are reproducible run to run but not bit-comparable with a libm build */
    int32_t sample_kind;    /* 0 none, 1 categorical -> unit-circle action, 2 Gaussian          */
    int32_t w2_layout;      /* 0: w2 = [N][h1][h2] (the reference's layout transposed, like w1 / w3);
                             * 1: w2 = float32 matrix-core fragments [N][ceil(h2/32)][ceil(h1/16)][2][64][4] with
                             *    w2[a][c][s][q][l][j] = W2_a[16 s + 8 (l >> 5) + 4 q + j][32 c + (l & 31)], zero beyond
                             *    h1 / h2 -- packed once per weight update, read with 16-byte coalesced loads (the fast
                             *    path of rounds 3-5).  Same arithmetic, float32 throughout.
                             *    The packed array must be 16-byte aligned (EINVAL otherwise).
                             * 2: (0.6.0; d_in <= 14; what the host class BatchedMLP passes) w2 = ONE stream per agent holding
                             *    W1, b1, W2 and W3 in the consumption order of the row-tile kernel (a wave owns 32 env rows
                             *    and every output chunk; layers meet in registers): [N][B][4][64][4] float32 with
                             *    B = dronesim_mlp_rt_blocks(h1, h2, nout) blocks of four 1-KiB pieces [lane = 32 half + i][4]:
                             *      passes P = ceil(C2 / 7), chunks per pass = ceil(C2 / P), C1 = ceil(h1/32), C2 = ceil(h2/32);
                             *      per pass p (output chunks S_p): for c1 < C1: L1(c1), L2(c1, c2) for c2 in S_p; then -- nout > 4
                             *      only -- L3(c2) for c2 in S_p; behind the last pass 4 zero blocks;
                             *      L1(c1): piece 0 = W1[2 r + half][32 c1 + i], r = 0..3; piece 1 = the same for r = 4..6,
                             *              then b1[32 c1 + i] in lanes < 32; pieces 2, 3 zero;
                             *      L2(c1, c2): piece q = W2[32 c1 + 8 q + 4 half + j][32 c2 + i], j = 0..3;
                             *      L3(c2):     piece q = W3[32 c2 + 8 q + 4 half + j][i]       (zero beyond d_in / h1 / h2 / nout).
                             *    w1, b1 are not read (may be NULL); b2, b3 as always; w3 = the plain [N][h2][nout] array when
                             *    nout <= 4 (layer 3 then runs on the vector ALU and the stream holds no L3 blocks), not read
                             *    otherwise.  Same arithmetic (float32 fmaf chains) in a different summation order; 16-byte aligned. */
    const float *w1, *b1, *w2, *b2, *w3, *b3;
} DroneMlp;
int dronesim_mlp_rt_blocks(int h1, int h2, int nout);   /* blocks (4 KiB) per agent of the w2_layout = 2 stream              */
int dronesim_mlp_forward(const DroneMlp *m, const float *x, float *out, float *act, int32_t *act_idx,
                         uint64_t seed, uint64_t counter, int64_t env_base,
                         const int32_t *t, const int32_t *episode, int E, void *stream);

/* Opt-in bfloat16 variant of dronesim_mlp_forward (weights and activations in bf16, float32 accumulation,
 * ~16x the matrix rate; outputs agree with the float32 path to bf16 round-off, ~1e-2 relative).
 * Weights are pre-packed per matrix-core fragment: for a layer with K inputs and F outputs,
 *   wp[agent][c][s][lane][j] = W[kmap(s, lane >> 5, j)][32 c + (lane & 31)]   (0 beyond K or F)
 * for feature chunks c < ceil(F/32), k-steps s and j < 8, as bf16 (16 bytes per lane), with
 *   layers 1, 2:  kmap(s, h, j) = 16 s + 8 h + j
 *   layer 3:      kmap(s, h, j) = 16 s + 8 (j >> 2) + 4 h + (j & 3)   (the order in which a lane of the
 *                 layer-2 accumulator tile holds its features, so layer 3 is fed from registers).
 * k-steps: layer 1: 1 (d_in <= 16), layer 2: 2 ceil(h1/32), layer 3: 2 ceil(h2/32) with a single chunk
 * (nout <= 32).  */
typedef struct DroneMlpBf16 {
    int32_t N, d_in, h1, h2, nout, out_kind, sample_kind, reserved;
    const void *w1p, *w2p, *w3p;     /* packed bf16 fragments */
    const float *b1, *b2, *b3;       /* [N][h1], [N][h2], [N][nout] float32 */
    /* dronesim_mlp_forward_f16x2 only (NULL = all ones; ignored by the other entry points): [N][3] float32, POWERS OF TWO.
     * wscale[i][l] says that the packed image of layer l of agent i holds the weights MULTIPLIED by wscale[i][l] before
     * they were split; the kernel multiplies the layer's accumulators by 1 / wscale[i][l] (exact) before the bias-free
     * part meets the next layer.  A float16 part below 2^-14 is subnormal and keeps an ABSOLUTE 2^-24: an unscaled weight
     * of 0.05 is represented to 6e-7 of itself instead of 2^-22, which large activations multiply -- scaled so that the
     * layer's largest weight sits near 2^14, every weight keeps its 22 bits (round 5).  The factors live in DEVICE memory,
     * so the entry point cannot validate them: anything but a (normal) power of two makes the inverse inexact.
     * ABI NOTE: this field was appended in 0.5.0 -- a caller built against the 0.4.0 header passes a SHORTER struct and
     * must be rebuilt (dronesim_version() >= 500) before it calls dronesim_mlp_forward_f16x2.                         */
    const float *wscale;
} DroneMlpBf16;
int dronesim_mlp_forward_bf16(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                              uint64_t seed, uint64_t counter, int64_t env_base,
                              const int32_t *t, const int32_t *episode, int E, void *stream);

/* float32-ACCURATE variant on the bf16 matrix instructions ("bf16x3"): every weight and activation is split by
 * truncation into three bf16 parts, v = hi + mid + lo exactly, and a product is the float32 sum of its six largest
 * partial products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the rest is below 2^-24 of the product).  Results
 * agree with dronesim_mlp_forward to float32 round-off (same 1e-5 bar against the reference's modules) at 6/16 of
 * its matrix time.  Same struct as the bf16 variant with a different weight image: w1p holds, per (agent, wave w < 4),
 * ONE stream of S = dronesim_mlp_bf16x3_stages(h1, h2) stages of 3 KiB -- a stage = the hi | mid | lo fragments
 * [3][64 lanes][8] bf16 of one (32-feature chunk, k-step), packed as for the bf16 variant with layers 2 AND 3 in the
 * "accumulator" k order (kmap(s, h, j) = 16 s + 8 (j >> 2) + 4 h + (j & 3)) and layer 1 in the linear one -- in the
 * order the kernel consumes them.  With W1(c) = layer 1, chunk c;  W2(c, ss, i) = layer 2, chunk w + 4 i, k-step
 * 2 c + ss;  W3(i, ss) = layer 3, k-step 2 (w + 4 i) + ss;  i running over the wave's chunks (w + 4 i < ceil(h2/32)):
 *     W1(0), W2(0,0,*), W1(1), W2(0,1,*), W2(1,0,*), W1(2), W2(1,1,*), ..., W2(C-1,1,*), W3(0,0), W3(0,1), W3(1,0), ...
 * then zero stages up to S.  [N][4][S][3][64][8] bf16 in all; w2p / w3p are unused, `reserved` must hold S.  */
int dronesim_mlp_bf16x3_stages(int h1, int h2);
int dronesim_mlp_forward_bf16x3(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                                uint64_t seed, uint64_t counter, int64_t env_base,
                                const int32_t *t, const int32_t *episode, int E, void *stream);

/* The same with a two-part float16 split ("f16x2"): v = hi + lo with hi = float16(v), lo = float16(v - hi), exact to
 * 2^-22 of v (the float16 matrix instruction honours subnormal parts), and a product is the float32 sum of hi*hi,
 * hi*lo, lo*hi (the rest is below 2^-22 of the product): float32-level agreement with dronesim_mlp_forward (same 1e-5
 * bar) at 3/16 of its matrix time and 2/3 of the weight bytes of bf16x3.  DOMAIN: every (scaled, see
 * DroneMlpBf16.wscale) weight, input and hidden activation must be below 65504 in magnitude (float16 range) -- beyond
 * it the result is inf / NaN; use bf16x3 or dronesim_mlp_forward for such networks.  Weight image as for bf16x3 with two float16 parts per stage:
 * [N][4][S][2][64][8] float16, 2 KiB per stage.  */
int dronesim_mlp_forward_f16x2(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                               uint64_t seed, uint64_t counter, int64_t env_base,
                               const int32_t *t, const int32_t *episode, int E, void *stream);

/* f16x2 with ROW-TILE ownership (0.6.0; d_in <= 16): a wave owns 32 env rows and every output chunk of layer 2, the four waves of a
 * workgroup share one weight ring; layer 3 runs in exact float32 on the vector ALU when nout <= 4 (the reference's Gaussian actor and
 * critic) and on the matrix cores, from the split of the relu'd layer-2 tiles, otherwise.  Same arithmetic contract as
 * dronesim_mlp_forward_f16x2 (two-part float16 split, three products, float32 accumulation).  DroneMlpBf16 fields as there, except:
 *   w1p      = ONE stream per agent, [N][B][4][64][8] float16 with B = dronesim_mlp_rt16_blocks(h1, h2, nout) blocks of four 1-KiB
 *              pieces ([lane = 32 half + i][8]), weights multiplied by wscale like the split image:
 *                passes P = ceil(C2 / 7), chunks per pass ceil(C2 / P); per pass p (output chunks S_p): for c1 < C1:
 *                  L1(c1)     = (W1 hi, W1 lo, 0, 0): piece[l][j] = part(W1[8 half + j][32 c1 + i])                (one 16-wide k-step)
 *                  L2(c1, c2) = (hi, lo of s = 2 c1), (hi, lo of s = 2 c1 + 1): piece[l][j] = part(W2[16 s + 8 (j >> 2) + 4 half + (j & 3)][32 c2 + i])
 *                for c2 in S_p; then, nout > 4 only, for c2 in S_p:
 *                  L3(c2)     = (hi, lo of s = 2 c2), (hi, lo of s = 2 c2 + 1): piece[l][j] = part(W3[16 s + 8 (j >> 2) + 4 half + (j & 3)][i])
 *                (outputs i >= nout zero); zero blocks up to B;
 *   w3p      = nout <= 4: the plain float32 [N][h2][nout] output layer (NOT multiplied by wscale; wscale[i][2] is not used);
 *              nout > 4: not read;
 *   reserved = dronesim_mlp_rt16_blocks(h1, h2, nout).                                                                       */
int dronesim_mlp_rt16_blocks(int h1, int h2, int nout);
int dronesim_mlp_forward_f16x2_rt(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                                  uint64_t seed, uint64_t counter, int64_t env_base,
                                  const int32_t *t, const int32_t *episode, int E, void *stream);

const char *dronesim_last_error(void);
const char *dronesim_error_string(int code);
int dronesim_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DRONESIM_H */
