// dronesim.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI (include/dronesim.h) of the
// batched drone_env hot path.  Written for wave64 / LDS / HBM3E; no MFMA (the path has
// no dense contraction), no CUDA compatibility layer.
//
// Reference semantics (paths relative to /root/reference/):
//   integrator              drone_env.py:227-238      x' = x + dt*u, v' = u
//   distance_data           drone_env.py:295-334      d_ij = min(|xi-xj|-li-lj, dhat_i), ...
//   rewards                 drone_env.py:260-293
//   localized_states        drone_env.py:336-401      k nearest by d_ij, ghost rows, Ni
//   termination / t         drone_env.py:247-258
//   reset / init_agents     drone_env.py:98-102, 171-212
//
// Work decomposition (one launch = one env.step() for E envs)
//   N <= 64 : lane = agent; floor(64/N) envs are packed into one wave, 4 waves per
//             workgroup.  N = 64 -> one wave per env, N = 5 -> 12 envs per wave.
//   N  > 64 : one workgroup per env, thread = agent (N <= 1024).
//   The integrated positions of the workgroup's envs are staged once in LDS, stored
//   TWICE back to back per env (x_0..x_{N-1}, x_0..x_{N-1}) so that lane i reads its
//   r-th partner j = (i + r) mod N at the wrap-free address base_i + r: the unrolled
//   pair loop has immediate offsets only, consecutive lanes hit consecutive banks
//   (conflict-free ds_read_b64), and the self pair r = 0 is never visited.
//   Pass 1 ("far filter", ~6 VALU/pair): squared distance against the row's
//   early-out radius (dhat_i + l_i + l_max)^2.  A pair beyond it has d_ij = dhat_i,
//   log term 0, no collision, and -- when max(Delta) < min(dhat), the regime of every
//   config in BASELINE.json -- is outside every Delta mask, so it contributes nothing.
//   Survivors are recorded as one bit per partner in a per-lane 32-bit mask.
//   Pass 2 ("near pairs"): each lane walks ITS OWN set bits, so a wave spends
//   max-over-lanes(popcount) iterations instead of one per partner; only here are
//   sqrt / log / the Delta mask / the (k+1)-entry sorted neighbour list evaluated.
//   When far agents can matter (z rows carry v,l of tie-ordered agents for c = 5, or
//   Delta >= dhat), the FAR variant sends every pair through pass 2 (exact general
//   semantics, slower).
//   Ordering: the neighbour list is ordered by (d_ij, j) lexicographically = the
//   first k+1 entries of a stable argsort of row i, independent of visiting order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dronesim.h"

namespace {

constexpr int kWave = 64;
constexpr int kChunk = 16;         // partners whose LDS reads are in flight together (pass 1)
constexpr int kPad = kChunk;       // LDS slack so the last chunk may over-read
constexpr float kLn2 = 0.693147180559945309f;

enum Mode { kStep = 0, kObserve = 1, kRollout = 2 };

struct KArgs {
    int N, c, max_steps, E, T;
    int P, epb;                     // envs per wave (0 when N > 64), envs per workgroup
    float dt, q, b, done_radius, ghost_factor, radius_max;
    const float *xF, *d_hat, *delta, *radius;
    float *pos, *vel;
    int *t;
    const float *act;
    float *reward, *true_reward, *z;
    int *nbr_idx, *n_coll;
    uint8_t *done;
    const uint8_t *mask;
};

// (d, j) as ONE unsigned key whose integer order is the lexicographic order of the pair:
// high word = order-preserving image of the float d, low word = j.
__device__ __forceinline__ unsigned long long nbr_key(float d, int j)
{
    const unsigned b = __float_as_uint(d);
    const unsigned o = b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned)j;
}

// insert `key` into the ascending list of K+1 keys (first K+1 entries of a stable argsort)
template <int K>
__device__ __forceinline__ void nbr_insert(unsigned long long (&list)[K + 1], unsigned long long key)
{
#pragma unroll
    for (int s = 0; s <= K; ++s) {
        const bool lt = key < list[s];
        const unsigned long long cur = list[s];
        list[s] = lt ? key : cur;
        key = lt ? cur : key;
    }
}

__device__ __forceinline__ float nan_to_num_f32(float x)   // np.nan_to_num, drone_env.py:287-288
{
    if (x != x) return 0.0f;
    return fminf(fmaxf(x, -3.402823466e+38f), 3.402823466e+38f);
}

template <int K, bool FAR, int MODE, int MAXT>
__global__ void __launch_bounds__(MAXT) drone_kernel(const KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N;
    const int stride = 2 * N + kPad;                         // float2 per env slot
    float2 *spos = reinterpret_cast<float2 *>(smem);                                       // [epb][stride]
    float2 *sconst = spos + (size_t)a.epb * stride;                                        // [N] (Delta_j, l_j)
    int *sred = reinterpret_cast<int *>(sconst + N);                                       // [epb][2]

    const int tid = threadIdx.x;
    int slot, agent;
    bool valid;
    if (a.P > 0) {                                           // one wave per workgroup, P envs packed in it
        const int sub = tid / N;
        slot = sub;
        agent = tid - sub * N;
        valid = sub < a.P;
    } else {
        slot = 0;
        agent = tid;
        valid = tid < N;
    }
    const int env = blockIdx.x * a.epb + slot;
    valid = valid && env < a.E;
    if (MODE == kObserve && valid && a.mask != nullptr) valid = a.mask[env] != 0;

    for (int s = tid; s < 2 * a.epb; s += blockDim.x) sred[s] = 0;
    for (int s = tid; s < N; s += blockDim.x) sconst[s] = make_float2(a.delta[s], a.radius[s]);

    // per-agent constants (shared by all envs; L2 resident)
    float xFx = 0.f, xFy = 0.f, dhat = 1.f, delta_i = 0.f, li = 0.f;
    if (valid) {
        const float2 g = reinterpret_cast<const float2 *>(a.xF)[agent];
        xFx = g.x; xFy = g.y;
        dhat = a.d_hat[agent];
        delta_i = a.delta[agent];
        li = a.radius[agent];
    }
    const float reach = dhat + li + a.radius_max;
    const float thr = reach * reach * 1.000001f;             // early-out radius^2 (conservative)
    const float log2_dhat = __builtin_amdgcn_logf(dhat);     // v_log_f32 = log2
    const size_t ga = (size_t)env * N + agent;               // global agent index
    const size_t step_agents = (size_t)a.E * N;              // rollout: per-step output stride

    float xi = 0.f, yi = 0.f, vxi = 0.f, vyi = 0.f;
    int tcur = 0;
    if (valid) {
        const float2 p = reinterpret_cast<const float2 *>(a.pos)[ga];
        xi = p.x; yi = p.y;
        if (MODE == kObserve) {
            const float2 v = reinterpret_cast<const float2 *>(a.vel)[ga];
            vxi = v.x; vyi = v.y;
        } else if (agent == 0) {
            tcur = a.t[env];
        }
    }
    float2 *spos_env = spos + (size_t)slot * stride;
    const int nsteps = (MODE == kRollout) ? a.T : 1;

    for (int step = 0; step < nsteps; ++step) {
        const size_t so = (MODE == kRollout) ? (size_t)step * step_agents : 0;   // output offset (agents)
        const float *velsrc = (MODE == kObserve) ? a.vel : a.act + 2 * so;       // v of other agents
        if (valid) {
            if (MODE != kObserve) {
                const float2 u = reinterpret_cast<const float2 *>(a.act)[so + ga];
                xi = fmaf(a.dt, u.x, xi);                     // drone_env.py:235
                yi = fmaf(a.dt, u.y, yi);
                vxi = u.x; vyi = u.y;                         // drone_env.py:238
            }
            spos_env[agent] = make_float2(xi, yi);
            spos_env[agent + N] = make_float2(xi, yi);
        }
        __syncthreads();

        if (valid) {
            float s_all = 0.f, s_msk = 0.f;
            int ncoll = 0;
            unsigned long long list[K + 1];
#pragma unroll
            for (int s = 0; s <= K; ++s) list[s] = ~0ull;
            // self entry: d_ii = min(-2 l_i, dhat_i), ratio 1 -> log 0, never a collision (:323-325)
            const float dii = fminf(-li - li, dhat);
            list[0] = nbr_key(dii, agent);
            int in_range = ((dii <= delta_i) ? 1 : 0) - 1;    // :346 (N_delta[i,i] uses Delta_i), minus itself

            for (int r0 = 1; r0 < N; r0 += 64) {
                // ---- pass 1: far filter over up to 64 partners, 16 LDS reads in flight
                unsigned long long near = 0ull;
                const int left = N - r0;
                if (FAR) {
                    near = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
                } else {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const int cnt = left - c4 * kChunk;
                        if (cnt > 0) {
                            const float2 *pp = spos_env + agent + r0 + c4 * kChunk;
                            float2 pj[kChunk];
#pragma unroll
                            for (int u = 0; u < kChunk; ++u) pj[u] = pp[u];
                            unsigned m = 0u;
#pragma unroll
                            for (int u = 0; u < kChunk; ++u) {
                                const float dx = xi - pj[u].x, dy = yi - pj[u].y;
                                const float d2 = fmaf(dy, dy, dx * dx);
                                m |= (d2 < thr ? 1u : 0u) << u;
                            }
                            if (cnt < kChunk) m &= (1u << cnt) - 1u;
                            near |= (unsigned long long)m << (c4 * kChunk);
                        }
                    }
                }
                // ---- pass 2: every lane walks its own surviving partners
                while (near) {
                    const int u = __builtin_ctzll(near);
                    near &= near - 1ull;
                    int j = agent + r0 + u;
                    const float2 pj = spos_env[j];
                    j -= (j >= N) ? N : 0;
                    const float2 cj = sconst[j];                              // (Delta_j, l_j)
                    const float dx = xi - pj.x, dy = yi - pj.y;
                    const float dist = __builtin_amdgcn_sqrtf(fmaf(dy, dy, dx * dx));
                    float d = fminf(dist - li - cj.y, dhat);                  // :318
                    d = (d == 0.0f) ? -1e-6f : d;                             // :319-320
                    const bool coll = d < 0.0f;                               // :327 (dhat > 0)
                    // log(dhat/d) = ln2 * (log2 dhat - log2 d); collisions contribute 9990 (:330-332)
                    const float lg = coll ? 9.99e3f : kLn2 * (log2_dhat - __builtin_amdgcn_logf(d));
                    const bool inm = d <= cj.x;                               // :328 (Delta_j!)
                    s_all += lg;                                              // :283
                    s_msk += inm ? lg : 0.0f;                                 // :282
                    ncoll += coll ? 1 : 0;                                    // :284
                    in_range += inm ? 1 : 0;
                    nbr_insert<K>(list, nbr_key(d, j));                       // :338
                }
            }

            // rewards (:276, :287-288)
            const float gx = xFx - xi, gy = xFy - yi;
            const float err2 = fmaf(gy, gy, gx * gx);
            const float to_goal = a.q * err2;
            if (a.reward) a.reward[so + ga] = -nan_to_num_f32(fmaf(a.b, s_msk, to_goal));
            if (a.true_reward) a.true_reward[so + ga] = -nan_to_num_f32(fmaf(a.b, s_all, to_goal));

            // localized state rows + neighbour list (:344-397)
            const int c = a.c;
            float *zr = a.z + (so + ga) * (size_t)((K + 1) * c);
            int *nb = a.nbr_idx + (so + ga) * (size_t)(K + 1);
            const float zx = xi - xFx, zy = yi - xFy;                         // :357
            const float gsc = __builtin_amdgcn_rsqf(err2) * delta_i * a.ghost_factor;
            const float ghx = zx * gsc, ghy = zy * gsc;                       // :386 (NaN when on the goal)
            if (c == 2) {
                reinterpret_cast<float2 *>(zr)[0] = make_float2(zx, zy);
            } else {
                zr[0] = zx; zr[1] = zy; zr[2] = vxi; zr[3] = vyi; zr[4] = li;
            }
            nb[0] = agent;
#pragma unroll
            for (int kth = 1; kth <= K; ++kth) {
                const unsigned j = (unsigned)list[kth];
                const bool have = j < (unsigned)N;
                const bool real = kth <= in_range && have;                    // :362
                float rx = ghx, ry = ghy;
                if (real) {
                    const float2 pj = spos_env[j];
                    rx = pj.x - xi; ry = pj.y - yi;                           // :368
                }
                nb[kth] = real ? (int)j : -1;
                if (c == 2) {
                    reinterpret_cast<float2 *>(zr)[kth] = make_float2(rx, ry);
                } else {
                    float *row = zr + kth * 5;
                    row[0] = rx; row[1] = ry;
                    if (have) {                                               // :367 / :385
                        const float2 vj = reinterpret_cast<const float2 *>(velsrc)[(size_t)env * N + j];
                        row[2] = vj.x; row[3] = vj.y; row[4] = sconst[j].y;
                    } else {
                        row[2] = row[3] = row[4] = __builtin_nanf("");
                    }
                }
            }

            if (MODE != kObserve) {
                if (MODE != kRollout || step == nsteps - 1) {                 // final state only
                    reinterpret_cast<float2 *>(a.pos)[ga] = make_float2(xi, yi);
                    reinterpret_cast<float2 *>(a.vel)[ga] = make_float2(vxi, vyi);
                }
                if (!(__builtin_amdgcn_sqrtf(err2) <= a.done_radius)) atomicOr(&sred[2 * slot + 1], 1);   // :249-251
            }
            if (ncoll) atomicAdd(&sred[2 * slot], ncoll);
        }
        __syncthreads();
        if (valid && agent == 0) {
            const size_t eo = (MODE == kRollout) ? (size_t)step * a.E + env : (size_t)env;
            if (a.n_coll) a.n_coll[eo] = sred[2 * slot];
            if (MODE != kObserve) {
                a.done[eo] = (uint8_t)((sred[2 * slot + 1] == 0) || (tcur >= a.max_steps - 1));   // :251
                tcur += 1;                                                                        // :256
            }
            sred[2 * slot] = 0;
            sred[2 * slot + 1] = 0;
        }
    }
    if (MODE != kObserve && valid && agent == 0) a.t[env] = tcur;
}

// ---------------------------------------------------------------------------------------
// reset: N distinct lattice nodes per env by parallel rejection on a Philox4x32-10 stream
// (restated integer-exactly on the CPU in oracle/drone_oracle.c:oracle_reset).

__device__ __forceinline__ uint32_t philox4x32_10_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

struct RArgs {
    int N, E, P, epb, div_y;
    uint32_t M, key0, key1;
    long long env_base;
    float pitch;
    const uint8_t *mask;
    float *pos, *vel;
    int *t, *episode, *node_out;
};

__global__ void __launch_bounds__(1024) reset_kernel(const RArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N;
    int *snode = reinterpret_cast<int *>(smem);            // [epb][N] settled node or -1
    int *scand = snode + (size_t)a.epb * N;                // [epb][N] proposal of this round or -1
    const int tid = threadIdx.x;
    int slot, agent;
    bool valid;
    if (a.P > 0) {
        const int sub = tid / N;
        slot = sub;
        agent = tid - sub * N;
        valid = sub < a.P;
    } else {
        slot = 0; agent = tid; valid = tid < N;
    }
    const int env = blockIdx.x * a.epb + slot;
    valid = valid && env < a.E;
    if (valid && a.mask != nullptr) valid = a.mask[env] != 0;
    const uint32_t gid = (uint32_t)(a.env_base + env);
    const uint32_t epi = valid ? (uint32_t)a.episode[env] : 0u;   // resets this env has seen so far
    int *mynode = snode + (size_t)slot * N, *mycand = scand + (size_t)slot * N;

    int node = -1;
    int remaining;
    uint32_t round = 0;
    do {
        int cand = -1;
        if (valid) {
            if (node < 0) {
                const uint32_t w = philox4x32_10_word0((uint32_t)agent, round, gid, epi, a.key0, a.key1);
                cand = (int)__umulhi(w, a.M);
            }
            mynode[agent] = node;
            mycand[agent] = cand;
        }
        __syncthreads();
        if (valid && node < 0) {
            bool ok = true;
            for (int j = 0; j < N; ++j) {
                if (j == agent) continue;
                const int nj = mynode[j];
                if (nj >= 0) ok = ok && (nj != cand);                 // held since an earlier round
                else if (j < agent) ok = ok && (mycand[j] != cand);   // lower index wins the round
            }
            if (ok) node = cand;
        }
        remaining = __syncthreads_count(valid && node < 0);
        ++round;
    } while (remaining > 0 && round < (1u << 20));

    if (valid && node >= 0) {
        const size_t ga = (size_t)env * N + agent;
        const int idx = node / a.div_y, jdx = node - idx * a.div_y;
        reinterpret_cast<float2 *>(a.pos)[ga] = make_float2((float)idx * a.pitch, (float)jdx * a.pitch);
        reinterpret_cast<float2 *>(a.vel)[ga] = make_float2(0.f, 0.f);   // drone_env.py:189
        if (a.node_out) a.node_out[ga] = node;
        if (agent == 0) {
            a.t[env] = 0;                                                // drone_env.py:100
            a.episode[env] = (int)(epi + 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

struct Geometry {
    int P, epb, threads, blocks;
    size_t lds;
};

Geometry geometry(int N, int E)
{
    Geometry g;
    if (N <= kWave) {                  // one wave = one workgroup: barriers are free, waves run unlocked
        g.P = kWave / N;
        g.threads = kWave;
        g.epb = g.P;
    } else {
        g.P = 0;
        g.threads = ((N + kWave - 1) / kWave) * kWave;
        g.epb = 1;
    }
    g.blocks = (E + g.epb - 1) / g.epb;
    g.lds = sizeof(float2) * ((size_t)g.epb * (2 * (size_t)N + kPad) + (size_t)N) + sizeof(int) * 2 * (size_t)g.epb;
    return g;
}

int check_params(const DroneParams *p, int E)
{
    if (!p) return fail(DRONESIM_EINVAL, "params is NULL");
    if (E < 0) return fail(DRONESIM_EINVAL, "E < 0");
    if (p->N < 2 || p->N > DRONESIM_MAX_AGENTS) return fail(DRONESIM_EUNSUPPORTED, "N must be in 2..1024");
    if (p->k < 1 || p->k > p->N - 1) return fail(DRONESIM_EINVAL, "k_closest must be in 1..N-1");
    if (p->k > DRONESIM_MAX_K) return fail(DRONESIM_EUNSUPPORTED, "k_closest > DRONESIM_MAX_K");
    if (p->c != 2 && p->c != 5) return fail(DRONESIM_EINVAL, "c must be 2 or 5");
    if (!p->xF || !p->d_hat || !p->delta || !p->radius) return fail(DRONESIM_EINVAL, "constant array is NULL");
    if (!(p->d_hat_min > 0.0f)) return fail(DRONESIM_EINVAL, "d_hat_min must be > 0");
    return DRONESIM_OK;
}

template <int K, bool FAR, int MAXT>
void launch_mode(int mode, const KArgs &a, const Geometry &g, hipStream_t s)
{
    const dim3 grid(g.blocks), block(g.threads);
    switch (mode) {
    case kStep: hipLaunchKernelGGL((drone_kernel<K, FAR, kStep, MAXT>), grid, block, g.lds, s, a); break;
    case kObserve: hipLaunchKernelGGL((drone_kernel<K, FAR, kObserve, MAXT>), grid, block, g.lds, s, a); break;
    default: hipLaunchKernelGGL((drone_kernel<K, FAR, kRollout, MAXT>), grid, block, g.lds, s, a); break;
    }
}

template <int K>
void launch_k(int mode, bool far, const KArgs &a, const Geometry &g, hipStream_t s)
{
    // register budget follows the workgroup size: one wave for N <= 64, 256 threads for N <= 256
    // (covers every BASELINE config), 1024 threads only for the largest envs
    if (g.threads <= 64) {
        if (far) launch_mode<K, true, 64>(mode, a, g, s);
        else launch_mode<K, false, 64>(mode, a, g, s);
    } else if (g.threads <= 256) {
        if (far) launch_mode<K, true, 256>(mode, a, g, s);
        else launch_mode<K, false, 256>(mode, a, g, s);
    } else {
        if (far) launch_mode<K, true, 1024>(mode, a, g, s);
        else launch_mode<K, false, 1024>(mode, a, g, s);
    }
}

int launch(int mode, const DroneParams *p, KArgs &a, int E, void *stream)
{
    if (E == 0) return DRONESIM_OK;
    const Geometry g = geometry(p->N, E);
    a.N = p->N; a.c = p->c; a.max_steps = p->max_steps; a.E = E;
    a.P = g.P; a.epb = g.epb;
    a.dt = p->dt; a.q = p->q; a.b = p->b; a.done_radius = p->done_radius;
    a.ghost_factor = p->ghost_factor; a.radius_max = p->radius_max;
    a.xF = p->xF; a.d_hat = p->d_hat; a.delta = p->delta; a.radius = p->radius;
    // far agents matter when a z row carries (v, l) of a tie-ordered agent (c = 5) or
    // when a clipped distance can pass a Delta mask (Delta_j >= dhat_i possible)
    const bool far = (p->c == 5) || !(p->delta_max < p->d_hat_min);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (p->k) {
    case 1: launch_k<1>(mode, far, a, g, s); break;
    case 2: launch_k<2>(mode, far, a, g, s); break;
    case 3: launch_k<3>(mode, far, a, g, s); break;
    case 4: launch_k<4>(mode, far, a, g, s); break;
    case 5: launch_k<5>(mode, far, a, g, s); break;
    case 6: launch_k<6>(mode, far, a, g, s); break;
    case 7: launch_k<7>(mode, far, a, g, s); break;
    case 8: launch_k<8>(mode, far, a, g, s); break;
    default: return fail(DRONESIM_EUNSUPPORTED, "k_closest > DRONESIM_MAX_K");
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

}   // namespace

extern "C" {

int dronesim_step(const DroneParams *p, float *pos, float *vel, int32_t *t, const float *act,
                  float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                  int32_t *n_coll, uint8_t *done, int E, void *stream)
{
    const int rc = check_params(p, E);
    if (rc) return rc;
    if (!pos || !vel || !t || !act || !z || !nbr_idx || !done)
        return fail(DRONESIM_EINVAL, "dronesim_step: required buffer is NULL");
    KArgs a{};
    a.pos = pos; a.vel = vel; a.t = t; a.act = act; a.reward = reward; a.true_reward = true_reward;
    a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll; a.done = done; a.T = 1;
    return launch(kStep, p, a, E, stream);
}

int dronesim_observe(const DroneParams *p, const float *pos, const float *vel,
                     float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                     int32_t *n_coll, const uint8_t *mask, int E, void *stream)
{
    const int rc = check_params(p, E);
    if (rc) return rc;
    if (!pos || !vel || !z || !nbr_idx) return fail(DRONESIM_EINVAL, "dronesim_observe: required buffer is NULL");
    KArgs a{};
    a.pos = const_cast<float *>(pos); a.vel = const_cast<float *>(vel);
    a.reward = reward; a.true_reward = true_reward; a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll;
    a.mask = mask; a.T = 1;
    return launch(kObserve, p, a, E, stream);
}

int dronesim_rollout(const DroneParams *p, float *pos, float *vel, int32_t *t, const float *act,
                     float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                     int32_t *n_coll, uint8_t *done, int E, int T, void *stream)
{
    const int rc = check_params(p, E);
    if (rc) return rc;
    if (T < 0) return fail(DRONESIM_EINVAL, "T < 0");
    if (!pos || !vel || !t || !act || !z || !nbr_idx || !done)
        return fail(DRONESIM_EINVAL, "dronesim_rollout: required buffer is NULL");
    if (T == 0) return DRONESIM_OK;
    KArgs a{};
    a.pos = pos; a.vel = vel; a.t = t; a.act = act; a.reward = reward; a.true_reward = true_reward;
    a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll; a.done = done; a.T = T;
    return launch(kRollout, p, a, E, stream);
}

int dronesim_reset(const DroneParams *p, int div_x, int div_y, float pitch,
                   uint64_t seed, int64_t env_base, const uint8_t *mask,
                   float *pos, float *vel, int32_t *t, int32_t *episode, int32_t *node_out,
                   int E, void *stream)
{
    if (!p) return fail(DRONESIM_EINVAL, "params is NULL");
    if (p->N < 1 || p->N > DRONESIM_MAX_AGENTS) return fail(DRONESIM_EUNSUPPORTED, "N must be in 1..1024");
    if (E < 0 || div_x < 1 || div_y < 1) return fail(DRONESIM_EINVAL, "bad E / lattice size");
    if (!pos || !vel || !t || !episode) return fail(DRONESIM_EINVAL, "dronesim_reset: required buffer is NULL");
    const uint64_t M = (uint64_t)div_x * (uint64_t)div_y;
    if (M < (uint64_t)p->N) return fail(DRONESIM_EINVAL, "lattice has fewer nodes than agents (random.sample would raise)");
    if (M > 0xFFFFFFFFull) return fail(DRONESIM_EUNSUPPORTED, "lattice larger than 2^32 nodes");
    if (E == 0) return DRONESIM_OK;
    const Geometry g = geometry(p->N, E);
    RArgs a{};
    a.N = p->N; a.E = E; a.P = g.P; a.epb = g.epb; a.div_y = div_y;
    a.M = (uint32_t)M;
    a.key0 = (uint32_t)seed;
    a.key1 = (uint32_t)(seed >> 32);
    a.env_base = env_base; a.pitch = pitch; a.mask = mask;
    a.pos = pos; a.vel = vel; a.t = t; a.episode = episode; a.node_out = node_out;
    const size_t lds = sizeof(int) * 2 * (size_t)g.epb * p->N;
    hipLaunchKernelGGL(reset_kernel, dim3(g.blocks), dim3(g.threads), lds, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

const char *dronesim_last_error(void) { return g_err; }

const char *dronesim_error_string(int code)
{
    switch (code) {
    case DRONESIM_OK: return "ok";
    case DRONESIM_EINVAL: return "invalid argument";
    case DRONESIM_EUNSUPPORTED: return "unsupported size (N or k beyond the compiled kernels)";
    case DRONESIM_ELAUNCH: return "HIP launch failure";
    default: return "unknown error";
    }
}

int dronesim_version(void) { return DRONESIM_VERSION; }

}   // extern "C"
