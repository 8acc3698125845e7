// dronesim.hip -- C ABI (include/dronesim.h) of the batched drone_env hot path on gfx950 (MI355X / CDNA4), plus the
// small kernels around the step kernel: reset, classical controllers, learner-side scans, episode reductions.
// The step / observe / rollout kernel itself lives in drone_kernel.hpp and is instantiated per k_closest value in
// drone_kernel_k.hip; this file fills its arguments and dispatches.
#include "drone_kernel.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// reset: N distinct lattice nodes per env by parallel rejection on a Philox4x32-10 stream
// (restated integer-exactly on the CPU in oracle/drone_oracle.c:oracle_reset).

struct RArgs {
    int N, E, P, epb, div_y, nt, shift;     // nt: sampling-table entries per env (2^k >= 2 N), shift = 32 - k
    uint32_t M, key0, key1;
    long long env_base;
    float pitch;
    const uint8_t *mask;
    float *pos, *vel;
    int *t, *episode, *node_out;
    double *acc;                    // DroneEpisodeAcc[E] (8 x 8 bytes per env) or nullptr: retire the episode in progress
};

__global__ void __launch_bounds__(1024) reset_kernel(const RArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N;
    int2 *sst = reinterpret_cast<int2 *>(smem);            // [epb][nt] open-addressing table of (node, owner)
    const int tid = threadIdx.x;
    int slot, agent;
    bool valid;
    if (a.P > 0) {                                           // same lane -> (slot, agent) map as drone_kernel
        const int lane = tid & (kWave - 1);
        const int sub = lane / N;
        slot = (tid >> 6) * a.P + sub;
        agent = lane - sub * N;
        valid = sub < a.P;
    } else {
        slot = 0; agent = tid; valid = tid < N;
    }
    const int env = blockIdx.x * a.epb + slot;
    valid = valid && env < a.E;
    if (valid && a.mask != nullptr) valid = a.mask[env] != 0;
    const uint32_t gid = (uint32_t)(a.env_base + env);
    const uint32_t epi = valid ? (uint32_t)a.episode[env] : 0u;   // resets this env has seen so far
    // Round r: unsettled agent i proposes node mulhi(philox(i, r, env, episode), M) and settles on it iff no settled
    // agent holds that node and no lower-index agent proposed it this round.  The rule is evaluated through an
    // open-addressing table per env (2^k >= 2 N entries, keys compared exactly): settled agents enter their node as
    // blockers (owner -1), proposers enter theirs with owner = min(agent index); a proposer wins iff it owns its entry.
    const int nt = a.nt;
    int2 *tbl = sst + (size_t)slot * nt;

    int node = -1;
    int remaining;
    uint32_t round = 0;
    do {
        if (valid)
            for (int o = agent; o < nt; o += N) tbl[o] = make_int2(-1, 0x7fffffff);
        __syncthreads();
        int cand = node, h = 0;
        if (valid) {
            if (node < 0) {
                const uint32_t w = philox4x32_10_word0((uint32_t)agent, round, gid, epi, a.key0, a.key1);
                cand = (int)__umulhi(w, a.M);
            }
            h = (int)(((uint32_t)cand * 0x9E3779B1u) >> a.shift);
            for (;;) {                                                // linear probing, load factor <= 1/2
                const int old = atomicCAS(&tbl[h].x, -1, cand);
                if (old == -1 || old == cand) break;
                h = (h + 1) & (nt - 1);
            }
            atomicMin(&tbl[h].y, node >= 0 ? -1 : agent);
        }
        __syncthreads();
        if (valid && node < 0 && tbl[h].y == agent) node = cand;
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
            if (a.acc != nullptr) {                                      // train_problem.py:118-121: log, then reset
                double *rec = a.acc + 8 * (size_t)env;
                int *reci = reinterpret_cast<int *>(rec);
                if (reci[5] > 0) {                                       // ep_len: an episode was in progress
                    rec[4] += rec[0]; rec[5] += rec[1];
                    reinterpret_cast<long long *>(rec)[6] += reci[4];
                    reinterpret_cast<long long *>(rec)[7] += reci[5];
                    reci[6] += 1;
                    rec[0] = 0.0; rec[1] = 0.0; reci[4] = 0; reci[5] = 0;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Classical controllers (drone_env.py:609-679), same lane <-> agent geometry as drone_kernel.
// gradient_control sums a repulsion term over the partners inside dhat_i + l_i + l_j (:641-644).  Envs of >= kBucketMinN
// agents find them the way the step kernel's far filter does (round 6): every agent ORs its bit into the mask of its x
// cell and of its y cell (64 hashed cells per axis, at least one reach wide), an agent's candidates are
// (3 x masks) & (3 y masks), and only those get the exact test -- in ascending partner order, so the sum is the one the
// all-partner loop forms (far partners never contributed a term): C3 12.8 -> see DESIGN.md section 7.  Smaller envs keep the loop.
struct CArgs {
    int N, E, P, epb, kind;
    float u_max;
    const float *xF, *xF_lo, *d_hat, *radius, *pos;
    float *act;
    int bucket, tab_off;            // cell-mask filter in use; byte offset of its tables in LDS
    float inv_cell;                 // 1 / cell width
};

// one ordered pair of the repulsion sum (:639-644); `cand` = the conservative reach test of the far filter
__device__ __forceinline__ void gradient_pair(float xi, float yi, float ri, float dhat, float2 pj, float rj, float &t2x, float &t2y)
{
    const float dx = xi - pj.x, dy = yi - pj.y;
    const float d2 = fmaf(dy, dy, dx * dx);
    const float reach = dhat + ri + rj;
    if (d2 <= reach * reach * 1.000001f) {
        const float nrm = sqrtf(d2);
        const float dij = nrm - ri - rj;                                     // :641
        if (dij <= dhat) {                                                   // :643
            const float w = 1.0f / (dij * nrm);
            t2x = fmaf(dx, w, t2x); t2y = fmaf(dy, w, t2y);                  // :644
        }
    }
}

template <bool WL>
__global__ void __launch_bounds__(1024) control_kernel(const CArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const unsigned lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
    int slot, agent, env0, nval;
    if (WL) {
        const int sub = (int)lane / N;
        slot = wave * a.P + sub;
        agent = (int)lane - sub * N;
        env0 = blockIdx.x * a.epb + wave * a.P;
        nval = max(0, min(a.P, a.E - env0)) * N;
    } else {
        slot = 0; agent = tid; env0 = blockIdx.x;
        nval = max(0, min(kWave, N - wave * kWave));
    }
    const size_t wga0 = (size_t)env0 * N + (WL ? 0 : wave * kWave);
    const bool valid = (int)lane < nval;
    const bool gradient = a.kind == DRONESIM_CONTROL_GRADIENT;               // launch-uniform
    const bool bucket = gradient && a.bucket != 0;
    float2 *spos = reinterpret_cast<float2 *>(smem);                         // [epb][N]
    float *srad = reinterpret_cast<float *>(spos + (size_t)a.epb * N);       // [WL ? nwaves : 1][N]
    float *srad_w = srad + (WL ? (size_t)wave * N : 0);
    // cell tables (bucket): [axis][W words of 64 agents][64 cells]; one env per wave (WL: N >= kBucketMinN) or per workgroup
    const int W = WL ? 1 : nwaves;
    unsigned long long *tb = reinterpret_cast<unsigned long long *>(smem + a.tab_off) + (WL ? (size_t)wave * 2 * kCells : 0);
    float xi = 0.f, yi = 0.f, xFx = 0.f, xFy = 0.f, xLx = 0.f, xLy = 0.f, dhat = 1.f, ri = 0.f;
    if (valid) {
        const float2 p = (reinterpret_cast<const float2 *>(a.pos) + wga0)[lane];
        const float2 g = reinterpret_cast<const float2 *>(a.xF)[(unsigned)agent];
        if (a.xF_lo) { const float2 gl = reinterpret_cast<const float2 *>(a.xF_lo)[(unsigned)agent]; xLx = gl.x; xLy = gl.y; }
        dhat = a.d_hat[(unsigned)agent];
        ri = a.radius[(unsigned)agent];
        xi = p.x; yi = p.y; xFx = g.x; xFy = g.y;
        if (gradient) spos[(size_t)slot * N + agent] = p;
    }
    if (gradient) {
        if (WL) { if ((int)lane < N) srad_w[lane] = a.radius[lane]; }
        else for (int s = tid; s < N; s += blockDim.x) srad_w[s] = a.radius[s];
        if (bucket) {
            if (WL) { for (int o = lane; o < 2 * kCells; o += kWave) tb[o] = 0ull; }
            else for (int o = tid; o < 2 * kCells * W; o += blockDim.x) tb[o] = 0ull;
        }
        group_sync<WL>();
    }
    float ux, uy;
    if (!gradient) {
        ux = (xFx - xi) + xLx; uy = (xFy - yi) + xLy;                        // :667-668, k_gain = 1
        const float nrm = sqrtf(fmaf(uy, uy, ux * ux));
        if (nrm > a.u_max) { ux = ux / nrm * a.u_max; uy = uy / nrm * a.u_max; }   // :670-673
    } else {
        const float2 *pe = spos + (size_t)slot * N;
        float t2x = 0.f, t2y = 0.f;
        if (bucket) {
            const int cx = (int)__builtin_floorf(xi * a.inv_cell) & (kCells - 1), cy = (int)__builtin_floorf(yi * a.inv_cell) & (kCells - 1);
            if (valid) {
                atomicOr(&tb[(agent >> 6) * kCells + cx], 1ull << (agent & 63));
                atomicOr(&tb[(W + (agent >> 6)) * kCells + cy], 1ull << (agent & 63));
            }
            group_sync<WL>();
            if (valid) {
                const int cxm = (cx - 1) & (kCells - 1), cxp = (cx + 1) & (kCells - 1);
                const int cym = (cy - 1) & (kCells - 1), cyp = (cy + 1) & (kCells - 1);
                for (int w = 0; w < W; ++w) {                                // ascending partner order, like the loop below
                    const unsigned long long *tx = tb + w * kCells, *ty = tb + (W + w) * kCells;
                    unsigned long long m = (tx[cxm] | tx[cx] | tx[cxp]) & (ty[cym] | ty[cy] | ty[cyp]);
                    if (w == (agent >> 6)) m &= ~(1ull << (agent & 63));
                    while (m) {
                        const int j = 64 * w + __builtin_ctzll(m);
                        m &= m - 1ull;
                        gradient_pair(xi, yi, ri, dhat, pe[j], srad_w[j], t2x, t2y);
                    }
                }
            }
        } else if (valid) {
            for (int j = 0; j < N; ++j)
                if (j != agent) gradient_pair(xi, yi, ri, dhat, pe[j], srad_w[j], t2x, t2y);
        }
        const float gx = 2.0f * ((xi - xFx) - xLx) - 0.1f * t2x, gy = 2.0f * ((yi - xFy) - xLy) - 0.1f * t2y;   // :633, :646
        ux = fminf(fmaxf(-gx, -a.u_max), a.u_max);                           // :647
        uy = fminf(fmaxf(-gy, -a.u_max), a.u_max);
    }
    if (valid) (reinterpret_cast<float2 *>(a.act) + wga0)[lane] = make_float2(ux, uy);
}

// ---------------------------------------------------------------------------------------
// Learner-side scans over a stored rollout [T][E][N] (SAC_agents.py:304-307, 333-351).  One thread per
// (env, agent) column; consecutive threads touch consecutive addresses at every t (coalesced streams).
// One thread per (env, agent) column walks the T axis.  The streams are touched once (streaming loads / stores) and
// kStageT time steps are requested together before the sequential recurrence consumes them.
constexpr int kStageT = 8;

// done[t][e] for the eight time steps t_of(0..7) of a stage, fetched by the WAVE with one load instruction instead of
// eight per thread: lane 8 k + u reads the flag of (step u, k-th env the wave touches), every thread then picks its
// env's flags with a lane permute.  (One 1-byte load per thread and step -- all lanes of a wave on the same address --
// made the returns scan 2x slower than a copy of the same size: 147 us against 74 without the flags at T = 200 x C3.)
// Valid when the wave's columns span at most 8 envs (`coop`, wave-uniform); otherwise threads read their own flags.
struct DoneStage {
    unsigned v;
    template <typename TOf>
    __device__ __forceinline__ void fetch(const uint8_t *done, size_t e_first, int E, int T, TOf t_of)
    {
        const int lane = threadIdx.x & 63, k = lane >> 3, u = lane & 7;
        const int t = t_of(u);
        v = (t >= 0 && t < T && e_first + k < (size_t)E) ? done[(size_t)t * E + e_first + k] : 0u;
    }
    __device__ __forceinline__ bool get(int de, int u) const { return __shfl((int)v, 8 * de + u, 64) != 0; }
};

// The scan is software-pipelined: the rewards (and done flags) of stage s + 1 are requested BEFORE stage s is folded and
// stored, so that a thread always has kStageT .. 2 kStageT loads in flight and the sequential recurrence runs in the shadow
// of the next stage's HBM round trip.  (Round 4 requested a stage, waited for all of it, folded, stored, and only then
// requested the next: 4 waves per SIMD x 8 loads did not cover the round trip -- 84 us = 0.62 of the roofline against
// 74 us for a copy of the same bytes.)  The recurrence itself is unchanged: G[t] = fma(G[t+1], gamma, r[t]), in order.
constexpr int kRetStageT = 8;      // (the done-flag fetch of DoneStage covers 8 steps: see returns_fetch)
// V = 4: a thread owns FOUR adjacent columns (agents of one env: N % 4 == 0) and moves them as 16-byte loads / stores --
// a quarter of the memory instructions for the same bytes in flight (round 5; the un-pipelined round-4 scan lost with
// four columns per thread because a wave then had nothing to overlap its stage wait with).  V = 1: any shape.
template <int V> struct RetVec;
template <> struct RetVec<1> { typedef float type; };
template <> struct RetVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int V> struct RetStage {
    typename RetVec<V>::type r[kRetStageT];
    DoneStage ds;
    bool last[kRetStageT];
};

template <bool COOP, int V>
__device__ __forceinline__ void returns_fetch(RetStage<V> &st, const float *__restrict__ reward, const uint8_t *__restrict__ done,
                                              size_t EN, size_t col, size_t e, size_t e_first, int E, int T, int t0)
{
    typedef typename RetVec<V>::type vec;
    if (COOP) st.ds.fetch(done, e_first, E, T, [&](int u) { return t0 - u; });
#pragma unroll
    for (int u = 0; u < kRetStageT; ++u) {
        const int t = t0 - u;
        st.r[u] = t >= 0 ? __builtin_nontemporal_load(reinterpret_cast<const vec *>(reward + (size_t)t * EN + col)) : vec(0.0f);
        if (!COOP) st.last[u] = t == T - 1 || (done != nullptr && t >= 0 && done[(size_t)t * E + e] != 0);
    }
}

template <bool COOP, int V>
__device__ __forceinline__ void returns_scan(const float *__restrict__ reward, const uint8_t *__restrict__ done, float gamma,
                                             float *__restrict__ G, int T, int E, size_t EN, size_t col, size_t e,
                                             size_t e_first, int de, bool act)
{
    typedef typename RetVec<V>::type vec;
    vec g = vec(0.0f);
    RetStage<V> cur, nxt;
    returns_fetch<COOP, V>(cur, reward, done, EN, col, e, e_first, E, T, T - 1);
    for (int t0 = T - 1; t0 >= 0; t0 -= kRetStageT) {
        if (t0 - kRetStageT >= 0) returns_fetch<COOP, V>(nxt, reward, done, EN, col, e, e_first, E, T, t0 - kRetStageT);
#pragma unroll
        for (int u = 0; u < kRetStageT; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
                const bool last = COOP ? (t == T - 1 || cur.ds.get(de, u)) : cur.last[u];
                if (V == 1) {
                    g = last ? cur.r[u] : vec(fmaf(*reinterpret_cast<float *>(&g), gamma, *reinterpret_cast<const float *>(&cur.r[u])));   // :306  Gt[t] = Gt[t+1]*discount + r[t]
                } else {
                    float *gp = reinterpret_cast<float *>(&g);
                    const float *rp = reinterpret_cast<const float *>(&cur.r[u]);
#pragma unroll
                    for (int q = 0; q < V; ++q) gp[q] = last ? rp[q] : fmaf(gp[q], gamma, rp[q]);                                         // :306
                }
                if (act) __builtin_nontemporal_store(g, reinterpret_cast<vec *>(G + (size_t)t * EN + col));
            }
        }
        cur = nxt;
    }
}

template <int V>
__global__ void __launch_bounds__(256) returns_kernel(const float *__restrict__ reward, const uint8_t *__restrict__ done,
                                                      float gamma, float *__restrict__ G, int T, int E, int N)
{
    const size_t EN = (size_t)E * N;
    const size_t col0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool act = col0 < EN;                               // (no early exit: the flag fetch is a wave operation)
    const size_t col = act ? col0 : EN - V;
    const size_t e = col / N;                                 // (V == 4: N % 4 == 0, the four columns belong to one env)
    const size_t e_first = (size_t)__shfl((long long)e, 0, 64);
    const int de = (int)(e - e_first);
    const bool coop = done != nullptr && __builtin_amdgcn_ballot_w64(de >= 8) == 0ull;
    if (coop) returns_scan<true, V>(reward, done, gamma, G, T, E, EN, col, e, e_first, de, act);
    else returns_scan<false, V>(reward, done, gamma, G, T, E, EN, col, e, e_first, de, act);
}

// K1C = K + 1 at compile time (3 for the reference's k_closest = 2: the neighbour triple is one 12-byte load), 0 = any
// K1 at run time.  A stage of eight steps requests V and the neighbour ids of all eight first, then the G values they
// name (slot 0 is the agent itself: a coalesced row read; the others fall into the same 4 N-byte row), then folds.
template <int K1C>
__global__ void __launch_bounds__(256) advantage_kernel(const float *__restrict__ G, const float *__restrict__ V,
                                                        const int *__restrict__ nbr, const uint8_t *__restrict__ done,
                                                        float gamma, float *__restrict__ w, int T, int E, int N, int K1)
{
    const size_t EN = (size_t)E * N;
    const size_t col0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = col0 < EN;
    const size_t col = act ? col0 : EN - 1;
    const size_t e = col / N;
    const size_t e_first = (size_t)__shfl((long long)e, 0, 64);
    const int de = (int)(e - e_first);
    const bool coop = done != nullptr && __builtin_amdgcn_ballot_w64(de >= 8) == 0ull;
    double disc = 1.0;                                         // gamma^t kept in double: 200 products stay exact to f32
    const float inv_n = 1.0f / (float)N;
    for (int t0 = 0; t0 < T; t0 += kStageT) {
        DoneStage ds;
        if (coop) ds.fetch(done, e_first, E, T, [&](int u) { return t0 + u; });
        if (K1C > 0) {
            float v[kStageT], adv[kStageT];
            int j[kStageT][K1C > 0 ? K1C : 1];
            float g[kStageT][K1C > 0 ? K1C : 1];
#pragma unroll
            for (int u = 0; u < kStageT; ++u) {
                const int t = min(t0 + u, T - 1);
                const size_t o = (size_t)t * EN + col;
                v[u] = __builtin_nontemporal_load(V + o);
#pragma unroll
                for (int q = 0; q < K1C; ++q) j[u][q] = __builtin_nontemporal_load(nbr + o * K1C + q);
            }
#pragma unroll
            for (int u = 0; u < kStageT; ++u) {
                const int t = min(t0 + u, T - 1);
                const float *Grow = G + (size_t)t * EN + e * N;
#pragma unroll
                for (int q = 0; q < K1C; ++q) g[u][q] = Grow[max(j[u][q], 0)];
            }
            // (all 8 K1 gathers are requested above before the first is used, and the "no neighbour" case is an AND mask:
            // written as `j >= 0 ? gq - v : 0` hipcc puts a branch around every gather and waits for each in turn --
            // 24 round trips in series per stage, found in the ISA in round 4)
#pragma unroll
            for (int u = 0; u < kStageT; ++u) {
                float a = 0.0f;
#pragma unroll
                for (int q = 0; q < K1C; ++q)
                    a += __uint_as_float(__float_as_uint(g[u][q] - v[u]) & (j[u][q] >= 0 ? 0xffffffffu : 0u));   // :345-346  (i itself is slot 0)
                adv[u] = a;
            }
#pragma unroll
            for (int u = 0; u < kStageT; ++u) {
                const int t = t0 + u;
                if (t < T) {
                    if (act) __builtin_nontemporal_store(inv_n * (float)disc * adv[u], w + (size_t)t * EN + col);   // :351
                    const bool d = coop ? ds.get(de, u) : (done != nullptr && done[(size_t)t * E + e] != 0);
                    disc = d ? 1.0 : disc * (double)gamma;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < kStageT; ++u) {
                const int t = t0 + u;
                if (t < T) {
                    const size_t o = (size_t)t * EN + col;
                    const float v = V[o];
                    const int *nb = nbr + o * K1;
                    const float *Grow = G + (size_t)t * EN + e * N;
                    float adv = 0.0f;
                    for (int q = 0; q < K1; ++q) {
                        const int jq = nb[q];
                        if (jq >= 0) adv += Grow[jq] - v;      // :345-346  (i itself is slot 0)
                    }
                    if (act) w[o] = inv_n * (float)disc * adv; // :351
                    const bool d = coop ? ds.get(de, u) : (done != nullptr && done[(size_t)t * E + e] != 0);
                    disc = d ? 1.0 : disc * (double)gamma;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The logged statistic of one step (train_problem.py:98-100: sums of rewards, true rewards and collisions),
// accumulated in float64 into acc[5] = (sum r, sum true r, sum collisions, agent-steps, env-steps).
// One launch: every workgroup reduces a slice to three partial sums in `scratch` (fixed LDS tree); the workgroup that
// arrives last reduces the partials with the same fixed tree over the block index: bit-reproducible run to run.
constexpr int kStatBlocks = 256;

__global__ void __launch_bounds__(256) stats_kernel(const float *__restrict__ reward, const float *__restrict__ true_reward,
                                                    const int *__restrict__ n_coll, int E, int N, double *acc,
                                                    double *scratch)
{
    __shared__ double sh[3][256];
    __shared__ bool last;
    const int tid = threadIdx.x;
    const size_t n = (size_t)E * N;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    // 16 bytes per lane when the arrays allow it (E * N a multiple of 4 and 16-byte aligned bases), else 4
    const bool wide = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(reward) | reinterpret_cast<uintptr_t>(true_reward)) & 15u) == 0;
    if (wide) {
        const float4 *r4 = reinterpret_cast<const float4 *>(reward), *t4 = reinterpret_cast<const float4 *>(true_reward);
        for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n / 4; i += (size_t)gridDim.x * 256) {
            const float4 a = r4[i], b = t4[i];
            s0 += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
            s1 += ((double)b.x + (double)b.y) + ((double)b.z + (double)b.w);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) {
            s0 += (double)reward[i];
            s1 += (double)true_reward[i];
        }
    }
    for (int e = blockIdx.x * 256 + tid; e < E; e += gridDim.x * 256) s2 += (double)n_coll[e];
    sh[0][tid] = s0; sh[1][tid] = s1; sh[2][tid] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {                       // fixed tree: the order depends on nothing but (E, N)
        if (tid < w) { sh[0][tid] += sh[0][tid + w]; sh[1][tid] += sh[1][tid + w]; sh[2][tid] += sh[2][tid + w]; }
        __syncthreads();
    }
    unsigned *ticket = reinterpret_cast<unsigned *>(scratch + 3 * kStatBlocks);
    if (tid == 0) {
        for (int k = 0; k < 3; ++k) scratch[k * kStatBlocks + blockIdx.x] = sh[k][0];
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {                                               // the workgroup that arrives last adds the partials,
        __threadfence();                                      // again as a fixed tree over the block index
        for (int k = 0; k < 3; ++k) sh[k][tid] = tid < (int)gridDim.x ? __builtin_nontemporal_load(scratch + k * kStatBlocks + tid) : 0.0;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) { sh[0][tid] += sh[0][tid + w]; sh[1][tid] += sh[1][tid + w]; sh[2][tid] += sh[2][tid + w]; }
            __syncthreads();
        }
        if (tid < 3) acc[tid] += sh[tid][0];
        if (tid == 0) { acc[3] += (double)E * N; acc[4] += (double)E; *ticket = 0u; }
    }
}

// Sum of the E episode records of a rank -> out[8] (include/dronesim.h).  ONE workgroup: thread i adds the records
// i, i + 1024, ... in order, then a fixed LDS tree: the summation order does not depend on anything but E.
__global__ void __launch_bounds__(1024) episode_reduce_kernel(const double *__restrict__ acc, int E, double *__restrict__ out)
{
    __shared__ double sh[8][1024];
    const int tid = threadIdx.x;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < E; e += 1024) {
        const double *rec = acc + 8 * (size_t)e;
        const int *reci = reinterpret_cast<const int *>(rec);
        const long long *recl = reinterpret_cast<const long long *>(rec);
        v[0] += rec[4]; v[1] += rec[5]; v[2] += (double)recl[6]; v[3] += (double)recl[7]; v[4] += (double)reci[6];
        v[5] += rec[0]; v[6] += rec[1]; v[7] += (double)reci[5];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[k][tid] = v[k];
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sh[k][tid] += sh[k][tid + w];
        }
        __syncthreads();
    }
    if (tid < 8) out[tid] = sh[tid][0];
}

// ---------------------------------------------------------------------------------------
thread_local char g_err[256] = "";
long long *g_trace = nullptr;         // developer trace builds (kTrace): per-wave stamp buffer of the step kernel

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
}   // namespace

// shared with the other translation units of the library (csrc/common.hpp); not part of the C ABI
__attribute__((visibility("hidden"))) int dronesim_fail(int code, const char *msg) { return fail(code, msg); }

namespace {


Geometry geometry(int N, int E)
{
    Geometry g;
    if (N <= kWave) {                  // 4 (or 2) independent waves per workgroup, floor(64/N) envs per wave
        g.P = kWave / N;
        // one env per wave (N > 32): two waves per workgroup -- measured against 4 (the round-1 choice), 8 and 16 at C3:
        // 5.33 / 5.40 / 5.42 / 5.71 us per launch (a single wave per workgroup doubles the cost of an empty launch);
        // several small envs per wave keep 4 (C2: 3.85 us against 4.13 with 2)
        const int wpb = g.P == 1 ? 2 : 4;
        g.threads = wpb * kWave;
        g.epb = wpb * g.P;
        g.geo = kPacked;
    } else {
        g.P = 0;
        g.threads = ((N + kWave - 1) / kWave) * kWave;
        g.epb = 1;
        g.geo = g.threads <= 256 ? kBlock256 : kBlock1024;
    }
    g.blocks = (E + g.epb - 1) / g.epb;
    g.lds = 0;
    return g;
}

// dynamic LDS of drone_kernel (must mirror the carve-up in the kernel)
// zc: columns of a staged z row (2, or 5 when those fit); samp: bytes of the in-kernel reset's sampling table when it shares the
// cell tables' region (workgroup-per-env geometries, see drone_lds_tail_bytes)
size_t drone_lds_bytes(const Geometry &g, int N, int k, int zc = 2, bool rollout = false, size_t samp = 0)
{
    const size_t nwaves = (size_t)g.threads / kWave;
    const size_t nconst = g.P > 0 ? nwaves : 1;
    size_t b = sizeof(float2) * ((g.P > 0 ? (size_t)g.epb * 2 * (2 * (size_t)N + kPad) : (size_t)block_pos_entries(N)) +
                                 nconst * ((size_t)N + (N & 1)));
    size_t red = 2 * (size_t)g.epb;
    red += (red & 3) ? 4 - (red & 3) : 0;
    b += sizeof(int) * red;
    b += sizeof(unsigned) * nwaves * kWave * (size_t)(zc + 1) * (size_t)(k + 1);   // z (zc words) + Ni (1 word) per slot and lane
    // bucket filter tables: [2 axes][64 cells][words] per env slot, or kSym64's per-wave rows (whichever is larger)
    const size_t slots = g.P > 0 ? (size_t)g.epb : 1, words = g.P > 0 ? 1 : nwaves;
    size_t generic = (g.P > 0 && N < kBucketMinN) ? 0 : sizeof(unsigned long long) * slots * 2 * (size_t)bucket_cells(g.geo, rollout) * words;
    if (g.P == 0 && samp > generic) generic = samp;
    const size_t sym = g.P > 0 ? sizeof(ulonglong2) * nwaves * kBucketRows : 0;
    b += generic > sym ? generic : sym;
    return b;
}

// entries of the in-kernel reset's sampling table per env slot: the power of two >= 2 N
int samp_table_entries(int N)
{
    int nt = 4;
    while (nt < 2 * N) nt <<= 1;
    return nt;
}

// bookkeeping regions behind the bucket tables: [nwaves rounded to even][2] floats + [epb][table] int2
// Workgroup-per-env geometries, single-step launches (round 5): the sampling table of the in-kernel reset lives in the CELL
// TABLES' region -- a reset runs behind the step's last use of the tables -- so the tail holds the partial sums only: 25.8 ->
// 21.7 KiB per workgroup at N = 256 with the episode layer, 7 instead of 6 workgroups per CU (256 x 4096 envs: 20.0 -> 19.5 us).
// The fused rollouts keep the table in the tail (sharing measured +2 % there: profiles/r5_abtest_sampling_table_alias.log).
size_t drone_samp_bytes(const Geometry &g, int N) { return sizeof(int2) * (size_t)g.epb * samp_table_entries(N); }
size_t drone_lds_tail_bytes(const Geometry &g, int N, bool single_step)
{
    const size_t nwaves = (size_t)g.threads / kWave;
    return sizeof(float) * 2 * ((nwaves + 1) & ~(size_t)1) + ((g.P == 0 && single_step) ? 0 : drone_samp_bytes(g, N));
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
    if (!(p->d_hat_min > 0.0f) || !(p->d_hat_max >= p->d_hat_min))
        return fail(DRONESIM_EINVAL, "need 0 < d_hat_min <= d_hat_max");
    return DRONESIM_OK;
}

}   // namespace

extern "C" {    // drone_kernel_k.hip, one translation unit per k
int dronesim_launch_k1(int, int, const void *, const void *, void *);
int dronesim_launch_k2(int, int, const void *, const void *, void *);
int dronesim_launch_k3(int, int, const void *, const void *, void *);
int dronesim_launch_k4(int, int, const void *, const void *, void *);
int dronesim_launch_k5(int, int, const void *, const void *, void *);
int dronesim_launch_k6(int, int, const void *, const void *, void *);
int dronesim_launch_k7(int, int, const void *, const void *, void *);
int dronesim_launch_k8(int, int, const void *, const void *, void *);
}

namespace {

int reset_impl(const DroneParams *p, int div_x, int div_y, float pitch, uint64_t seed, int64_t env_base,
               const uint8_t *mask, float *pos, float *vel, int32_t *t, int32_t *episode, int32_t *node_out,
               double *acc, int E, void *stream);

// DroneEpisodeCtl -> kernel arguments (include/dronesim.h)
int apply_ctl(const DroneParams *p, const DroneEpisodeCtl *ctl, bool rand_act, KArgs &a)
{
    a.rand_act = rand_act ? 1 : 0;
    if (!ctl) return DRONESIM_OK;
    a.acc = reinterpret_cast<double *>(ctl->acc);
    a.auto_reset = ctl->auto_reset != 0 ? 1 : 0;
    if (a.auto_reset || rand_act) {
        if (!ctl->episode) return fail(DRONESIM_EINVAL, "DroneEpisodeCtl.episode is NULL");
        a.episode = ctl->episode;
        a.key0 = (uint32_t)ctl->seed;
        a.key1 = (uint32_t)(ctl->seed >> 32);
        a.gid_base = (uint32_t)ctl->env_base;
    }
    if (a.auto_reset) {
        if (ctl->div_x < 1 || ctl->div_y < 1) return fail(DRONESIM_EINVAL, "DroneEpisodeCtl: bad lattice size");
        const uint64_t M = (uint64_t)ctl->div_x * (uint64_t)ctl->div_y;
        if (M < (uint64_t)p->N) return fail(DRONESIM_EINVAL, "lattice has fewer nodes than agents (random.sample would raise)");
        if (M > 0xFFFFFFFFull) return fail(DRONESIM_EUNSUPPORTED, "lattice larger than 2^32 nodes");
        a.lat_M = (uint32_t)M; a.div_y = ctl->div_y; a.pitch = ctl->pitch;
        a.z_final = ctl->z_final; a.nbr_final = ctl->nbr_final; a.pos_final = ctl->pos_final;
    }
    return DRONESIM_OK;
}

int launch(int mode, const DroneParams *p, KArgs &a, int E, void *stream)
{
    if (E == 0) return DRONESIM_OK;
    Geometry g = geometry(p->N, E);
    // far agents matter when a z row carries (v, l) of a tie-ordered agent (c = 5) or
    // when a clipped distance can pass a Delta mask (Delta_j >= dhat_i possible)
    const bool far = (p->c == 5) || !(p->delta_max < p->d_hat_min);
    a.uniform = (p->d_hat_min == p->d_hat_max && p->delta_min == p->delta_max && p->radius_min == p->radius_max) ? 1 : 0;
    // one env per wave, uniform (d_hat, Delta, radius) only; its fixed-shape copy-out of c = 2 rows stores 16 bytes per lane.
    // Round 4: also the FAR variant -- the reference's DEFAULT construction (deltas=None, simplify_zstate=False) at N = 64
    if (p->N == 64 && a.uniform && ((reinterpret_cast<uintptr_t>(a.z) | reinterpret_cast<uintptr_t>(a.nbr_idx)) & 15u) == 0)
        g.geo = kSym64;
    // N = 256 with uniform constants (BASELINE configs[4]), fused rollouts: the workgroup-per-env kernel with four full waves known at compile time
    if (mode == kRollout && g.geo == kBlock256 && p->N == 256 && !far && a.uniform &&
        ((reinterpret_cast<uintptr_t>(a.z) | reinterpret_cast<uintptr_t>(a.nbr_idx)) & 15u) == 0)
        g.geo = kBlockU256;
    // the episode layer's regions: only when in use (dronesim_reset_observe: the sampling table of its draw, kept in the tail)
    const bool epi_regions = a.acc != nullptr || a.auto_reset != 0 || a.rand_act != 0 || a.do_reset != 0;
    const size_t tail = epi_regions ? drone_lds_tail_bytes(g, p->N, mode != kRollout && !a.do_reset) : 0;
    // kSym64 has its own carve-up: one block per wave (positions, staging area, cell tables)
    const bool share = g.P == 0 && mode != kRollout;                                    // the sampling table shares the cell tables' region
    const size_t samp = (share && a.auto_reset) ? drone_samp_bytes(g, p->N) : 0;
    g.lds = g.geo == kSym64 ? (size_t)(g.threads / kWave) * sym_wave_bytes(p->k) : drone_lds_bytes(g, p->N, p->k, 2, mode == kRollout, samp);
    a.stage5 = 0; a.lds_vel = 0;
    if (p->c == 5 && g.geo == kSym64) {                 // per-wave blocks with 5-column rows, then the waves' velocities
        a.stage5 = 1;
        a.lds_vel = (int)((size_t)(g.threads / kWave) * sym_wave_bytes(p->k, 5));
        g.lds = (size_t)a.lds_vel + sizeof(float2) * (size_t)g.epb * (size_t)p->N;
    } else if (p->c == 5) {
        // c = 5 rows: staged through LDS like the c = 2 ones, and the agents' velocities kept in LDS for the rows of the k
        // nearest, when the env's tile still fits (it does up to N = 1024 at k <= 5; the rows leave as 4-byte stores at a
        // 60-byte stride otherwise, as they all did through round 2)
        const size_t lds5 = drone_lds_bytes(g, p->N, p->k, 5, mode == kRollout, samp), vel = sizeof(float2) * (size_t)g.epb * (size_t)p->N;
        // (the velocity region is rounded up to 16 bytes: the episode layer reads its per-wave partial sums behind it as
        // ds_read_b128 -- 8 N bytes with odd N left `lds_tail` 8-byte aligned)
        const size_t vel16 = (vel + 15) & ~(size_t)15;
        if (lds5 + vel16 + tail <= 160 * 1024) {
            a.stage5 = 1; a.lds_vel = (int)lds5; g.lds = lds5 + vel16;
        }
    }
    g.lds = (g.lds + 15) & ~(size_t)15;                 // the tail's float4 reads (per-wave partial sums) need 16 bytes
    a.lds_tail = (int)g.lds;
    if (epi_regions) {
        g.lds += tail;
        a.samp_tbl = samp_table_entries(p->N);
        a.samp_shift = 32 - __builtin_ctz((unsigned)a.samp_tbl);
    }
    if (g.lds > 160 * 1024) return fail(DRONESIM_EUNSUPPORTED, "n_agents x k_closest too large for the 160 KiB LDS tile");
    a.N = p->N; a.c = p->c; a.max_steps = p->max_steps; a.E = E;
    a.P = g.P; a.epb = g.epb;
    a.trace = ((kTrace || kTraceSpan) && (mode == kStep || (mode == kObserve && a.do_reset))) ? g_trace : nullptr;
    a.dt = p->dt; a.q = p->q; a.b = p->b; a.done_radius = p->done_radius;
    a.ghost_factor = p->ghost_factor; a.radius_max = p->radius_max;
    a.reach_max = p->d_hat_max + 2.0f * p->radius_max;
    a.dhat_u = p->d_hat_max; a.delta_u = p->delta_max; a.radius_u = p->radius_max;
    a.xF = p->xF; a.xF_lo = p->xF_lo; a.d_hat = p->d_hat; a.delta = p->delta; a.radius = p->radius;
    a.bucket = (g.P > 0 && p->N >= kBucketMinN) ? 1 : 0;
    a.far_inm = !(p->delta_max < p->d_hat_min) ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static_assert(DRONESIM_MAX_K == 8, "eight translation units, one k value each");
    int le = 0;
    switch (p->k) {
    case 1: le = dronesim_launch_k1(mode, far, &a, &g, s); break;
    case 2: le = dronesim_launch_k2(mode, far, &a, &g, s); break;
    case 3: le = dronesim_launch_k3(mode, far, &a, &g, s); break;
    case 4: le = dronesim_launch_k4(mode, far, &a, &g, s); break;
    case 5: le = dronesim_launch_k5(mode, far, &a, &g, s); break;
    case 6: le = dronesim_launch_k6(mode, far, &a, &g, s); break;
    case 7: le = dronesim_launch_k7(mode, far, &a, &g, s); break;
    case 8: le = dronesim_launch_k8(mode, far, &a, &g, s); break;
    default: return fail(DRONESIM_EUNSUPPORTED, "k_closest > DRONESIM_MAX_K");
    }
    if (le != 0) {
        char msg[200];
        snprintf(msg, sizeof(msg), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for %zu bytes of LDS: %s",
                 g.lds, hipGetErrorString(static_cast<hipError_t>(le)));
        return fail(DRONESIM_ELAUNCH, msg);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

}   // namespace

extern "C" {

int dronesim_step_ex(const DroneParams *p, const DroneEpisodeCtl *ctl, float *pos, float *vel, int32_t *t,
                     const float *act, float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                     int32_t *n_coll, uint8_t *done, int E, void *stream)
{
    int rc = check_params(p, E);
    if (rc) return rc;
    if (!pos || !vel || !t || !act || !z || !nbr_idx || !done)
        return fail(DRONESIM_EINVAL, "dronesim_step: required buffer is NULL");
    KArgs a{};
    a.pos = pos; a.vel = vel; a.t = t; a.act = act; a.reward = reward; a.true_reward = true_reward;
    a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll; a.done = done; a.T = 1;
    if ((rc = apply_ctl(p, ctl, false, a)) != 0) return rc;
    return launch(kStep, p, a, E, stream);
}

int dronesim_step_call(const DroneStepCall *c, const float *act, void *stream)
{
    if (!c) return fail(DRONESIM_EINVAL, "dronesim_step_call: call is NULL");
    return dronesim_step_ex(c->p, c->ctl, c->pos, c->vel, c->t, act, c->reward, c->true_reward, c->z, c->nbr_idx,
                            c->n_coll, c->done, c->E, stream);
}

int dronesim_step(const DroneParams *p, float *pos, float *vel, int32_t *t, const float *act,
                  float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                  int32_t *n_coll, uint8_t *done, int E, void *stream)
{
    return dronesim_step_ex(p, nullptr, pos, vel, t, act, reward, true_reward, z, nbr_idx, n_coll, done, E, stream);
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

int dronesim_reset_observe(const DroneParams *p, const DroneEpisodeCtl *ctl, const uint8_t *mask, float *pos, float *vel,
                           int32_t *t, int32_t *node_out, float *z, int32_t *nbr_idx, int E, void *stream)
{
    int rc = check_params(p, E);
    if (rc) return rc;
    if (!ctl || !ctl->episode) return fail(DRONESIM_EINVAL, "dronesim_reset_observe: ctl with the lattice, seed / env_base and episode is required");
    if (!pos || !vel || !t || !z || !nbr_idx) return fail(DRONESIM_EINVAL, "dronesim_reset_observe: required buffer is NULL");
    if (ctl->div_x < 1 || ctl->div_y < 1) return fail(DRONESIM_EINVAL, "bad E / lattice size");
    const uint64_t M = (uint64_t)ctl->div_x * (uint64_t)ctl->div_y;
    if (M < (uint64_t)p->N) return fail(DRONESIM_EINVAL, "lattice has fewer nodes than agents (random.sample would raise)");
    if (M > 0xFFFFFFFFull) return fail(DRONESIM_EUNSUPPORTED, "lattice larger than 2^32 nodes");
    KArgs a{};
    a.pos = pos; a.vel = vel; a.t = t; a.z = z; a.nbr_idx = nbr_idx; a.mask = mask; a.T = 1;
    a.do_reset = 1; a.node_out = node_out;
    a.acc = reinterpret_cast<double *>(ctl->acc);             // retire the episode in progress (as dronesim_reset_ex does)
    a.episode = ctl->episode;
    a.key0 = (uint32_t)ctl->seed; a.key1 = (uint32_t)(ctl->seed >> 32);
    a.gid_base = (uint32_t)ctl->env_base;
    a.lat_M = (uint32_t)M; a.div_y = ctl->div_y; a.pitch = ctl->pitch;
    return launch(kObserve, p, a, E, stream);
}

int dronesim_rollout_ex(const DroneParams *p, const DroneEpisodeCtl *ctl, float *pos, float *vel, int32_t *t,
                        const float *act, float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                        int32_t *n_coll, uint8_t *done, int E, int T, void *stream)
{
    int rc = check_params(p, E);
    if (rc) return rc;
    if (T < 0) return fail(DRONESIM_EINVAL, "T < 0");
    if (!pos || !vel || !t || !act || !z || !nbr_idx || !done)
        return fail(DRONESIM_EINVAL, "dronesim_rollout: required buffer is NULL");
    if (T == 0) return DRONESIM_OK;
    KArgs a{};
    a.pos = pos; a.vel = vel; a.t = t; a.act = act; a.reward = reward; a.true_reward = true_reward;
    a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll; a.done = done; a.T = T;
    a.skin = kSkin * (p->d_hat_max + 2.0f * p->radius_max);
    if ((rc = apply_ctl(p, ctl, false, a)) != 0) return rc;
    return launch(kRollout, p, a, E, stream);
}

int dronesim_rollout(const DroneParams *p, float *pos, float *vel, int32_t *t, const float *act,
                     float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                     int32_t *n_coll, uint8_t *done, int E, int T, void *stream)
{
    return dronesim_rollout_ex(p, nullptr, pos, vel, t, act, reward, true_reward, z, nbr_idx, n_coll, done, E, T, stream);
}

int dronesim_rollout_random(const DroneParams *p, const DroneEpisodeCtl *ctl, float *pos, float *vel, int32_t *t,
                            float *act_out, float *reward, float *true_reward, float *z, int32_t *nbr_idx,
                            int32_t *n_coll, uint8_t *done, int E, int T, void *stream)
{
    int rc = check_params(p, E);
    if (rc) return rc;
    if (T < 0) return fail(DRONESIM_EINVAL, "T < 0");
    if (!ctl || !ctl->episode) return fail(DRONESIM_EINVAL, "dronesim_rollout_random: ctl with seed / env_base / episode is required");
    if (!pos || !vel || !t || !z || !nbr_idx || !done)
        return fail(DRONESIM_EINVAL, "dronesim_rollout_random: required buffer is NULL");
    if (T == 0) return DRONESIM_OK;
    KArgs a{};
    a.pos = pos; a.vel = vel; a.t = t; a.act = nullptr; a.reward = reward; a.true_reward = true_reward;
    a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll; a.done = done; a.T = T;
    a.skin = kSkin * (p->d_hat_max + 2.0f * p->radius_max);
    a.act_out = act_out;
    if ((rc = apply_ctl(p, ctl, true, a)) != 0) return rc;
    return launch(kRollout, p, a, E, stream);
}

int dronesim_reset(const DroneParams *p, int div_x, int div_y, float pitch,
                   uint64_t seed, int64_t env_base, const uint8_t *mask,
                   float *pos, float *vel, int32_t *t, int32_t *episode, int32_t *node_out,
                   int E, void *stream)
{
    return reset_impl(p, div_x, div_y, pitch, seed, env_base, mask, pos, vel, t, episode, node_out, nullptr, E, stream);
}

}   // extern "C"

namespace {
int reset_impl(const DroneParams *p, int div_x, int div_y, float pitch, uint64_t seed, int64_t env_base,
               const uint8_t *mask, float *pos, float *vel, int32_t *t, int32_t *episode, int32_t *node_out,
               double *acc, int E, void *stream)
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
    a.pos = pos; a.vel = vel; a.t = t; a.episode = episode; a.node_out = node_out; a.acc = acc;
    a.nt = samp_table_entries(p->N);
    a.shift = 32 - __builtin_ctz((unsigned)a.nt);
    const size_t lds = sizeof(int2) * (size_t)g.epb * a.nt;
    hipLaunchKernelGGL(reset_kernel, dim3(g.blocks), dim3(g.threads), lds, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}
}   // namespace

extern "C" {

int dronesim_reset_ex(const DroneParams *p, const DroneEpisodeCtl *ctl, const uint8_t *mask,
                      float *pos, float *vel, int32_t *t, int32_t *node_out, int E, void *stream)
{
    if (!ctl) return fail(DRONESIM_EINVAL, "dronesim_reset_ex: ctl is NULL");
    return reset_impl(p, ctl->div_x, ctl->div_y, ctl->pitch, ctl->seed, ctl->env_base, mask, pos, vel, t, ctl->episode,
                      node_out, reinterpret_cast<double *>(ctl->acc), E, stream);
}

int dronesim_episode_reduce(const DroneEpisodeAcc *acc, int E, double *out, void *stream)
{
    if (!acc || !out || E < 0) return fail(DRONESIM_EINVAL, "dronesim_episode_reduce: bad argument");
    hipLaunchKernelGGL(episode_reduce_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const double *>(acc), E, out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

int dronesim_control(const DroneParams *p, int kind, const float *pos, float *act, float u_max,
                     int E, void *stream)
{
    if (!p) return fail(DRONESIM_EINVAL, "params is NULL");
    if (p->N < 1 || p->N > DRONESIM_MAX_AGENTS) return fail(DRONESIM_EUNSUPPORTED, "N must be in 1..1024");
    if (kind != DRONESIM_CONTROL_PROPORTIONAL && kind != DRONESIM_CONTROL_GRADIENT)
        return fail(DRONESIM_EINVAL, "unknown controller kind");
    if (E < 0 || !pos || !act || !p->xF || !p->d_hat || !p->radius)
        return fail(DRONESIM_EINVAL, "dronesim_control: bad E or NULL buffer");
    if (E == 0) return DRONESIM_OK;
    const Geometry g = geometry(p->N, E);
    CArgs a{};
    a.N = p->N; a.E = E; a.P = g.P; a.epb = g.epb; a.kind = kind; a.u_max = u_max;
    a.xF = p->xF; a.xF_lo = p->xF_lo; a.d_hat = p->d_hat; a.radius = p->radius; a.pos = pos; a.act = act;
    const size_t nw = (size_t)g.threads / kWave;
    size_t lds = sizeof(float2) * (size_t)g.epb * p->N + sizeof(float) * (g.P > 0 ? nw : 1) * p->N;
    // gradient: the cell-mask far filter for envs of >= kBucketMinN agents (one env per wave, or one per workgroup)
    // (needs the bounds d_hat_max / radius_max of DroneParams: a caller that left them unset keeps the all-partner loop)
    const float cell_w = (p->d_hat_max + 2.0f * p->radius_max) * 1.001f;
    a.bucket = (kind == DRONESIM_CONTROL_GRADIENT && p->N >= kBucketMinN && g.P <= 1 && p->d_hat_max > 0.0f && p->radius_max >= 0.0f &&
                cell_w > 0.0f && cell_w < 3.0e38f) ? 1 : 0;
    if (a.bucket) {
        lds = (lds + 7) & ~(size_t)7;
        a.tab_off = (int)lds;
        a.inv_cell = 1.0f / cell_w;
        lds += sizeof(unsigned long long) * 2 * kCells * nw;                 // [wave | word][axis][cell]
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g.P > 0) hipLaunchKernelGGL(control_kernel<true>, dim3(g.blocks), dim3(g.threads), lds, s, a);
    else hipLaunchKernelGGL(control_kernel<false>, dim3(g.blocks), dim3(g.threads), lds, s, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

int dronesim_returns(const float *reward, const uint8_t *done, float gamma, float *G,
                     int T, int E, int N, void *stream)
{
    if (!reward || !G || T < 0 || E < 0 || N < 1) return fail(DRONESIM_EINVAL, "dronesim_returns: bad argument");
    if (T == 0 || E == 0) return DRONESIM_OK;
    const size_t cols = (size_t)E * N;
    // 16-byte column quadruples when every row of the [T][E N] arrays starts 16-byte aligned, an env's agents come in fours
    // and the quadruples still fill the chip: below one wave per SIMD (1024 x 64 x 4 columns) half the SIMDs idle -- T = 200 x
    // 512 x 256: 38.6 us with one column per thread, 45.6 with four --, at C3's 262144 columns four per thread win (78.1 ->
    // 69.8 us: 0.75 of the roofline), from a million columns up one per thread is 3 % ahead again (profiles/r5_retbench.log)
    const bool v4 = (N % 4) == 0 && ((reinterpret_cast<uintptr_t>(reward) | reinterpret_cast<uintptr_t>(G)) & 15u) == 0 &&
                    cols >= 262144 && cols < 1048576;
    if (v4)
        hipLaunchKernelGGL(returns_kernel<4>, dim3((unsigned)((cols / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           reward, done, gamma, G, T, E, N);
    else
        hipLaunchKernelGGL(returns_kernel<1>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           reward, done, gamma, G, T, E, N);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

int dronesim_advantage(const float *G, const float *V, const int32_t *nbr_idx, const uint8_t *done,
                       float gamma, float *w, int T, int E, int N, int K1, void *stream)
{
    if (!G || !V || !nbr_idx || !w || T < 0 || E < 0 || N < 1 || K1 < 1)
        return fail(DRONESIM_EINVAL, "dronesim_advantage: bad argument");
    if (T == 0 || E == 0) return DRONESIM_OK;
    const size_t cols = (size_t)E * N;
    if (K1 == 3)
        hipLaunchKernelGGL(advantage_kernel<3>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           G, V, nbr_idx, done, gamma, w, T, E, N, K1);
    else
        hipLaunchKernelGGL(advantage_kernel<0>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           G, V, nbr_idx, done, gamma, w, T, E, N, K1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

int dronesim_episode_stats(const float *reward, const float *true_reward, const int32_t *n_coll, int E, int N,
                           double *acc, double *scratch, void *stream)
{
    if (!reward || !true_reward || !n_coll || !acc || !scratch || E < 0 || N < 1)
        return fail(DRONESIM_EINVAL, "dronesim_episode_stats: bad argument");
    if (E == 0) return DRONESIM_OK;
    hipLaunchKernelGGL(stats_kernel, dim3(kStatBlocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reward, true_reward, n_coll, E, N, acc, scratch);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

// developer hook (tools/trace_*.py with a -DDRONESIM_TRACE build; not declared in include/dronesim.h): without that build
// the buffer is never handed to a kernel
void dronesim_debug_set_trace(long long *buf) { g_trace = buf; }

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
