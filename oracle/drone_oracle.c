/*
 * drone_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).
 *
 * A plain-C, float64 restatement of the step()/reset() hot path of the
 * reference environment /root/reference/drone_env.py (class `drones`),
 * batched over E independent environment instances.  It exists so that the
 * HIP kernels in scalable_collision_avoidance_rl_amd/csrc/ can be checked on
 * a machine where the (Python) reference itself is not available.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library.  The product path never calls it and has no CPU
 * fallback.
 *
 * PARITY PIN: the reference ships no tests or golden vectors of its own
 * (SURVEY.md section 4), so this restatement is pinned against outputs of the
 * reference itself: the .npz files under tests/golden/, generated in the build container by
 * tests/golden/gen_golden.py (which imports the unmodified reference), and
 * checked by tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/).
 *
 * Layouts (all row-major, env-major):
 *   pos[E][N][2], vel[E][N][2], act[E][N][2]      double
 *   reward[E][N], true_reward[E][N]               double
 *   z[E][N][k+1][c]   (c = 2 simplified, 5 full)  double
 *   nbr_idx[E][N][k+1]  int32, slot 0 = i, unused slots = -1
 *   n_coll[E] int32 (ordered pairs, always even), done[E] uint8, t[E] int32
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t N;            /* agents per env                                */
    int32_t k;            /* k_closest                      drone_env.py:69 */
    int32_t c;            /* 2 (simplify_zstate) or 5       drone_env.py:390-395 */
    int32_t max_steps;    /* max_time_steps = 200           drone_env.py:30 */
    double dt;            /* 0.05                           drone_env.py:29 */
    double q;             /* 2*dt                           drone_env.py:269 */
    double b;             /* collision_weight*dt            drone_env.py:270 */
    double done_radius;   /* 0.2                            drone_env.py:251 */
    double ghost_factor;  /* 1.1                            drone_env.py:386 */
    const double *xF;     /* [N][2] goal ring               drone_env.py:124-131 */
    const double *d_hat;  /* [N] safety distance            drone_env.py:137-153 */
    const double *delta;  /* [N] sensing radius (clipped)   drone_env.py:85-89 */
    const double *radius; /* [N] drone radius l             drone_env.py:75 */
} OracleParams;

static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int oracle_get_threads(void) { return g_threads; }

/* ---------------------------------------------------------------------- */
/* generate_formation("O") + d_safety       drone_env.py:115-153           */
/* returns 0 ok, -1 unknown formation                                      */
int oracle_formation(int N, double gx, double gy, const double *radius,
                     double *xF /*[N][2]*/, double *d_hat /*[N]*/)
{
    const double angle_step = 2.0 * M_PI / (double)N;          /* :126 */
    for (int i = 0; i < N; ++i) {
        xF[2 * i + 0] = cos(i * angle_step) * 0.9 * gx / 2 + gx / 2;   /* :130 */
        xF[2 * i + 1] = sin(i * angle_step) * 0.9 * gy / 2 + gy / 2;   /* :131 */
    }
    for (int i = 0; i < N; ++i) {                               /* :139-150 */
        double best = INFINITY;
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            const double dx = xF[2 * i] - xF[2 * j], dy = xF[2 * i + 1] - xF[2 * j + 1];
            const double d = sqrt(dx * dx + dy * dy) - radius[i] - radius[j];
            if (d < best) best = d;
        }
        d_hat[i] = floor(best * 100.0) / 100.0;                 /* :153 */
    }
    return 0;
}

/* ---------------------------------------------------------------------- */
/* distance_data for ONE env                 drone_env.py:295-334          */
/* d[N][N], logd[N][N]; ndelta/coll [N][N] uint8                           */
void oracle_distance_data(const OracleParams *p, const double *pos,
                          double *d, double *logd, uint8_t *ndelta, uint8_t *coll)
{
    const int N = p->N;
    for (int i = 0; i < N; ++i) {
        const double xi = pos[2 * i], yi = pos[2 * i + 1], li = p->radius[i];
        for (int j = 0; j < N; ++j) {
            double dij, ratio;
            if (j != i) {
                const double dx = xi - pos[2 * j], dy = yi - pos[2 * j + 1];
                dij = fmin(sqrt(dx * dx + dy * dy) - li - p->radius[j], p->d_hat[i]); /* :318 */
                if (dij == 0.0) dij = -1e-6;                                          /* :319-320 */
                ratio = p->d_hat[i] / dij;                                            /* :321 */
            } else {
                dij = fmin(-li - li, p->d_hat[i]);                                    /* :323 */
                ratio = 1.0;                                                          /* :325 */
            }
            const int is_coll = ratio <= 0.0;                                         /* :327 */
            d[i * N + j] = dij;
            coll[i * N + j] = (uint8_t)is_coll;
            ndelta[i * N + j] = (uint8_t)(dij <= p->delta[j]);   /* :328, broadcast over columns */
            logd[i * N + j] = is_coll ? 9.99e3 : log(ratio);     /* :330-332 */
        }
    }
}

static double nan_to_num(double x)   /* np.nan_to_num defaults, drone_env.py:287-288 */
{
    if (isnan(x)) return 0.0;
    if (isinf(x)) return x > 0 ? DBL_MAX : -DBL_MAX;
    return x;
}

/* rewards() + localized_states() for ONE env     drone_env.py:260-293, 336-401 */
static void observe_one(const OracleParams *p, const double *pos, const double *vel,
                        double *reward, double *true_reward, double *z,
                        int32_t *nbr_idx, int32_t *n_coll,
                        double *d, double *logd, uint8_t *ndelta, uint8_t *coll,
                        double *sd, int32_t *sj)
{
    const int N = p->N, k = p->k, c = p->c, K1 = k + 1;
    oracle_distance_data(p, pos, d, logd, ndelta, coll);                 /* :280 */

    int32_t ncoll = 0;
    for (int i = 0; i < N; ++i) {
        const double gx = p->xF[2 * i] - pos[2 * i], gy = p->xF[2 * i + 1] - pos[2 * i + 1];
        const double nrm = sqrt(gx * gx + gy * gy);
        const double to_goal = p->q * (nrm * nrm);                        /* :276 */
        double s_masked = 0.0, s_all = 0.0;
        int in_range = -1;                                                /* :346, minus itself */
        for (int j = 0; j < N; ++j) {
            s_masked += logd[i * N + j] * (double)ndelta[i * N + j];      /* :282 */
            s_all += logd[i * N + j];                                     /* :283 */
            ncoll += coll[i * N + j];                                     /* :284 */
            in_range += ndelta[i * N + j];
        }
        reward[i] = -nan_to_num(to_goal + p->b * s_masked);               /* :287 */
        true_reward[i] = -nan_to_num(to_goal + p->b * s_all);             /* :288 */

        /* first k+1 entries of a STABLE argsort of row i (ties -> lowest index).
           The reference's np.argsort (:338) is unstable; its tie order is
           implementation defined, so golden vectors avoid tied selections. */
        int m = 0;
        for (int j = 0; j < N; ++j) {
            const double dv = d[i * N + j];
            int pos_ins = m;
            while (pos_ins > 0 && dv < sd[pos_ins - 1]) --pos_ins;
            if (pos_ins >= K1) continue;
            const int last = m < K1 ? m : K1 - 1;
            for (int s = last; s > pos_ins; --s) { sd[s] = sd[s - 1]; sj[s] = sj[s - 1]; }
            sd[pos_ins] = dv; sj[pos_ins] = j;
            if (m < K1) ++m;
        }

        double *Zi = z + (size_t)i * K1 * c;
        int32_t *Ni = nbr_idx + (size_t)i * K1;
        const double zx = -(p->xF[2 * i] - pos[2 * i]);                   /* :357 */
        const double zy = -(p->xF[2 * i + 1] - pos[2 * i + 1]);
        Zi[0] = zx; Zi[1] = zy;
        if (c == 5) { Zi[2] = vel[2 * i]; Zi[3] = vel[2 * i + 1]; Zi[4] = p->radius[i]; } /* :355 */
        Ni[0] = i;                                                        /* :348 */
        for (int kth = 1; kth <= k; ++kth) {
            double *row = Zi + kth * c;
            const int j = kth < m ? sj[kth] : -1;    /* reference would raise IndexError if k >= N */
            if (kth <= in_range && j >= 0) {                              /* :362-368 */
                Ni[kth] = j;
                row[0] = pos[2 * j] - pos[2 * i];
                row[1] = pos[2 * j + 1] - pos[2 * i + 1];
            } else {                                                      /* :383-386 ghost */
                Ni[kth] = -1;
                const double zn = sqrt(zx * zx + zy * zy);
                row[0] = zx / zn * p->delta[i] * p->ghost_factor;
                row[1] = zy / zn * p->delta[i] * p->ghost_factor;
            }
            if (c == 5) {
                if (j >= 0) { row[2] = vel[2 * j]; row[3] = vel[2 * j + 1]; row[4] = p->radius[j]; }
                else { row[2] = row[3] = row[4] = NAN; }
            }
        }
    }
    *n_coll = ncoll;
}

typedef struct {
    double *d, *logd, *sd; uint8_t *ndelta, *coll; int32_t *sj;
} Scratch;

static int scratch_alloc(Scratch *s, int N, int K1)
{
    s->d = (double *)malloc(sizeof(double) * N * N);
    s->logd = (double *)malloc(sizeof(double) * N * N);
    s->ndelta = (uint8_t *)malloc((size_t)N * N);
    s->coll = (uint8_t *)malloc((size_t)N * N);
    s->sd = (double *)malloc(sizeof(double) * (K1 + 1));
    s->sj = (int32_t *)malloc(sizeof(int32_t) * (K1 + 1));
    return (s->d && s->logd && s->ndelta && s->coll && s->sd && s->sj) ? 0 : -1;
}
static void scratch_free(Scratch *s)
{
    free(s->d); free(s->logd); free(s->ndelta); free(s->coll); free(s->sd); free(s->sj);
}

/* rewards()+localized_states() over E envs, no integration (what init_agents runs, :208) */
int oracle_observe(const OracleParams *p, const double *pos, const double *vel,
                   double *reward, double *true_reward, double *z, int32_t *nbr_idx,
                   int32_t *n_coll, int E)
{
    const int N = p->N, K1 = p->k + 1, c = p->c;
    int err = 0;
#pragma omp parallel num_threads(g_threads)
    {
        Scratch s;
        if (scratch_alloc(&s, N, K1) != 0) {
#pragma omp atomic write
            err = -1;
        } else {
#pragma omp for schedule(static)
            for (int e = 0; e < E; ++e)
                observe_one(p, pos + (size_t)e * N * 2, vel + (size_t)e * N * 2,
                            reward + (size_t)e * N, true_reward + (size_t)e * N,
                            z + (size_t)e * N * K1 * c, nbr_idx + (size_t)e * N * K1,
                            n_coll + e, s.d, s.logd, s.ndelta, s.coll, s.sd, s.sj);
        }
        scratch_free(&s);
    }
    return err;
}

/* step() over E envs                         drone_env.py:214-258 */
int oracle_step(const OracleParams *p, double *pos, double *vel, int32_t *t, const double *act,
                double *reward, double *true_reward, double *z, int32_t *nbr_idx,
                int32_t *n_coll, uint8_t *done, int E)
{
    const int N = p->N;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int e = 0; e < E; ++e) {
        double *ps = pos + (size_t)e * N * 2, *vs = vel + (size_t)e * N * 2;
        const double *as = act + (size_t)e * N * 2;
        for (int i = 0; i < 2 * N; ++i) {
            ps[i] = ps[i] + p->dt * as[i];          /* :235, A = I, B = dt*I (:78-79) */
            vs[i] = as[i];                          /* :238 */
        }
    }
    const int rc = oracle_observe(p, pos, vel, reward, true_reward, z, nbr_idx, n_coll, E); /* :242 */
    if (rc) return rc;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int e = 0; e < E; ++e) {
        const double *ps = pos + (size_t)e * N * 2;
        int all_in = 1;
        for (int i = 0; i < N; ++i) {                                      /* :248-249 */
            const double ex = p->xF[2 * i] - ps[2 * i], ey = p->xF[2 * i + 1] - ps[2 * i + 1];
            if (!(sqrt(ex * ex + ey * ey) <= p->done_radius)) all_in = 0;
        }
        done[e] = (uint8_t)(all_in || t[e] >= p->max_steps - 1);          /* :251 */
        t[e] += 1;                                                         /* :256 */
    }
    return 0;
}

/* ---------------------------------------------------------------------- */
/* Smallest distance of any discrete decision of observe/step from its     */
/* threshold, per env: collision (d_ij vs 0), Delta mask (d_ij vs          */
/* Delta_j), neighbour ranking (gaps inside the first k+2 sorted           */
/* entries of each row, ties between clipped entries excluded because      */
/* both sides break them by index), done (|err - done_radius|).            */
/* Tests use it to compare discrete outputs only where fp32 and fp64       */
/* cannot legitimately disagree.                         */
int oracle_margins(const OracleParams *p, const double *pos, double *margin, int E)
{
    const int N = p->N, K2 = p->k + 2;
    double *d = (double *)malloc(sizeof(double) * N * N);
    double *logd = (double *)malloc(sizeof(double) * N * N);
    uint8_t *nd = (uint8_t *)malloc((size_t)N * N), *cl = (uint8_t *)malloc((size_t)N * N);
    double *row = (double *)malloc(sizeof(double) * N);
    if (!d || !logd || !nd || !cl || !row) return -1;
    for (int e = 0; e < E; ++e) {
        const double *ps = pos + (size_t)e * N * 2;
        oracle_distance_data(p, ps, d, logd, nd, cl);
        double m = INFINITY;
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;
                const double v = d[i * N + j];
                m = fmin(m, fabs(v));
                m = fmin(m, fabs(v - p->delta[j]));
                if (v < p->d_hat[i]) m = fmin(m, p->d_hat[i] - v);  /* near vs clipped boundary */
            }
            /* partial selection sort of the first K2 entries */
            memcpy(row, d + (size_t)i * N, sizeof(double) * N);
            const int lim = K2 < N ? K2 : N;
            for (int a = 0; a < lim; ++a) {
                int best = a;
                for (int b2 = a + 1; b2 < N; ++b2) if (row[b2] < row[best]) best = b2;
                const double tmp = row[a]; row[a] = row[best]; row[best] = tmp;
                if (a > 0) {
                    const int both_clipped = (row[a] >= p->d_hat[i]) && (row[a - 1] >= p->d_hat[i]);
                    if (!both_clipped) m = fmin(m, row[a] - row[a - 1]);
                }
            }
            const double ex = p->xF[2 * i] - ps[2 * i], ey = p->xF[2 * i + 1] - ps[2 * i + 1];
            m = fmin(m, fabs(sqrt(ex * ex + ey * ey) - p->done_radius));
        }
        margin[e] = m;
    }
    free(d); free(logd); free(nd); free(cl); free(row);
    return 0;
}

/* ---------------------------------------------------------------------- */
/* reset / init_agents                       drone_env.py:98-102, 171-212  */
/*                                                                         */
/* The reference draws N distinct lattice nodes (pitch 2*1.1*l = 0.22,     */
/* floor(G/0.22) nodes per axis, node (idx,jdx) -> (idx*0.22, jdx*0.22))   */
/* with Python's random.sample (:204).  Its Mersenne-Twister stream is not */
/* reproduced; the build's reset uses the counter-based Philox4x32-10      */
/* generator and the parallel rejection scheme restated below, and this    */
/* function is the integer-exact CPU statement of that scheme (node ids    */
/* must match the HIP kernel bit for bit).                                 */

static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint32_t oracle_philox_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                             uint32_t k0, uint32_t k1)
{
    uint32_t o[4];
    philox4x32_10(c0, c1, c2, c3, k0, k1, o);
    return o[0];
}

void oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
    philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

/* RandomAgent.forward: np.clip(-1 + 2*np.random.rand(n_actions), -1, 1)          SAC_agents.py:9-22
 * NumPy's stream is not reproduced; the build draws the same distribution from the
 * counter-based stream documented at dronesim_rollout_random (include/dronesim.h):
 *   philox4x32-10(ctr = (agent, env_base + e, t[e] >> 1, episode[e]); key = (seed.lo ^ "RAND", seed.hi)),
 *   words 2 (t & 1), 2 (t & 1) + 1;  a = -1 + (w >> 8) * 2^-23  -- exact in float32 and float64 alike.
 * act[E][N][2] for the CURRENT t / episode of every env.                                              */
void oracle_rand_actions(int N, int E, uint64_t seed, int64_t env_base, const int32_t *t,
                         const int32_t *episode, double *act)
{
    const uint32_t k0 = (uint32_t)seed ^ 0x52414E44u, k1 = (uint32_t)(seed >> 32);
    for (int e = 0; e < E; ++e)
        for (int i = 0; i < N; ++i) {
            uint32_t o[4];
            philox4x32_10((uint32_t)i, (uint32_t)(env_base + e), (uint32_t)t[e] >> 1, (uint32_t)episode[e], k0, k1, o);
            const int h = (t[e] & 1) ? 2 : 0;
            act[((size_t)e * N + i) * 2 + 0] = -1.0 + (double)(o[h] >> 8) * 0x1p-23;
            act[((size_t)e * N + i) * 2 + 1] = -1.0 + (double)(o[h + 1] >> 8) * 0x1p-23;
        }
}

/* Draw N distinct nodes out of M = div_x*div_y for env `env_gid`:
 *   round r: every unsettled agent i proposes node = mulhi32(philox(ctr = (i, r, env_gid,
 *            episode[e]); key = (seed_lo, seed_hi)).word0, M), episode[e] = number of resets
 *            env e has seen so far (incremented here);
 *   i settles iff no settled agent holds that node and no unsettled agent
 *   with a smaller index proposed it in this round.
 * Positions = (idx*pitch, jdx*pitch), node = idx*div_y + jdx (:197-200).
 * mask NULL = reset every env; otherwise only envs with mask[e] != 0.
 * returns 0, or -2 if M < N, -3 if the round cap is hit. */
int oracle_reset(int N, int div_x, int div_y, double pitch, uint64_t seed,
                 int64_t env_base, const uint8_t *mask,
                 double *pos, double *vel, int32_t *t, int32_t *episode, int32_t *node_out, int E)
{
    const uint64_t M = (uint64_t)div_x * (uint64_t)div_y;
    if (M < (uint64_t)N || M > 0xFFFFFFFFull) return -2;
    int32_t *node = (int32_t *)malloc(sizeof(int32_t) * N);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * N);
    uint8_t *win = (uint8_t *)malloc((size_t)N);
    if (!node || !cand || !win) return -1;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    int rc = 0;
    for (int e = 0; e < E && rc == 0; ++e) {
        if (mask && !mask[e]) continue;
        const uint32_t gid = (uint32_t)(env_base + e);
        for (int i = 0; i < N; ++i) node[i] = -1;
        int remaining = N;
        for (uint32_t r = 0; remaining > 0; ++r) {
            if (r >= (1u << 20)) { rc = -3; break; }
            for (int i = 0; i < N; ++i) {
                if (node[i] >= 0) continue;
                const uint32_t w = oracle_philox_word0((uint32_t)i, r, gid, (uint32_t)episode[e], k0, k1);
                cand[i] = (int32_t)(((uint64_t)w * M) >> 32);
            }
            /* decide against the state at the START of the round (as the parallel kernel
               does): proposals of agents that lose this round still block higher indices */
            for (int i = 0; i < N; ++i) win[i] = 0;
            for (int i = 0; i < N; ++i) {
                if (node[i] >= 0) continue;
                int ok = 1;
                for (int j = 0; j < N && ok; ++j) {
                    if (j == i) continue;
                    if (node[j] >= 0) { if (node[j] == cand[i]) ok = 0; }   /* settled earlier */
                    else if (j < i && cand[j] == cand[i]) ok = 0;            /* lower index wins */
                }
                win[i] = (uint8_t)ok;
            }
            for (int i = 0; i < N; ++i) {
                if (node[i] < 0 && win[i]) { node[i] = cand[i]; --remaining; }
            }
        }
        if (rc) break;
        for (int i = 0; i < N; ++i) {
            const int idx = node[i] / div_y, jdx = node[i] % div_y;
            pos[((size_t)e * N + i) * 2 + 0] = idx * pitch;
            pos[((size_t)e * N + i) * 2 + 1] = jdx * pitch;
            vel[((size_t)e * N + i) * 2 + 0] = 0.0;                      /* :189 */
            vel[((size_t)e * N + i) * 2 + 1] = 0.0;
            if (node_out) node_out[(size_t)e * N + i] = node[i];
        }
        t[e] = 0;                                                         /* :100 */
        episode[e] += 1;
    }
    free(node); free(cand); free(win);
    return rc;
}

/* ---------------------------------------------------------------------- */
/* Classical controllers     drone_env.py:609-679         */

/* gradient_control(state, env, u_max): log-barrier gradient over the complete graph.
 * act[E][N][2].  b = 0.1, q = 1 as in the reference (:623-624).           drone_env.py:609-650 */
int oracle_gradient_control(const OracleParams *p, const double *pos, double u_max, double *act, int E)
{
    const int N = p->N;
    const double b = 0.1, q = 1.0;
    for (int e = 0; e < E; ++e) {
        const double *ps = pos + (size_t)e * N * 2;
        for (int i = 0; i < N; ++i) {
            const double xi = ps[2 * i], yi = ps[2 * i + 1];
            const double t1x = 2 * (xi - p->xF[2 * i]), t1y = 2 * (yi - p->xF[2 * i + 1]);   /* :633 */
            double t2x = 0.0, t2y = 0.0;
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;
                const double dx = xi - ps[2 * j], dy = yi - ps[2 * j + 1];
                const double nrm = sqrt(dx * dx + dy * dy);
                const double dij = nrm - p->radius[i] - p->radius[j];                        /* :641 */
                if (dij <= p->d_hat[i]) { t2x += dx / (dij * nrm); t2y += dy / (dij * nrm); } /* :643-644 */
            }
            const double gx = q * t1x - b * t2x, gy = q * t1y - b * t2y;                     /* :646 */
            act[((size_t)e * N + i) * 2 + 0] = fmin(fmax(-gx, -u_max), u_max);               /* :647 */
            act[((size_t)e * N + i) * 2 + 1] = fmin(fmax(-gy, -u_max), u_max);
        }
    }
    return 0;
}

/* proportional_control(state, env): saturated P-controller, u_max = 1, k_gain = 1.   drone_env.py:652-679 */
int oracle_proportional_control(const OracleParams *p, const double *pos, double *act, int E)
{
    const int N = p->N;
    const double u_max = 1.0, k_gain = 1.0;
    for (int e = 0; e < E; ++e)
        for (int i = 0; i < N; ++i) {
            const double *x = pos + ((size_t)e * N + i) * 2;
            double ux = k_gain * (p->xF[2 * i] - x[0]), uy = k_gain * (p->xF[2 * i + 1] - x[1]);   /* :667-668 */
            const double nrm = sqrt(ux * ux + uy * uy);
            if (nrm > u_max) { ux = ux / nrm * u_max; uy = uy / nrm * u_max; }                     /* :670-673 */
            act[((size_t)e * N + i) * 2 + 0] = ux;
            act[((size_t)e * N + i) * 2 + 1] = uy;
        }
    return 0;
}
