// verify_f64.hip -- float64 VERIFICATION variant of the env step (test-only; include/dronesim_verify.h:
// dronesim_step_f64 / dronesim_observe_f64; built as libdronesim_verify.so, NOT linked into the product library).  The reference is float64 throughout (drone_env.py:189); the product kernels of dronesim.hip
// compute in float32, which forces two allowances in their parity tests (coordinate differences carry ulp32(G), and
// free-running trajectories drift).  This kernel runs the SAME per-pair arithmetic (`pair_terms<Real>` of common.hpp,
// instantiated for double) and the same epilogue semantics in float64 on the GPU, so that
//   * the reference's golden vectors are met without any float32 allowance (tests: 1e-9 instead of 1e-5 + ulp32(G)),
//   * a free-running 200-step C3 episode tracks the float64 oracle,
//   * the float32 kernels can be compared with a float64 evaluation of the same states ON the device.
// Written for clarity, not speed: one workgroup per env, thread = agent, every ordered pair visited in ascending j
// (the exact general semantics = the FAR path of drone_kernel), no far filter, no staging.
//
// Reference semantics (paths relative to /root/reference/): drone_env.py:214-258 (step), :260-293 (rewards),
// :295-334 (distance_data), :336-401 (localized_states).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdio.h>

#include "dronesim_verify.h"
#include "common.hpp"

// this library's own error string (common.hpp declares the helper; the product library has its own definition)
namespace { thread_local char g_verify_err[256] = ""; }
__attribute__((visibility("hidden"))) int dronesim_fail(int code, const char *msg)
{
    snprintf(g_verify_err, sizeof(g_verify_err), "%s", msg);
    return code;
}

namespace {

struct F64Args {
    int N, c, max_steps, E, step;          // step: 1 = integrate first (drones.step), 0 = observe only
    double dt, q, b, done_radius, ghost_factor;
    const double *xF, *d_hat, *delta, *radius;
    double *pos, *vel;
    int *t;
    const double *act;
    double *reward, *true_reward, *z;
    int *nbr_idx, *n_coll;
    uint8_t *done;
};

// (d, j) ordered lexicographically = stable argsort of row i (drone_env.py:338), for float64 d
struct Key64 {
    unsigned long long o;      // order-preserving image of d
    unsigned j;
};
__device__ __forceinline__ Key64 key64(double d, unsigned j)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    Key64 k;
    k.o = b ^ ((unsigned long long)((long long)b >> 63) | 0x8000000000000000ull);
    k.j = j;
    return k;
}
__device__ __forceinline__ bool key_less(const Key64 &a, const Key64 &b) { return a.o < b.o || (a.o == b.o && a.j < b.j); }

__device__ __forceinline__ double nan_to_num_f64(double x)   // np.nan_to_num, drone_env.py:287-288
{
    if (x != x) return 0.0;
    return ::fmin(::fmax(x, -1.7976931348623157e308), 1.7976931348623157e308);
}

template <int K>
__global__ void __launch_bounds__(1024) drone_kernel_f64(const F64Args a)
{
#pragma clang fp contract(off)             // x + dt*u and dx*dx + dy*dy rounded like the reference's NumPy expressions
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, i = threadIdx.x, env = blockIdx.x;
    double2 *spos = reinterpret_cast<double2 *>(smem);                 // [N] integrated positions
    double2 *sconst = spos + N;                                        // [N] (Delta_j, l_j)
    int *sred = reinterpret_cast<int *>(sconst + N);                   // collisions, "someone outside the goal disk"
    if (i < 2) sred[i] = 0;
    const bool valid = i < N;
    double xi = 0, yi = 0, vxi = 0, vyi = 0, xFx = 0, xFy = 0, dhat = 1, delta_i = 0, li = 0;
    const size_t ga = (size_t)env * N + i;
    if (valid) {
        const double2 p = reinterpret_cast<const double2 *>(a.pos)[ga];
        xi = p.x; yi = p.y;
        if (a.step) {
            const double2 u = reinterpret_cast<const double2 *>(a.act)[ga];
            xi = xi + a.dt * u.x;                                      // drone_env.py:235
            yi = yi + a.dt * u.y;
            vxi = u.x; vyi = u.y;                                      // :238
        } else {
            const double2 v = reinterpret_cast<const double2 *>(a.vel)[ga];
            vxi = v.x; vyi = v.y;
        }
        const double2 g = reinterpret_cast<const double2 *>(a.xF)[i];
        xFx = g.x; xFy = g.y;
        dhat = a.d_hat[i]; delta_i = a.delta[i]; li = a.radius[i];
        spos[i] = make_double2(xi, yi);
        sconst[i] = make_double2(delta_i, li);
    }
    __syncthreads();
    if (valid) {
        Key64 list[K + 1];
#pragma unroll
        for (int s = 0; s <= K; ++s) { list[s].o = ~0ull; list[s].j = ~0u; }
        const double dii = ::fmin(-li - li, dhat);             // :323
        list[0] = key64(dii, (unsigned)i);
        int in_range = ((dii <= delta_i) ? 1 : 0) - 1;                 // :346
        double s_all = 0.0, s_msk = 0.0;
        int ncoll = 0;
        const double log2_dhat = RealOps<double>::log2(dhat);
        for (int j = 0; j < N; ++j) {                                  // ascending j: the summation order of every path
            if (j == i) continue;
            const double2 pj = spos[j];
            const double2 cj = sconst[j];
            const double dx = xi - pj.x, dy = yi - pj.y;
            const PairTerms<double> pt = pair_terms<double>(dx * dx + dy * dy, li, cj.y, dhat, log2_dhat, cj.x);
            s_all += pt.lg;                                            // :283
            s_msk += pt.inm ? pt.lg : 0.0;                             // :282
            ncoll += pt.coll ? 1 : 0;                                  // :284
            in_range += pt.inm ? 1 : 0;
            Key64 key = key64(pt.d, (unsigned)j);                      // :338 (sorted insert, first K+1 kept)
#pragma unroll
            for (int s = 0; s <= K; ++s) {
                const bool lt = key_less(key, list[s]);
                const Key64 cur = list[s];
                list[s] = lt ? key : cur;
                key = lt ? cur : key;
            }
        }
        const double gx = xFx - xi, gy = xFy - yi;
        const double err2 = gx * gx + gy * gy;
        const double to_goal = a.q * err2;                             // :276
        if (a.reward) a.reward[ga] = -nan_to_num_f64(to_goal + a.b * s_msk);          // :282, :287
        if (a.true_reward) a.true_reward[ga] = -nan_to_num_f64(to_goal + a.b * s_all); // :283, :288
        const double zx = xi - xFx, zy = yi - xFy;                     // :357
        const double gsc = delta_i * a.ghost_factor / ::sqrt(err2);            // :386 (NaN on the goal)
        double *zr = a.z + ga * (size_t)((K + 1) * a.c);
        int *nb = a.nbr_idx + ga * (size_t)(K + 1);
#pragma unroll
        for (int kth = 0; kth <= K; ++kth) {
            const unsigned j = list[kth].j;
            const bool have = kth == 0 || j < (unsigned)N;
            const bool real = kth == 0 || (kth <= in_range && have);                   // :362
            double rx = kth == 0 ? zx : zx * gsc, ry = kth == 0 ? zy : zy * gsc;
            if (kth > 0 && real) { rx = spos[j].x - xi; ry = spos[j].y - yi; }         // :368
            nb[kth] = kth == 0 ? i : (real ? (int)j : -1);
            double *row = zr + kth * a.c;
            row[0] = rx; row[1] = ry;
            if (a.c == 5) {
                if (kth == 0) { row[2] = vxi; row[3] = vyi; row[4] = li; }             // :355
                else if (have) {                                                       // :367 / :385
                    const double2 vj = reinterpret_cast<const double2 *>(a.step ? a.act : a.vel)[(size_t)env * N + j];
                    row[2] = vj.x; row[3] = vj.y; row[4] = sconst[j].y;
                } else { row[2] = row[3] = row[4] = __builtin_nan(""); }
            }
        }
        if (a.step) {
            reinterpret_cast<double2 *>(a.pos)[ga] = make_double2(xi, yi);
            reinterpret_cast<double2 *>(a.vel)[ga] = make_double2(vxi, vyi);
            if (!(::sqrt(err2) <= a.done_radius)) atomicOr(&sred[1], 1);       // :249-251
        }
        if (ncoll) atomicAdd(&sred[0], ncoll);
    }
    __syncthreads();
    if (i == 0) {
        if (a.n_coll) a.n_coll[env] = sred[0];
        if (a.step) {
            const int tcur = a.t[env];
            a.done[env] = (uint8_t)((sred[1] == 0) || (tcur >= a.max_steps - 1));      // :251
            a.t[env] = tcur + 1;                                                       // :256
        }
    }
}

template <int K>
void launch_k(const F64Args &a, int threads, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL(drone_kernel_f64<K>, dim3(a.E), dim3(threads), lds, s, a);
}

int run(const DroneParamsF64 *p, F64Args &a, int E, void *stream)
{
    if (!p) return dronesim_fail(DRONESIM_EINVAL, "params is NULL");
    if (E < 0) return dronesim_fail(DRONESIM_EINVAL, "E < 0");
    if (p->N < 2 || p->N > DRONESIM_MAX_AGENTS) return dronesim_fail(DRONESIM_EUNSUPPORTED, "N must be in 2..1024");
    if (p->k < 1 || p->k > p->N - 1) return dronesim_fail(DRONESIM_EINVAL, "k_closest must be in 1..N-1");
    if (p->k > DRONESIM_MAX_K) return dronesim_fail(DRONESIM_EUNSUPPORTED, "k_closest > DRONESIM_MAX_K");
    if (p->c != 2 && p->c != 5) return dronesim_fail(DRONESIM_EINVAL, "c must be 2 or 5");
    if (!p->xF || !p->d_hat || !p->delta || !p->radius) return dronesim_fail(DRONESIM_EINVAL, "constant array is NULL");
    if (E == 0) return DRONESIM_OK;
    a.N = p->N; a.c = p->c; a.max_steps = p->max_steps; a.E = E;
    a.dt = p->dt; a.q = p->q; a.b = p->b; a.done_radius = p->done_radius; a.ghost_factor = p->ghost_factor;
    a.xF = p->xF; a.d_hat = p->d_hat; a.delta = p->delta; a.radius = p->radius;
    const int threads = ((p->N + 63) / 64) * 64;
    const size_t lds = sizeof(double2) * 2 * (size_t)p->N + 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (p->k) {
    case 1: launch_k<1>(a, threads, lds, s); break;
    case 2: launch_k<2>(a, threads, lds, s); break;
    case 3: launch_k<3>(a, threads, lds, s); break;
    case 4: launch_k<4>(a, threads, lds, s); break;
    case 5: launch_k<5>(a, threads, lds, s); break;
    case 6: launch_k<6>(a, threads, lds, s); break;
    case 7: launch_k<7>(a, threads, lds, s); break;
    default: launch_k<8>(a, threads, lds, s); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dronesim_fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

}   // namespace

extern "C" {

int dronesim_step_f64(const DroneParamsF64 *p, double *pos, double *vel, int32_t *t, const double *act,
                      double *reward, double *true_reward, double *z, int32_t *nbr_idx,
                      int32_t *n_coll, uint8_t *done, int E, void *stream)
{
    if (!pos || !vel || !t || !act || !z || !nbr_idx || !done)
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_step_f64: required buffer is NULL");
    F64Args a{};
    a.step = 1; a.pos = pos; a.vel = vel; a.t = t; a.act = act; a.reward = reward; a.true_reward = true_reward;
    a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll; a.done = done;
    return run(p, a, E, stream);
}

int dronesim_observe_f64(const DroneParamsF64 *p, const double *pos, const double *vel,
                         double *reward, double *true_reward, double *z, int32_t *nbr_idx,
                         int32_t *n_coll, int E, void *stream)
{
    if (!pos || !vel || !z || !nbr_idx) return dronesim_fail(DRONESIM_EINVAL, "dronesim_observe_f64: required buffer is NULL");
    F64Args a{};
    a.step = 0; a.pos = const_cast<double *>(pos); a.vel = const_cast<double *>(vel);
    a.reward = reward; a.true_reward = true_reward; a.z = z; a.nbr_idx = nbr_idx; a.n_coll = n_coll;
    return run(p, a, E, stream);
}

const char *dronesim_verify_last_error(void) { return g_verify_err; }

}   // extern "C"
