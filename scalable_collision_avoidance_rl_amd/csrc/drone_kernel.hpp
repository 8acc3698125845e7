// drone_kernel.hpp -- the step / observe / rollout kernel of the batched drone_env hot path (gfx950) and its launchers.
// Included by drone_kernel_k.hip (one translation unit per k_closest value: the instantiations of that k) and by
// dronesim.hip (the C ABI, which only needs KArgs / Geometry and the LDS carve-up constants).
//
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
// Work decomposition (one launch = one env.step() for E envs); DESIGN.md section 3 has the measurements
//   N <= 64 : lane = agent; floor(64/N) envs packed per wave, 4 independent waves per 256-thread
//             workgroup (2 per 128-thread workgroup when a wave holds one env), wave-local synchronisation only.  N = 64 -> one wave per env.
//   N  > 64 : one workgroup per env, thread = agent (N <= 1024).
//   Every wave covers a contiguous range of global agents: streams are wave-uniform base + lane.
//   LDS tile: the integrated positions of an env, read by agent index.
//   Pass 1 ("far filter"): a pair beyond the early-out radius reach_i = dhat_i + l_i + l_max has
//   d_ij = dhat_i, log term 0, no collision, and -- when max(Delta) < min(dhat), the regime of every
//   config in BASELINE.json -- is outside every Delta mask, so it contributes nothing.  Instead of testing
//   all pairs, each env hashes its agents into 64 cells per axis (cell width >= max reach) and keeps, per
//   cell, the mask of the agents in it (LDS, built with ds_or_b64): an agent's candidates are
//   (masks of its x cell +-1) & (masks of its y cell +-1); those few get the exact squared-distance
//   test.  Survivors become one bit per partner.  Crowded waves (an agent with > 10 candidates) test all
//   partners instead (N = 64: every unordered pair once, verdict handed over as a rotated ballot);
//   small packed envs (N < 40) and the FAR variant scan partners through relative windows of a doubled
//   position array (lane i reads partner (i + r) mod N at the wrap-free address base_i + r, two copies
//   one element apart so that every lane has a 16-byte aligned window).
//   Pass 2 ("near pairs"): each lane walks ITS OWN set bits, so a wave spends
//   max-over-lanes(popcount) iterations instead of one per partner; only here are
//   sqrt / log / the Delta mask / the (k+1)-entry sorted neighbour list evaluated.
//   When far agents can matter (z rows carry v,l of tie-ordered agents for c = 5, or
//   Delta >= dhat), the FAR variant sends every pair through pass 2 (exact general
//   semantics, slower).
//   Ordering: the neighbour list is ordered by (d_ij, j) lexicographically = the
//   first k+1 entries of a stable argsort of row i, independent of visiting order.
//   Outputs: z rows / Ni are transposed through LDS and leave as full 128-byte lines; all outputs
//   use streaming (non-temporal) stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>

#include "dronesim.h"
#include "common.hpp"

namespace {

constexpr int kWave = 64;
constexpr int kChunk = 16;         // partners whose LDS reads are in flight together (pass 1)
constexpr int kPad = kChunk + 4;   // LDS slack so the last chunk may over-read (+ the shifted copy's offset)
constexpr int kCells = 64;          // cells per axis of the bucket filter (coordinates are hashed: cell & 63)
// workgroup-per-env geometries (N > 64), single-step launches: their grids are wide -- G = 256 at BASELINE configs[4], 91
// cells of one reach per axis -- and hashed into 64 cells a third of the cells alias on either axis, which triples the
// candidates the exact test has to reject (~0.9 instead of 0.28 per agent at the C5 shard).  128 cells hold grids up to
// 360 reaches without aliasing: C5 shard 5.10 -> 4.97 us per step, with the episode layer 5.70 -> 5.55; 256 cells lose
// it again to the zeroing of the tables (round 5, profiles/r5_abtest_block_cells.log)
constexpr int kCellsBlock = 128;
constexpr int kBucketRows = 66;     // cells 0..63 plus one empty guard row at either end
constexpr int kBucketMinN = 40;     // packed envs smaller than this scan all partners (the tables would cost more)
constexpr float kSkin = 0.4f;       // fused rollout: slack of the register-resident candidate list, in units of the reach
constexpr int kBucketMax = 10;      // candidates per agent beyond which the all-pairs scan is cheaper
constexpr int kFarAllMaxN = 8;      // FAR variant: envs this small send every pair through pass 2 (no filter, no far tail)
constexpr float kLn2 = 0.693147180559945309f;
// Round 6: pair-parallel near-pair phase of the one-env-per-wave kernels (kSym64, not FAR).  The env's unordered listed
// pairs (i < j) are compacted into lanes through the wave's LDS block, every pair is evaluated ONCE for both of its rows
// at full lane utilisation, and each row owner folds its partners' results in ascending partner order (the order of the
// per-lane walk: sums and neighbour lists are bit-identical).  In the product it serves the FUSED ROLLOUTS, whose pair list
// survives from rebuild to rebuild (in-kernel actions -5 %, pool actions -0.3 ... -0.5 %); the single-step kernels measured
// +1.4 ... +2.8 % with it -- the same 319 vector instructions per wave as the per-lane walk's 320, at 65 instead of 56 %
// lane utilisation, behind a longer chain of LDS round trips (profiles/r6_abtest_pair_parallel.log, r6_sq_c3_pair_parallel_step.csv;
// DESIGN.md section 3) -- and keep the walk.  kPairParallelStep (common.hpp: -DDRONESIM_PAIR_PARALLEL_STEP) builds them with it.

// kRolloutPool / kRolloutRand: the fused rollout of the episode layer with its action source known at COMPILE time (pool
// actions / actions drawn in the kernel) -- the one-env-per-wave and N = 256 geometries (every BASELINE shape's rollout):
// no run-time `rand_act` branches in the per-step loop and none of the other source's registers live across it (round 5).
// kRollout itself keeps the run-time flag (the other geometries, and every plain rollout).
enum Mode { kStep = 0, kObserve = 1, kRollout = 2, kRolloutPool = 3, kRolloutRand = 4 };
constexpr bool is_rollout(int mode) { return mode >= kRollout; }

// Developer trace builds (-DDRONESIM_TRACE / -DDRONESIM_TRACE_FINE: kTrace / kTraceFine of common.hpp) stamp per-wave
// phase times into KArgs.trace; in the product build the stamps are dead code.  TRACE_FINE moves stamps 1 and 2 into the
// filter phase of the generic bucket filter (tables built / candidates tested).
#define TRACE_MARK(k) do { if (kTrace && a.trace && (threadIdx.x & 63) == 0) a.trace[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#define TRACE_COARSE(k) do { if (!kTraceFine) TRACE_MARK(k); } while (0)
#define TRACE_FINE(k) do { if (kTraceFine) TRACE_MARK(k); } while (0)

struct KArgs {
    long long *trace;               // developer builds (-DDRONESIM_TRACE) only: per-wave phase timestamps
    int N, c, max_steps, E, T;
    int P, epb;                     // envs per wave (0 when N > 64), envs per workgroup
    float dt, q, b, done_radius, ghost_factor, radius_max, reach_max;
    const float *xF, *d_hat, *delta, *radius;
    const float *xF_lo;             // low-order part of the goal ring (xF + xF_lo = the float64 goal to 2^-48), or NULL
    float *pos, *vel;
    int *t;
    const float *act;
    float *reward, *true_reward, *z;
    int *nbr_idx, *n_coll;
    uint8_t *done;
    const uint8_t *mask;
    float skin;                     // rollout, kSym64: slack radius of the register-resident candidate list
    int bucket;                     // packed geometry: use the bucket far filter (N >= kBucketMinN)
    int far_inm;                    // FAR: some clipped distance can pass a Delta mask (Delta_j >= dhat_i possible, e.g. deltas=None)
    int stage5;                     // c = 5: the 5 (k+1)-word z rows are staged through LDS too and the velocities of the env's
    int lds_vel;                    //        agents sit in LDS at this byte offset (both sized by the host when they fit)
    int uniform;                    // all agents share d_hat, Delta and radius (host-known): constants come
    float dhat_u, delta_u, radius_u;   //   from the kernel arguments, no per-agent table is read
    // episode bookkeeping / in-kernel reset / in-kernel random actions (DroneEpisodeCtl; all off when zero)
    double *acc;                    // DroneEpisodeAcc[E] as 8 x 8 bytes per env, or nullptr
    int auto_reset, rand_act;
    int div_y;
    uint32_t lat_M, key0, key1, gid_base;   // lattice nodes, Philox key (seed), global id of env 0
    float pitch;
    int *episode;
    float *act_out;                 // rand_act: optional record of the actions drawn, [T][E][N][2]
    float *z_final, *pos_final;     // auto_reset: terminal observation / state of the envs that finish (or nullptr)
    int *nbr_final;
    int lds_tail;                   // byte offset of the bookkeeping regions behind the bucket tables
    int samp_tbl, samp_shift;       // in-kernel reset: entries of the sampling table per env slot (2^k >= 2 N), 32 - k
    int do_reset;                   // observe mode only (dronesim_reset_observe): draw the state instead of loading it
    int *node_out;                  //   optional record of the lattice nodes drawn, [E][N]
};

// @phase h_nbr_list
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

// The first K+1 entries of a stable argsort of row i (drone_env.py:338): entry 0 starts as the agent itself.
//   ASC = false: general form, one 64-bit key per entry, valid for any visiting order (the relative-window scans visit
//                partner (i + r) mod N for r = 1, 2, ...).
//   ASC = true:  partners arrive in ASCENDING agent index (bucket filter, symmetric filter, the in-kernel reset's
//                re-observation).  A new partner then follows every earlier partner of equal distance, so a strict
//                float compare orders it exactly; only the agent's own entry (index i, sitting between the j < i and
//                the j > i) needs the index, and a partner can only reach it with d <= d_ii -- exactly coincident
//                agents, or a larger partner overlapping a smaller agent's centre.  That case takes the general
//                insertion, wave-uniformly; everything else inserts behind entry 0 with 5 instructions per stage
//                instead of 8 on 64-bit keys.
template <int K, bool ASC> struct NbrList;
template <int K> struct NbrList<K, false> {
    unsigned long long key[K + 1];
    __device__ __forceinline__ void init(float dii, int agent)
    {
#pragma unroll
        for (int s = 0; s <= K; ++s) key[s] = ~0ull;
        key[0] = nbr_key(dii, agent);
    }
    template <bool DEFER = false>
    __device__ __forceinline__ void insert(float d, int j, float) { nbr_insert<K>(key, nbr_key(d, j)); }
    __device__ __forceinline__ unsigned index(int kth) const { return (unsigned)key[kth]; }
    __device__ __forceinline__ void pin()
    {
#pragma unroll
        for (int s = 0; s <= K; ++s) asm volatile("" : "+v"(key[s]));
    }
};
template <int K> struct NbrList<K, true> {
    float d[K + 1];
    unsigned j[K + 1];
    __device__ __forceinline__ void init(float dii, int agent)
    {
#pragma unroll
        for (int s = 0; s <= K; ++s) { d[s] = __builtin_inff(); j[s] = ~0u; }
        d[0] = dii; j[0] = (unsigned)agent;
    }
    // DEFER: the caller re-runs the whole walk with DEFER = false when `deg` comes back non-zero (a partner that can
    // precede the agent itself was seen) -- the hot loop then holds the 5-instruction stages only, with no second path
    // merging into it (the merge cost 6 register copies and two scalar branches per visited partner)
    bool deg = false;                                        // per lane; the wave's verdict is a ballot after the walk
    template <bool DEFER = false>
    __device__ __forceinline__ void insert(float dn, int jn, float dii)
    {
        if (DEFER) {
            deg |= !(dn > dii);
        } else if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(dn > dii)) != 0ull, 0)) {
            // general insertion on (d, j) keys over all K+1 entries (NaN distances order like their bit patterns)
            unsigned long long key = nbr_key(dn, jn);
#pragma unroll
            for (int s = 0; s <= K; ++s) {
                const unsigned long long cur = j[s] == ~0u && d[s] == __builtin_inff() ? ~0ull : nbr_key(d[s], (int)j[s]);
                const bool lt = key < cur;
                const unsigned long long out = lt ? key : cur;
                key = lt ? cur : key;
                if (out == ~0ull) { d[s] = __builtin_inff(); j[s] = ~0u; }
                else {
                    const unsigned o = (unsigned)(out >> 32);
                    d[s] = __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
                    j[s] = (unsigned)out;
                }
            }
            return;
        }
        float kd = dn;
        unsigned kj = (unsigned)jn;
#pragma unroll
        for (int s = 1; s <= K; ++s) {
            const bool lt = kd < d[s];
            const float cd = d[s];
            const unsigned cj = j[s];
            d[s] = lt ? kd : cd; j[s] = lt ? kj : cj;
            kd = lt ? cd : kd; kj = lt ? cj : kj;
        }
    }
    __device__ __forceinline__ unsigned index(int kth) const { return j[kth]; }
    // the entries as opaque register values at this point (see the hoisted block of drone_kernel)
    __device__ __forceinline__ void pin()
    {
#pragma unroll
        for (int s = 0; s <= K; ++s) asm volatile("" : "+v"(d[s]), "+v"(j[s]));
    }
};

struct Defer { static constexpr bool value = true; };
struct NoDefer { static constexpr bool value = false; };
struct UniArgs { static constexpr bool value = true; };
struct UniRuntime { static constexpr bool value = false; };
template <int K> __device__ __forceinline__ bool list_degenerate(NbrList<K, true> &l)
{
    const bool d = __builtin_amdgcn_ballot_w64(l.deg) != 0ull;
    l.deg = false;
    return d;
}
template <int K> __device__ __forceinline__ bool list_degenerate(NbrList<K, false> &) { return false; }

// @phase h_nan_to_num
__device__ __forceinline__ float nan_to_num_f32(float x)   // np.nan_to_num, drone_env.py:287-288
{
    if (x != x) return 0.0f;
    return fminf(fmaxf(x, -3.402823466e+38f), 3.402823466e+38f);
}

// @phase h_stores
// Output stores.  The outputs of a step are never re-read by the launch that writes them, so they are
// written with the non-temporal (streaming) policy: lines drain to memory while the kernel runs instead
// of sitting dirty in the XCD L2 until the end-of-kernel write-back.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ void st_out(T *p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_out2(float *p, float x, float y)
{
    f32x2 v; v.x = x; v.y = y;
    st_out(reinterpret_cast<f32x2 *>(p), v);
}

// LDS addresses as plain 32-bit integers: a value that is to be formed EARLY (ahead of the state loads' return) and
// kept opaque with an empty asm must not be a generic pointer, or every access through it turns into a flat access
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}

// Global-memory pointers that pass through an empty asm (to pin them in scalar registers early) must carry their address
// space in the type: a generic pointer that comes out of an asm is accessed with FLAT instructions (64-bit per-lane
// addresses, and an lgkmcnt count on every store).
typedef __attribute__((address_space(1))) float g_f32;
typedef __attribute__((address_space(1))) f32x2 g_f32x2;
typedef __attribute__((address_space(1))) int g_i32;
typedef __attribute__((address_space(1))) unsigned g_u32;
typedef __attribute__((address_space(1))) uint8_t g_u8;
typedef __attribute__((address_space(1))) u32x4 g_u32x4;
template <typename T> __device__ __forceinline__ void st_g(__attribute__((address_space(1))) T *p, T v)
{
    __builtin_nontemporal_store(v, p);
}
__device__ __forceinline__ void st_g2(g_f32 *p, float x, float y)
{
    f32x2 v; v.x = x; v.y = y;
    st_g((g_f32x2 *)p, v);
}

// @phase h_copy_out
// Cooperative copy of `n` 4-byte words from a wave's LDS staging area to global memory: 16 bytes per
// lane when the destination is 16-byte aligned (full 128-B lines, 1 KiB per wave-instruction),
// 4 bytes per lane otherwise.
__device__ __forceinline__ void wave_copy_out(unsigned *__restrict__ dst, const unsigned *__restrict__ src,
                                              int n, int lane)
{
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int n4 = n & ~3;
        for (int o = lane * 4; o < n4; o += 4 * kWave)
            st_out(reinterpret_cast<u32x4 *>(dst + o), *reinterpret_cast<const u32x4 *>(src + o));
        if (lane < n - n4) st_out(dst + n4 + lane, src[n4 + lane]);
    } else {
        for (int o = lane; o < n; o += kWave) st_out(dst + o, src[o]);
    }
}

// @phase h_reduce
// ---- fixed-order reductions of one float per lane (episode bookkeeping: sum of the step's rewards per env)

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_or_zero(float v)   // v of the lane CTRL selects, 0 where there is none / masked
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}

// Two sums over all 64 lanes for the price of one tree: the wave's halves are exchanged (v_permlane32_swap: lanes
// 32-63 of `a` <-> lanes 0-31 of `b`) so that lanes 0-31 carry a[l] + a[l+32] and lanes 32-63 carry b[l-32] + b[l];
// one DPP tree over the 32-lane halves (row_shr 1,2,3 -> quads, row_shr 4 / 8 -> rows, row_bcast 15 -> half) then
// leaves sum(a) in lane 31 and sum(b) in lane 63.  The order of the additions is fixed: bit-reproducible.
// Every stage is a full-mask DPP add through the compiler's own builtin (one v_add_f32_dpp each after its DPP
// combine): lanes other than 15 / 31 / 63 of a row end up with partial sums nobody reads, and -- unlike the inline-asm
// form of round 2, whose stages each carried their own s_nop -- the scheduler can fill the two wait states behind
// every stage with the z-row arithmetic that surrounds the call (the chain is dependent; alone it is 8 idle slots).
// (Tried on the matrix pipe instead -- two chained v_mfma_f32_16x16x4_f32 with B = ones per sum: +0.3 us per launch.)
__device__ __forceinline__ float wave_sum64_tree(float a, float b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    const float v = a + b;
    float t = v + dpp_or_zero<0x111, 0xf, 0xf>(v);
    t += dpp_or_zero<0x112, 0xf, 0xf>(v);
    t += dpp_or_zero<0x113, 0xf, 0xf>(v);
    t += dpp_or_zero<0x114, 0xf, 0xf>(t);                   // row_shr:4  (lane 15 of a row: + lanes 8..11's quad sum ...)
    t += dpp_or_zero<0x118, 0xf, 0xf>(t);                   // row_shr:8
    t += dpp_or_zero<0x142, 0xf, 0xf>(t);                   // row_bcast:15 (lane 15 of a row -> the next row)
    return t;                                               // sum(a) in lane 31, sum(b) in lane 63
}
__device__ __forceinline__ float2 wave_sum64_pair(float a, float b)
{
    const float t = wave_sum64_tree(a, b);
    return make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 31)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63)));
}

// inclusive prefix sum of one int per lane over the wave's 64 lanes: quads and rows by row_shr 1, 2, 3 / 4 / 8 (lanes a shift
// would pull from outside their row of 16 read zero), rows to the wave by row_bcast 15 / 31 -- seven DPP adds
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true); }
__device__ __forceinline__ int wave_incl_scan64(int v)
{
    int t = v + dpp_or_zero_i<0x111, 0xf>(v);
    t += dpp_or_zero_i<0x112, 0xf>(v);
    t += dpp_or_zero_i<0x113, 0xf>(v);                       // lane i of a row: v[i-3] + ... + v[i]
    t += dpp_or_zero_i<0x114, 0xf>(t);                       // ... v[i-7] + ... + v[i]
    t += dpp_or_zero_i<0x118, 0xf>(t);                       // prefix inside the row of 16
    t += dpp_or_zero_i<0x142, 0xa>(t);                       // rows 1, 3 += lane 15 of rows 0, 2
    t += dpp_or_zero_i<0x143, 0xc>(t);                       // rows 2, 3 += lane 31
    return t;
}

// sum over the n consecutive lanes of a segment (an env slot of a packed wave); valid in the segment's first lane
// (`idx` = position inside the segment).  Fixed tree over lane offsets 1, 2, 4, ...: bit-reproducible.
__device__ __forceinline__ float segment_sum(float v, int idx, int n)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float o = __shfl_down(v, off, kWave);
        v += (idx + off < n) ? o : 0.0f;
    }
    return v;
}

// RandomAgent.forward (SAC_agents.py:9-22): clip(-1 + 2 rand, -1, 1) on the 2^24-point grid of [-1, 1)
__device__ __forceinline__ float unit_action(uint32_t w) { return fmaf((float)(w >> 8), 1.1920928955078125e-07f, -1.0f); }

__device__ __forceinline__ uint32_t philox4x32_10_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1)
{
    uint32_t o[4];
    philox4x32_10(c0, c1, c2, c3, k0, k1, o);
    return o[0];
}
constexpr uint32_t kRandActKey = 0x52414E44u;   // "RAND": separates the action stream from the reset stream

// @phase h_misc
// Workgroup geometries
//   kPacked  : N <= 64.  256 (N > 32: 128) threads = 4 (2) independent waves; each wave holds floor(64/N) whole envs and
//              touches only its own LDS, so all synchronisation is wave-local (no s_barrier).
//   kSym64   : kPacked specialised for N == 64 without far agents: every unordered pair is evaluated
//              ONCE (lane i tests partners i+1..i+32; the verdict reaches the partner as a rotated
//              ballot), halving the far-filter arithmetic.
//   kBlock256 / kBlock1024 : N > 64, one workgroup per env, thread = agent.
//   kBlockU256: kBlock256 specialised for N == 256 with uniform (d_hat, Delta, l) and no far agents (BASELINE configs[4]):
//              four full waves per env, so lane masks, word counts, the LDS carve-up and the per-agent constants are
//              compile-time / scalar (what kSym64 is to kPacked).  FUSED ROLLOUTS ONLY: there the per-step scalar
//              bookkeeping it removes is 12-13 % of a step (C5 shard 2.46 -> 2.15 us per step, with in-kernel actions
//              and the episode layer 3.32 -> 2.91; N = 256 x 4096 envs 13.8 -> 12.0); the single-step kernels measured
//              +1 ... +6 % with it (their launch is a latency chain that the fully unrolled, hoisted form lengthens:
//              profiles/r4_abtest_block_u256.log), so step / observe keep kBlock256.
enum Geo { kPacked = 0, kSym64 = 1, kBlock256 = 2, kBlock1024 = 3, kBlockU256 = 4 };
// Occupancy targets (waves per SIMD the launch bounds ask for; tools/kernel_resources.py lists what every instantiation got)
constexpr int kSymStepWaves = 8;
constexpr int kBlockStepWaves = 8;
constexpr int kBlockStepWavesEpi = 6;      // (8 makes the episode-layer kernels spill on the hot path)
// 4 = the 128-register budget (small spills): N = 256 x 4096 envs with in-kernel actions 14.1 us per step against 16.4 at
// 3 (168 registers, no spills).  (The k = 3 kernel of this family was the one hipcc mis-lowered while the re-observation
// after an in-kernel reset still walked one exec-masked loop per word -- see the cold path; tests/test_gpu_fuzz.py and
// test_rollout_with_pool_actions_equals_steps_across_resets guard it.)
constexpr int kBlockRolloutEpiWaves = 4;
// Round 5: the packed rollout of the episode layer (run-time choice of pool / in-kernel actions, records, in-kernel reset:
// all of it live across the per-step loop) spilled 21 registers INSIDE the loop at the 128-register budget; at 3 waves
// per SIMD (168 registers) the loop is spill-free and C2's rollout runs 2.97 -> 2.54 us per step, 5 x 65536 envs
// 10.5 -> 7.5 (profiles/r5_abtest_rollout_epi_waves.log; tests/test_host_logic.py:
// test_no_rollout_kernel_spills_on_its_hot_path).  The general workgroup-per-env rollout (65 ... 255 agents, or N = 256
// with non-uniform constants) was measured the same way and KEEPS 4: spill-free at 168 registers it is 3 ... 13 % SLOWER
// (N = 128 x 4096 envs 8.62 -> 8.91 us, with in-kernel actions 8.15 -> 9.26; N = 200 x 2048 envs 7.23 -> 7.99) -- its 18
// scratch accesses per step cost less than the fourth wave per SIMD.  kBlockU256 (N = 256, uniform constants: BASELINE
// configs[4]) holds no spill in its loop at 128 registers (only the out-of-line reset does).
constexpr int kPackedRolloutEpiWaves = 3;

template <int GEO> struct GeoTraits {
    static constexpr int kMaxThreads = GEO == kBlock1024 ? 1024 : 256;
    // C3 needs 4 resident waves per SIMD (4096 envs = 4096 waves on 1024 SIMDs): cap the register budget at 128.
    // kSym64 single-step launches: 64 registers and < 5 KiB of LDS per wave, so that 8 waves per SIMD are resident and a
    // launch of up to 8192 envs runs in one generation (the fused rollout keeps its 112-128 registers: LICM of a 200-step
    // loop, and its launches never hold more than 4096 envs per 4 waves anyway)
    // (the neighbour list, the rows and their staging grow with k: the 64-register budget holds without spills for
    // k <= 2 -- every BASELINE config --, 80 registers for k <= 4, the round-2 budget of 128 beyond;
    // tools/kernel_resources.py lists registers / scratch of every instantiation of a built library)
    // (FAR: the c = 5 rows and the far tail want registers too -- at 64 the C5-sized default construction was 7.6 % slower)
    static constexpr int min_waves(int k, int mode, bool epi, bool far)
    {
        const int want = GEO == kBlock1024 ? 1 : (GEO == kSym64 && !is_rollout(mode)) ? kSymStepWaves
                                                                          : ((GEO == kBlock256 || GEO == kBlockU256) && !is_rollout(mode)) ? ((epi || far) ? kBlockStepWavesEpi : kBlockStepWaves)
                                     : (GEO == kBlockU256 && epi) ? kBlockRolloutEpiWaves
                                     : (GEO == kPacked && epi && !far) ? kPackedRolloutEpiWaves : 4;   // (rollouts AND the packed step
        // kernels of the episode layer: the latter were not the reason for the 168-register budget, but measured faster with it too --
        // C2's graded step kernel 4.29 -> 4.19 us -- and stay included on purpose)
        const int cap = k <= 2 ? 8 : k <= 4 ? (((far || epi) && GEO == kBlock256 && !is_rollout(mode)) ? 5 : 6) : 4;   // (k = 3 / 4, FAR or episode layer: 8-12 B of spills at 80 registers)
        return want < cap ? want : cap;
    }
    static constexpr bool kWaveLocal = GEO == kPacked || GEO == kSym64;
};

// cells per axis of the generic bucket tables: ONE predicate for the kernel's carve-up and the host's LDS size (dronesim.hip).
// Single-step launches of the workgroup-per-env geometries of up to 256 agents take kCellsBlock (see there), everything else kCells.
__host__ __device__ constexpr int bucket_cells(int geo, bool rollout)
{
    return ((geo == kBlock256 || geo == kBlockU256) && !rollout) ? kCellsBlock : kCells;
}
// workgroup-per-env geometries: float2 entries of the position tile (N, the over-read slack, even for 16-byte alignment)
__host__ __device__ constexpr int block_pos_entries(int N) { return (N + kPad + 1) & ~1; }
// kSym64's LDS block per wave: [64 positions][staging: 64 x (z row + Ni row)][x cells | y cells] -- see the carve-up
// (zc = columns of a staged z row: 2, or 5 for the FAR variant with c = 5 rows)
constexpr int sym_wave_bytes(int K, int zc = 2) { return 64 * 8 + 64 * (zc + 1) * (K + 1) * 4 + 2 * kCells * 8; }

template <bool WAVE_LOCAL>
__device__ __forceinline__ void group_sync()
{
    if (WAVE_LOCAL) {          // a wave's own LDS traffic is ordered; only the compiler must not reorder
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// EPI = true adds the episode bookkeeping of the *_ex entry points (DroneEpisodeCtl): per-env running sums, in-kernel
// reset of finished envs, in-kernel random actions.  EPI = false is the plain step / observe / rollout: none of that
// code exists in it.
// byte offset of `rest` in drone_kernel's kernel-argument segment: two pointers + four ints in front of it
constexpr int kKArgsOffset = 2 * 8 + 4 * 4;
static_assert(kKArgsOffset % alignof(KArgs) == 0, "KArgs sits right behind the leading scalar arguments");

template <int K, bool FAR, int MODE, int GEO, bool EPI>
__global__ void __launch_bounds__(GeoTraits<GEO>::kMaxThreads, GeoTraits<GEO>::min_waves(K, MODE, EPI, FAR)) drone_kernel(
    // The first 8 dwords of the kernel arguments are preloaded into SGPRs at wave launch (Makefile:
    // -amdgpu-kernarg-preload-count=8; only leading scalar arguments qualify, not the struct): exactly what a
    // wave needs to issue its pos / act loads, which therefore no longer wait for a kernel-argument fetch
    // (-0.2 us per launch at C3).  They override the same-named fields of `rest`.
    float *pos, const float *vel_or_act, int P, int epb, int E, int n_agents, const KArgs rest)
{
    // @phase setup
    KArgs a = rest;
    // EPI: which parts of DroneEpisodeCtl are in use travels in the high half of the preloaded `epb` argument
    // (bit 16 records, bit 17 auto-reset, bit 18 random actions), so that the branches that depend on it never wait
    // for the kernel-argument fetch ahead of the first pos / act loads
    const int epi_flags = EPI ? (epb >> 16) : 0;
    if (EPI) epb &= 0xffff;
    const unsigned xcd_blocks = (unsigned)P & ~255u;         // workgroups in whole groups of 256 (XCD map below)
    a.pos = pos; a.P = P & 255; a.epb = epb; a.E = E; a.N = n_agents;
    if (MODE == kObserve) a.vel = const_cast<float *>(vel_or_act); else a.act = vel_or_act;
    constexpr bool WL = GeoTraits<GEO>::kWaveLocal;
    constexpr bool SYM = GEO == kSym64;
    // generic bucket filter (see below): always for N > 64, for packed envs when the host asks for it (N >= kBucketMinN)
    constexpr bool BU = GEO == kBlockU256;                   // N == 256, uniform constants: four FULL waves per env
    constexpr bool B256 = GEO == kBlock256 || BU;            // workgroup-per-env, at most four 64-agent words
    constexpr bool BLOCKGEO = B256 || GEO == kBlock1024;
    constexpr int WMAX = GEO == kBlock1024 ? 16 : B256 ? 4 : 1;   // 64-agent words per env
    static_assert(!(BU && FAR), "kBlockU256 assumes far agents never matter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TRACE_MARK(0);
    const long long trace_rt0 = (kTrace || kTraceSpan) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz, the same clock on every XCC
    const int N = SYM ? 64 : BU ? 256 : a.N;
    const int tid = threadIdx.x;
    const unsigned lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = BU ? 4 : blockDim.x >> 6;
    // Every wave covers a CONTIGUOUS range of global agents [wga0, wga0 + nval): lane l <-> agent wga0 + l.
    // wga0 / nval are wave-uniform (SGPRs), so every per-agent array is addressed as uniform base + lane.
    int slot, agent, env0, nval;
    // XCD-aware workgroup -> env mapping.  Consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2;
    // with env = workgroup id the 128-byte lines of the per-env arrays (t, n_coll: 32 envs per line, done: 128) were
    // written piecewise by all 8 L2s and left the chip as 8 partial write-backs at the end of the launch.  Every XCD
    // now owns runs of 32 consecutive virtual workgroups (128 envs at N = 64): whole lines per L2 (-0.11 us per launch at C3).
    // (the number of workgroups in whole groups of 256 travels in the high bits of the preloaded `P` argument: reading
    // gridDim would put a scalar fetch and its wait ahead of the first state loads)
    unsigned vb = blockIdx.x;
    if (vb < xcd_blocks) vb = ((vb >> 8) << 8) + ((vb & 7u) << 5) + ((vb >> 3) & 31u);
    if (SYM) {                                               // one env per wave: the shortest way to the first loads
        slot = wave;
        agent = (int)lane;
        env0 = (int)vb * a.epb + wave;
        nval = env0 < a.E ? kWave : 0;
    } else if (WL) {                                         // lane -> (env slot inside the wave, agent)
        const int sub = (int)lane / N;
        slot = wave * a.P + sub;
        agent = (int)lane - sub * N;
        env0 = (int)vb * a.epb + wave * a.P;
        nval = max(0, min(a.P, a.E - env0)) * N;
    } else {
        slot = 0;
        agent = tid;
        env0 = (int)vb;
        nval = BU ? kWave : max(0, min(kWave, N - wave * kWave));
    }
    const size_t wga0 = (size_t)env0 * N + (WL ? 0 : wave * kWave);
    const int env = env0 + ((WL && !SYM) ? slot - wave * a.P : 0);
    const bool masked = MODE == kObserve && a.mask != nullptr;
    bool valid;
    if (SYM) {
        // one env per wave and waves never synchronise with each other: a wave without an env (ragged last workgroup,
        // masked-out env) simply leaves, and in every other wave all 64 lanes are agents -- no lane masking anywhere
        // below, which also lets the scheduler move code across what would otherwise be exec-mask boundaries
        if (nval == 0) return;
        if (masked && a.mask[env] == 0) return;
        valid = true;
    } else if (BU) {
        // four full waves of ONE env: a masked-out env's whole workgroup leaves (no barrier is left waiting), and in
        // every other workgroup all 256 lanes are agents -- no lane masking anywhere below
        if (masked && a.mask[env] == 0) return;
        valid = true;
    } else {
        valid = (int)lane < nval;
        if (masked && valid) valid = a.mask[env] != 0;
    }
    const size_t step_agents = (size_t)a.E * N;              // rollout: per-step output stride
    // all agents share d_hat, Delta and radius (host-known): the constants are kernel-argument scalars.  kSym64 is only
    // chosen for such envs (launch()), so that its constants are wave-uniform at compile time: no per-agent loads, whose
    // return the early (hoisted) uses would otherwise wait for behind the state loads
    const bool uniform = SYM || BU || a.uniform != 0;

    // @phase loads
    // ---- longest-latency loads first: this agent's state (HBM), then the shared constants (L2)
    float xi = 0.f, yi = 0.f, vxi = 0.f, vyi = 0.f;
    float2 u0 = make_float2(0.f, 0.f);
    int tcur = 0;
    // episode bookkeeping (dronesim_*_ex): the running sums of an env's DroneEpisodeAcc record live in the registers
    // of its agent-0 lane for the whole launch; `epi` = resets the env has seen (stream id of reset and actions)
    static_assert(!(EPI && MODE == kObserve), "observe has no episode bookkeeping");
    static_assert(!((MODE == kRolloutPool || MODE == kRolloutRand) && !EPI), "the compile-time action source belongs to the episode layer");
    // (kSym64: the word is made opaque so that every test is one s_bitcmp on it; as boolean values the compiler keeps
    // them as 64-bit lane masks and spends a v_cndmask / v_cmp pair on each negation)
    int epi_word = epi_flags;
    if (!kTrace && SYM && EPI && !is_rollout(MODE)) asm volatile("" : "+s"(epi_word));
#define has_acc (EPI && (epi_word & 1) != 0)
#define auto_reset (EPI && (epi_word & 2) != 0)
    const bool rand_act = MODE == kRolloutRand ? true : MODE == kRolloutPool ? false : (EPI && is_rollout(MODE) && (epi_flags & 4) != 0);
    // the 32 hot bytes of the env's record live in registers for the whole launch, split over two lanes so that one
    // load and one store instruction move them: agent 0 holds (ep_return, ep_true_return) as two doubles, agent 1
    // holds (ep_collisions, ep_len, episodes, reserved) as four ints
    uint4 accw = make_uint4(0u, 0u, 0u, 0u);
    uint32_t epi = 0u;
    uint32_t rnd[4] = {0u, 0u, 0u, 0u};                      // rand_act: the Philox block of steps (t & ~1, t | 1)
    const uint32_t gid = a.gid_base + (uint32_t)env;         // global env id (independent of the sharding)
    float xFx = 0.f, xFy = 0.f, xLx = 0.f, xLy = 0.f, dhat = 1.f, delta_i = 0.f, li = 0.f;
    // base addresses of the first loads, computed on the scalar unit before the branch
    const float2 *pos_in = reinterpret_cast<const float2 *>(a.pos) + wga0;
    const float2 *vel_in = reinterpret_cast<const float2 *>(MODE == kObserve ? a.vel : a.act) + wga0;
    asm volatile("" : : "s"(pos_in), "s"(vel_in));
    // dronesim_reset_observe (observe mode, launch-uniform flag): the state is DRAWN below -- env.reset() as one launch --
    // instead of being loaded; none of this exists in the step / rollout instantiations
    const bool fresh = MODE == kObserve && a.do_reset != 0;
    if (valid) {
        // read once per launch: streaming loads (the state and the actions do not displace anything in L2)
        // (measured: -0.1 us at C3; the workgroup-per-env shapes at BASELINE size -- C5 shard, 2 MB of state --
        // are 0.2 us faster with plain loads, and a run-time choice costs more than either)
        float2 p = make_float2(0.f, 0.f);
        if (fresh) {
        } else if (BLOCKGEO) {
            p = pos_in[lane];
        } else {
            const f32x2 pl = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(pos_in) + lane);
            p = make_float2(pl.x, pl.y);
        }
        if (MODE == kObserve) {
            if (!fresh) {
                const float2 v = vel_in[lane];
                vxi = v.x; vyi = v.y;
            } else {
                epi = (uint32_t)a.episode[env];               // resets this env has seen so far: the stream id of the draw
            }
        } else {
            if (rand_act) {
                // no action pool: the first action is drawn below, once t and the episode counter have arrived
            } else if (BLOCKGEO) {
                u0 = vel_in[lane];
            } else {
                const f32x2 ul = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(vel_in) + lane);
                u0 = make_float2(ul.x, ul.y);
            }
            if (rand_act) {                                  // every lane follows its env's counters
                tcur = a.t[env];
                epi = (uint32_t)a.episode[env];
            } else if (agent == 0) {
                tcur = a.t[env];
            }
            if (has_acc && agent < 2)                        // (read ahead of the hoisted bases: its address is formed in place)
                accw = *reinterpret_cast<const uint4 *>(a.acc + 8 * (size_t)env + 2 * agent);
        }
        const float2 g = reinterpret_cast<const float2 *>(a.xF)[(unsigned)agent];
        if (a.xF_lo) {                                        // wave-uniform
            const float2 gl = reinterpret_cast<const float2 *>(a.xF_lo)[(unsigned)agent];
            xLx = gl.x; xLy = gl.y;
        }
        if (uniform) {
            dhat = a.dhat_u; delta_i = a.delta_u; li = a.radius_u;
        } else {
            dhat = a.d_hat[(unsigned)agent];
            delta_i = a.delta[(unsigned)agent];
            li = a.radius[(unsigned)agent];
        }
        xi = p.x; yi = p.y;
        xFx = g.x; xFy = g.y;
    }

    // @phase hoist
    // ---- What the epilogue needs from the kernel arguments, and this wave's output bases, are fetched / formed HERE, in
    // the shadow of the state loads (~2000 cycles during which the wave has nothing else to do).  Left to the compiler,
    // each scalar load and each 64-bit base addition sits next to its first use -- behind the loads' return, on the
    // wave's critical path, where every instruction costs the launch time (a lone wave issues one instruction per
    // ~9 cycles, tools/ubench_valu.hip).  The empty asm statements make the values opaque at this point, so they
    // cannot be re-materialised or sunk.  All of them are wave-uniform (scalar registers).
    constexpr int kZRow = 2 * (K + 1), kNRow = K + 1;        // words per agent in the c = 2 layout
    const bool w_reward = a.reward != nullptr, w_true = a.true_reward != nullptr, w_ncoll = a.n_coll != nullptr;
    g_f32 *o_reward = (g_f32 *)(a.reward + wga0), *o_true = (g_f32 *)(a.true_reward + wga0);   // (dereferenced only under w_*)
    g_f32 *o_pos = (g_f32 *)(a.pos + 2 * wga0), *o_vel = (g_f32 *)(a.vel + 2 * wga0);
    g_u32 *o_gz = (g_u32 *)(reinterpret_cast<unsigned *>(a.z) + wga0 * kZRow);      // staged (c = 2) rows of this wave's agents
    g_u32 *o_gn = (g_u32 *)(reinterpret_cast<unsigned *>(a.nbr_idx) + wga0 * kNRow);
    float k_q = a.q, k_b = a.b, k_ghost = a.ghost_factor, k_done_radius = a.done_radius;
    int k_last_t = a.max_steps - 1;
    // one env per wave: the per-env words have scalar addresses too
    g_i32 *o_ncoll = (g_i32 *)(a.n_coll + (SYM ? env0 : 0));
    g_u8 *o_done = (g_u8 *)(a.done + (SYM ? env0 : 0));
    g_i32 *o_t = (g_i32 *)(a.t + (SYM ? env0 : 0));
    g_u32x4 *o_acc = (g_u32x4 *)(a.acc + 8 * (size_t)(SYM ? env0 : 0));
    // (kSym64 only for now: the workgroup-per-env kernels of the episode layer sit at 85 scalar registers without them;
    // not the fused rollout with the episode layer either, which is at its register limits as it is)
    // (round 5, with the action source at compile time: the pool-action rollout WITH the pins measured the same as without,
    // 3.19 us per step at C3, at 120 instead of 113 registers: profiles/r5_abtest_rollout_action_source.log)
    constexpr bool PIN = !kTrace && SYM && !(is_rollout(MODE) && EPI);   // (the trace build's stamps make hipcc lose the uniformity)
    if (PIN && MODE != kObserve)
        asm volatile("" : "+s"(o_reward), "+s"(o_true), "+s"(o_pos), "+s"(o_vel), "+s"(o_gz), "+s"(o_gn),
                          "+s"(k_q), "+s"(k_b), "+s"(k_ghost), "+s"(k_done_radius), "+s"(k_last_t));
    else if (PIN)
        asm volatile("" : "+s"(o_reward), "+s"(o_true), "+s"(o_gz), "+s"(o_gn), "+s"(k_q), "+s"(k_b), "+s"(k_ghost));
    if (PIN && MODE != kObserve) asm volatile("" : "+s"(o_ncoll), "+s"(o_done), "+s"(o_t));
    if (PIN && EPI) asm volatile("" : "+s"(o_acc));

    // @phase lds_setup
    // ---- LDS carve-up (all region sizes multiples of 16 bytes); wave-local geometries give every wave
    //      its own copy of the (Delta_j, l_j) table so that no cross-wave barrier is ever needed
    // positions of one env slot: S0[m] = dup[m] and S1[m + 1] = dup[m], dup = x_0..x_{N-1}, x_0..x_{N-1}
    // (two copies one element apart, so every lane has a copy in which its partner window is 16-byte aligned)
    const int stride = 2 * N + kPad;                         // float2 per copy (even)
    const int nconst = WL ? nwaves : 1;
    float2 *spos = reinterpret_cast<float2 *>(smem);                                       // [epb][2][stride]
    // (workgroup-per-env: always the bucket filter, so only the N positions themselves -- read by agent index -- plus the
    // slack the crowded path's 16-partner reads may run into: 2.2 instead of 8.5 KB at N = 256)
    float2 *sconst_all = spos + (BLOCKGEO ? (size_t)block_pos_entries(N) : (size_t)a.epb * 2 * stride);   // [nconst][N + (N&1)]
    int *sred = reinterpret_cast<int *>(sconst_all + (size_t)nconst * (N + (N & 1)));      // [epb][2]
    const int epb_c = BLOCKGEO ? 1 : a.epb;                  // envs per workgroup
    const int nred = 2 * epb_c + ((2 * epb_c) & 3 ? 4 - ((2 * epb_c) & 3) : 0);
    unsigned *sstage = reinterpret_cast<unsigned *>(sred + nred);
    // c = 2: [64][kZRow] z words + [64][kNRow] Ni words per wave; FAR with staged c = 5 rows: 5 (K+1) z words per lane
    const int zrow_w = (FAR && a.stage5) ? 5 * (K + 1) : kZRow;
    unsigned *stage_z = sstage + (size_t)wave * kWave * (zrow_w + kNRow);
    // kSym64 (uniform constants, one env per wave) has its own, smaller carve-up: one block per wave of
    // [64 positions | staging area | x cells | y cells] = sym_wave_bytes(K) (3840 B at k = 2: with the episode layer's
    // tail 9.5 KiB per two-wave workgroup, 16 workgroups = 8 waves per SIMD per CU).  No (Delta_j, l_j) table, no per-env
    // verdict words; the doubled / shifted position copies of the crowded fallback, which only it reads, run on from the
    // 64 positions INTO the staging area (and, for k = 1, the cell tables, which are dead by then): the staging area is
    // not written before the epilogue, and the rows' neighbour positions are read from the first 64 entries only.
    const int sym_zc = (FAR && a.stage5) ? 5 : 2;            // (FAR: the reference's default construction at N = 64, round 4)
    char *const sym_block = smem + (size_t)wave * sym_wave_bytes(K, sym_zc);
    if (SYM) stage_z = reinterpret_cast<unsigned *>(sym_block + 64 * 8);
    unsigned *stage_n = stage_z + kWave * zrow_w;
    float2 *sconst = sconst_all + (WL ? (size_t)wave * (N + (N & 1)) : 0);
    // bucket filter tables.  kSym64: per wave, cell -> lane mask, entries -1..64 of (x mask, y mask).
    // Other geometries: per env slot, [axis][W words of 64 agents][64 cells].
    unsigned long long *sbt_all = reinterpret_cast<unsigned long long *>(sstage + (size_t)nwaves * kWave * (zrow_w + kNRow));
    // kSym64: [x | y][64 cells] masks of 8 bytes per wave.  Agents live in cells 1..62 (coordinates beyond clamp to the
    // end cells, which only adds candidates), so cells 0 and 63 stay empty: every lane zeroes its own entry of either
    // table and reads cells c-1, c, c+1 without a guard row or an edge test.  (Round 2 kept (x, y) pairs at a 16-byte
    // stride: cells c and c + 8 then shared their banks -- 2-3 way conflicts on the ds_or / ds_read of every launch,
    // 40 % of the kernel's LDS cycles; at 8 bytes per cell the 23 cells of C3 are conflict-free.)
    unsigned long long *sbx = SYM ? reinterpret_cast<unsigned long long *>(sym_block + 64 * 8 + 64 * (sym_zc + 1) * (K + 1) * 4)
                                  : sbt_all + (size_t)wave * (2 * kCells);
    unsigned long long *sby = sbx + kCells;
    const int W = BU ? 4 : BLOCKGEO ? nwaves : 1;
    const bool use_bucket = !SYM && (BLOCKGEO || a.bucket != 0);                           // launch-uniform
    // cells per axis of the generic tables (single-step launches of the workgroup-per-env geometries: see kCellsBlock; their
    // fused rollouts keep 64 -- the candidate list's cells are 1.4 reaches wide, 65 of them at G = 256, and the tables are
    // zeroed at every rebuild: 128 cells measured +0.4 ... +0.9 % there, profiles/r5_abtest_block_cells.log)
    // (N > 256 keeps 64 as well: 16 words of 128 cells on two axes are 32 KiB of an LDS tile that N = 1024 with k = 8 fills)
    constexpr int NCELL = bucket_cells(GEO, is_rollout(MODE));
    unsigned long long *sbt = sbt_all + (size_t)slot * (2 * NCELL) * W;                    // this lane's env
    // tail (episode bookkeeping): per-wave partial reward sums [nwaves][2], then the sampling tables of the in-kernel
    // reset, [epb][samp_tbl] x (node, owner)
    float *spart = reinterpret_cast<float *>(smem + a.lds_tail);

    // (Delta_j, l_j) of a partner: kernel-argument scalars when all agents share them -- except in the fused
    // rollout, whose register budget is tighter (there the LDS table is the cheaper source)
    const bool uni_args = SYM || BU || (!is_rollout(MODE) && uniform);   // (kSym64: always uniform; scalar registers cost it nothing)
    if (WL) {
        if (!SYM && (int)lane < 2 * a.P) sred[2 * wave * a.P + lane] = 0;   // kSym64 keeps these verdicts in scalar registers
        if (!uni_args && (int)lane < N)
            sconst[lane] = uniform ? make_float2(a.delta_u, a.radius_u) : make_float2(a.delta[lane], a.radius[lane]);
    } else {
        if (tid < 4) sred[tid] = 0;                          // (collisions, outside-the-goal flag, CACHED_B's "moved" flag, spare)
        if (!uni_args)
            for (int s = tid; s < N; s += blockDim.x)
                sconst[s] = uniform ? make_float2(a.delta_u, a.radius_u) : make_float2(a.delta[s], a.radius[s]);
    }

    // pair-parallel phase (kPairParallel): its LDS regions.  The pair list (i | j << 8, i < j; 2 bytes per pair) takes the
    // place of the x cell table, which is dead once the candidates have been read (in the fused rollout the list lives there
    // until the next rebuild zeroes the tables); results (d, log term, pair: 16 bytes per pair slot) and the rows' inboxes (one
    // 64-bit mask of pair slots per agent) use the staging area, which is not written before the epilogue.
    constexpr bool PP = SYM && !FAR && (is_rollout(MODE) || kPairParallelStep);
    const unsigned pp_pairs_a = lds_addr(smem) + (unsigned)(reinterpret_cast<const char *>(sbx) - smem);
    const unsigned pp_res_a = lds_addr(smem) + (unsigned)(reinterpret_cast<const char *>(stage_z) - smem);
    unsigned long long *const pp_in = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(stage_z) + 16 * kWave);
    static_assert(!PP || 64 * 3 * (K + 1) * 4 >= 16 * kWave + 8 * kWave, "results + inboxes fit the staging area");
    if (SYM && !is_rollout(MODE)) { sbx[lane] = 0ull; sby[lane] = 0ull; }   // (the fused rollout zeroes them per rebuild)
    if (PP && !is_rollout(MODE)) pp_in[lane] = 0ull;
    // Workgroup-per-env, single step: the cell tables are zeroed HERE and the barrier that orders the zeroing (and the
    // words above) against the other waves' atomics is taken in the shadow of the state loads -- an LDS-only barrier
    // (__syncthreads() carries a release fence, i.e. a vmcnt(0) wait for the loads in flight).  One barrier instead of
    // two, and no zeroing loop, between the loads' return and the first mask read.
    constexpr bool EARLY_TABLES = BLOCKGEO && !is_rollout(MODE);
    if (EARLY_TABLES) {
        for (int o = tid; o < 2 * NCELL * W; o += blockDim.x) sbt_all[o] = 0ull;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    const float reach = SYM ? a.reach_max : dhat + li + a.radius_max;
    float thr = reach * reach * 1.000001f;                   // early-out radius^2 (conservative)
    float log2_dhat = __builtin_amdgcn_logf(dhat);           // v_log_f32 = log2
    float2 *spos_env = SYM ? reinterpret_cast<float2 *>(sym_block) : spos + (size_t)slot * 2 * stride;   // S0 of this lane's env
    // pass-1 window of this lane starts at dup index agent + (odd r): pick the copy where that is even
    const float2 *pwin = (agent & 1) ? spos_env + agent + 1 : spos_env + stride + agent + 2;
    const int nsteps = (is_rollout(MODE)) ? a.T : 1;
    // c = 5 rows carry (v, l) of tie-ordered agents, which forces the FAR variant on the host: every other
    // instantiation knows c = 2 at compile time (no dead c = 5 code, and no conservative s_waitcnt for its loads)
    const int zc = FAR ? a.c : 2;
    const bool staged = (zc == 2 || (FAR && a.stage5)) && !masked;   // z / Ni leave through LDS as full lines
    // FAR, c = 5: (vx, vy) of the env's agents in LDS (the rows of the k nearest carry them, :367 / :385)
    // (addressed as integer LDS offsets, see the staging addresses below)
    const unsigned svel_a = lds_addr(smem) + (unsigned)a.lds_vel + (unsigned)slot * (unsigned)N * 8u;

    // ---- candidate list (fused rollout of kSym64 only): the far filter is run with radius reach + skin and
    // its verdicts are kept in registers until some agent of the env has moved more than skin/2 from where
    // they were taken -- by the triangle inequality every pair inside `reach` is then still on the list.
    // Outputs are bit-identical to filtering every step: listed pairs beyond `reach` are skipped by the
    // exact test in pass 2.
    // (round 3: also the workgroup-per-env rollout of up to 256 agents, CACHED_B -- same list, one 64-bit word per 64
    // partners, the "somebody moved" verdict agreed through an LDS word at the step's first barrier)
    constexpr bool CACHED_B = B256 && is_rollout(MODE) && !FAR;
    // (not for FAR: its far tail needs the EXACT near set, and a listed-but-currently-far partner is not in it)
    constexpr bool CACHED = (SYM && !FAR && is_rollout(MODE)) || CACHED_B;
    const float thr_list = CACHED ? (reach + a.skin) * (reach + a.skin) * 1.000001f : thr;
    const float moved2 = 0.49f * a.skin * 0.49f * a.skin;
    const float inv_cell = __builtin_amdgcn_rcpf((CACHED ? (SYM ? reach : a.reach_max) + a.skin : a.reach_max) * 1.001f);   // bucket filter: cells a little wider than the list radius
    unsigned long long cand = 0ull;
    unsigned long long candw[CACHED_B ? WMAX : 1];           // CACHED_B: the listed partners, one word per 64 agents
#pragma unroll
    for (int w = 0; w < (CACHED_B ? WMAX : 1); ++w) candw[w] = 0ull;
    float refx = __builtin_nanf(""), refy = refx;            // NaN = no list yet
    // pair-parallel phase: this lane's segment of the pair list (its partners j > lane: slots pp_off .. pp_off + pp_cnt - 1)
    // and, in the fused rollout, the number of listed pairs (-1: no pair list, the per-lane masks `cand` are walked instead)
    int pp_off = 0, pp_cnt = 0, pp_T = -1;
    unsigned pp_pr = 0u;                                     // fused rollout: this PAIR lane's pair, read back once per rebuild
    const unsigned long long above = ~((2ull << lane) - 1ull);   // lanes > this lane
    // self entry: d_ii = min(-2 l_i, dhat_i), ratio 1 -> log 0, never a collision (:323-325); N_delta[i,i] uses Delta_i (:346)
    float dii = fminf(-li - li, dhat);
    int in_range0 = ((dii <= delta_i) ? 1 : 0) - 1;
    // this lane's slots of the wave's staging area (c = 2 rows leave through LDS as full lines): row of 2 (K+1) words of
    // z, row of K+1 words of Ni, and the 16 bytes per lane and round of the copy-out
    // (as offsets from the dynamic-LDS base: casting the derived generic pointers themselves makes hipcc emit an aperture
    // null test that it then fails to select -- "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
    const unsigned lds0 = lds_addr(smem);
    const unsigned stage_z_a = lds0 + (unsigned)(reinterpret_cast<const char *>(stage_z) - smem);
    unsigned zrow_a = stage_z_a + lane * (4 * kZRow);
    unsigned nrow_a = lds0 + (unsigned)(reinterpret_cast<const char *>(stage_n) - smem) + lane * (4 * kNRow);
    unsigned copy_a = stage_z_a + lane * 16;
    // formed in the shadow of the state loads, like the scalar side above
    unsigned long long self_bit = 1ull << lane;              // this lane's bit in the cell masks of the bucket filter
    // partners reach pass 2 in ascending agent order on every path of these geometries (bucket / symmetric filter)
    constexpr bool ASC = (SYM || BLOCKGEO) && !FAR;
    // start values of pass 2 (row sums, collision count, the neighbour list holding the agent itself): in a single-step
    // launch they are register values set up early as well -- as rematerialisable constants the compiler sets them twice
    // on the critical path (once around and once inside the wave's "does any lane have a partner" branch)
    NbrList<K, ASC> list0;
    list0.init(dii, agent);
    float sum0_all = 0.f, sum0_msk = 0.f;
    int ncoll0 = 0;
    if (PIN) {
        asm volatile("" : "+v"(thr), "+v"(log2_dhat), "+v"(dii), "+v"(in_range0), "+v"(zrow_a), "+v"(nrow_a), "+v"(copy_a),
                          "+v"(self_bit));
        list0.pin();                                         // (fused rollout: once per launch instead of twice per step)
        asm volatile("" : "+v"(sum0_all), "+v"(sum0_msk), "+v"(ncoll0));
        __builtin_amdgcn_sched_barrier(0);                   // nothing of the above sinks behind the first use of the state
    }

    // FAR with Delta_j >= dhat_i possible: how many partners j != i a distance clipped to dhat_i leaves inside their
    // Delta mask (position-independent; the far tail of every step uses it).  Non-uniform envs count it in the first
    // step, once the (Delta_j, l_j) table in LDS is visible.
    int far_total = (FAR && a.far_inm && uniform && a.dhat_u <= a.delta_u) ? N - 1 : 0;

    // per-step outputs of the fused rollout: running bases, advanced by one step's worth at the end of every step (two
    // scalar adds each; formed as base + step * stride they cost a 64-bit multiply chain and a 64-bit vector add per store)
    g_f32 *p_reward = o_reward, *p_true = o_true;
    g_u32 *p_gz = o_gz, *p_gn = o_gn;
    g_i32 *p_ncoll = o_ncoll;
    g_u8 *p_done = o_done;

    // Fused rollout: everything loaded ahead of the loop is waited for HERE, once.  Left pending, the compiler's wait
    // for it sits at its first use INSIDE the loop -- a vmcnt(0) in front of the rewards and a vmcnt(1) at the loop top
    // that on every later step wait for the step's own output stores and the next action's prefetch instead (gfx950
    // counts loads and stores in one in-order counter)
    // kSym64: NOT under the run-time `rand_act` of the episode-layer instances -- with the pin on one side of a branch
    // only, the other side's pending loads reach the loop header and the in-loop waits are back for BOTH sides: the fused
    // C3 rollout with pool actions and the episode layer ran 3.10-3.20 us per step against 2.89-2.95 with the pin on both
    // (round 4).  The workgroup-per-env geometries keep the one-sided pin: pinned on both sides the C5 shard's rollout
    // with pool actions and the episode layer goes from 2.46 to 2.72 us per step (same A/B, profiles/r4_abtest_rollout_pin.log)
    // (a plain `s_waitcnt vmcnt(0)` builtin on the rand_act side instead of the operand pin does not clear the
    // compiler's view of the pending loads: no gain)
    if (is_rollout(MODE) && (SYM || !rand_act))
        asm volatile("" : : "v"(xi), "v"(yi), "v"(u0.x), "v"(u0.y), "v"(xFx), "v"(xFy), "v"(xLx), "v"(xLy), "v"(dhat), "v"(delta_i),
                     "v"(li), "v"(tcur), "v"(epi), "v"(accw.x), "v"(accw.y), "v"(accw.z), "v"(accw.w));
    if (MODE == kObserve && fresh) {
        // @phase reset_draw
        // ---- env.reset() in ONE launch (drone_env.py:98-102, 171-212): N distinct lattice nodes per env by the rule of
        // reset_kernel / the in-kernel reset of the episode layer (an unsettled agent settles on its proposal iff no settled
        // agent holds that node and no lower-index agent proposed it this round; open-addressing table in LDS, keys compared
        // exactly: node ids bit-identical to oracle_reset), then the hot path below observes the new state straight from
        // registers / LDS -- no second launch, no round trip of the state through HBM (round 6: C3 12.5 us as two launches)
        const int nt = a.samp_tbl;
        int2 *tbl = reinterpret_cast<int2 *>(reinterpret_cast<float *>(smem + a.lds_tail) + 2 * ((nwaves + 1) & ~1)) + (size_t)slot * nt;
        const bool rs = valid;
        int node = -1;
        uint32_t round = 0;
        bool more;
        // The table is cleared ONCE and kept from round to round: a winner turns its entry into a blocker (owner -1), so at the top
        // of every round the table holds exactly the settled agents' nodes -- what reset_kernel rebuilds by re-entering all of them
        // after a clear -- and a later round costs the unsettled lanes' own probes only.  (The launch ends with its slowest wave: one
        // env in eight needs a second round at C3, one in two thousand a third, and with the full re-entry each cost the wave as
        // much as the first: trace of round 6, profiles/r6_trace_reset.log.)
        if (rs)
            for (int o = agent; o < nt; o += N) tbl[o] = make_int2(-1, 0x7fffffff);
        group_sync<WL>();
        do {
            const bool open = rs && node < 0;                 // unsettled: proposes this round
            int prop = -1, h = 0, old = -1;
            if (open) {
                // (the key made opaque HERE: as a loop invariant its ten-round schedule -- 18 scalar registers -- is hoisted out
                // of the sampling loop and pushes the kernel's pinned output bases into v_writelane spills around it)
                uint32_t rk0 = a.key0, rk1 = a.key1;
                asm volatile("" : "+s"(rk0), "+s"(rk1));
                prop = (int)__umulhi(philox4x32_10_word0((uint32_t)agent, round, gid, epi, rk0, rk1), a.lat_M);
                h = (int)(((uint32_t)prop * 0x9E3779B1u) >> a.samp_shift);
                for (;;) {                                                // linear probing, load factor <= 1/2
                    old = atomicCAS(&tbl[h].x, -1, prop);
                    if (old == -1 || old == prop) break;
                    h = (h + 1) & (nt - 1);
                }
            }
            // wave-local geometries, first round: when no lane of the wave met its own node in the table, all proposals of the
            // wave's envs are distinct and nothing is settled yet -- everybody wins, without the ownership exchange
            if (WL && round == 0 && __builtin_amdgcn_ballot_w64(open && old == prop) == 0ull) {
                if (open) node = prop;
                more = false;
                break;
            }
            if (open) atomicMin(&tbl[h].y, agent);            // owner = the lowest proposer (a blocker's -1 stays)
            group_sync<WL>();
            if (open && tbl[h].y == agent) { node = prop; tbl[h].y = -1; }   // settled: the entry blocks the rounds to come
            const bool left = rs && node < 0;
            more = WL ? (__builtin_amdgcn_ballot_w64(left) != 0ull) : (__syncthreads_or(left ? 1 : 0) != 0);
            if (WL) group_sync<true>();                       // this round's reads before the next round's atomics
            ++round;
        } while (more && round < (1u << 20));
        if (rs) {
            if (node >= 0) {                              // (an agent still unsettled after 2^20 rounds keeps the origin)
                const int idx = node / a.div_y, jdx = node - idx * a.div_y;
                xi = (float)idx * a.pitch; yi = (float)jdx * a.pitch;         // drone_env.py:196-205
            }
            vxi = 0.f; vyi = 0.f;                                             // :189
            st_g2(o_pos + 2 * lane, xi, yi);
            st_g2(o_vel + 2 * lane, 0.f, 0.f);
            if (a.node_out) a.node_out[wga0 + lane] = node;
            TRACE_MARK(1);                                // (trace builds: the draw is done)
            if (agent == 0) {
                a.t[env] = 0;                                                 // :100
                a.episode[env] = (int)(epi + 1u);        // (every wave of the env read the old value ahead of the first barrier)
                if (a.acc != nullptr) {                                       // train_problem.py:118-121: log, then reset
                    double *rec = a.acc + 8 * (size_t)env;
                    int *reci = reinterpret_cast<int *>(rec);
                    if (reci[5] > 0) {                                        // ep_len: an episode was in progress
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
    float2 unext = make_float2(0.f, 0.f);                    // fused rollout: the next step's action, in flight
    // @phase integrate
    for (int step = 0; step < nsteps; ++step) {
        const size_t so = (is_rollout(MODE)) ? (size_t)step * step_agents : 0;   // output offset (agents)
        const float *velsrc = (MODE == kObserve) ? a.vel : a.act + 2 * so;       // v of other agents
        if (rand_act) {
            // RandomAgent.forward (SAC_agents.py:9-22) drawn in place: one Philox block serves the steps t and t + 1
            // of an env (words 0,1 / 2,3); it is recomputed when t is even or the block is not the env's current one
            const bool need = valid && ((tcur & 1) == 0 || step == 0);
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                uint32_t fresh[4];
                philox4x32_10((uint32_t)agent, gid, (uint32_t)tcur >> 1, epi, a.key0 ^ kRandActKey, a.key1, fresh);
#pragma unroll
                for (int w = 0; w < 4; ++w) rnd[w] = need ? fresh[w] : rnd[w];
            }
            const bool odd = (tcur & 1) != 0;
            u0 = make_float2(unit_action(odd ? rnd[2] : rnd[0]), unit_action(odd ? rnd[3] : rnd[1]));
            if (valid && a.act_out != nullptr)
                st_out2(a.act_out + 2 * (so + wga0 + lane), u0.x, u0.y);
        }
        // fused rollout: the next step's action is prefetched under this step's work.  Requested by EVERY lane (a lane
        // without an agent repeats lane 0's address; a wave without agents reads the pool's first bytes): under `valid`
        // the value would be merged into u0 where the masked region ends, i.e. waited for six instructions later --
        // together with every output store of the previous step (N > 64: 2 of 3.6 us per step at the C5 shard)
        // (kept in registers of its own until it has arrived: merged into u0 right away, a half of the 8-byte load that
        // the allocator places elsewhere is copied -- and waited for -- right behind the request)
        // (the packed geometry keeps the round-2 form, prefetch under `valid` straight into u0: at its 128-register cap
        // the separate pair is itself copied and waited for at once -- C2 +5 %)
        constexpr bool PREFETCH_SEP = is_rollout(MODE) && GEO != kPacked;
        if (PREFETCH_SEP && !rand_act && step > 0) u0 = unext;
        const float2 u = u0;
        if (PREFETCH_SEP && !rand_act && step + 1 < nsteps) {
            const float2 *nxt = reinterpret_cast<const float2 *>(a.act) + (nval > 0 ? so + step_agents + wga0 : 0);
            unext = nxt[valid ? lane : 0u];
        }
        if (is_rollout(MODE) && !PREFETCH_SEP && !rand_act && valid && step + 1 < nsteps)
            u0 = (reinterpret_cast<const float2 *>(a.act) + so + step_agents + wga0)[lane];
        if (valid) {
            if (MODE != kObserve) {
                xi = fmaf(a.dt, u.x, xi);                     // drone_env.py:235
                yi = fmaf(a.dt, u.y, yi);
                vxi = u.x; vyi = u.y;                         // drone_env.py:238
            }
            spos_env[agent] = make_float2(xi, yi);
            if (FAR && a.stage5) { f32x2 v; v.x = vxi; v.y = vyi; *(lds_f32x2 *)(uintptr_t)(svel_a + 8u * (unsigned)agent) = v; }
            if (!use_bucket && !SYM) {                        // relative partner windows (dup index agent + r);
                spos_env[agent + N] = make_float2(xi, yi);    // kSym64 writes them only when its fallback runs
                spos_env[stride + agent + 1] = make_float2(xi, yi);
                spos_env[stride + agent + N + 1] = make_float2(xi, yi);
            }
        }
        // @phase filter_generic
        // ---- generic bucket filter, part 1: every agent ORs its bit into the mask of its x cell and of its y cell
        int bcx = 0, bcy = 0;
        if (use_bucket) {
            if (WL) { for (int o = lane; o < a.P * 2 * kCells; o += kWave) sbt_all[(size_t)wave * a.P * 2 * kCells + o] = 0ull; }
            else if (!EARLY_TABLES) { for (int o = tid; o < 2 * NCELL * W; o += blockDim.x) sbt_all[o] = 0ull; }
            bcx = (int)__builtin_floorf(xi * inv_cell) & (NCELL - 1);
            bcy = (int)__builtin_floorf(yi * inv_cell) & (NCELL - 1);
        }
        if (CACHED_B) {                                      // has any agent of the env left its skin/2 disk (or no list yet)?
            const float mx = xi - refx, my = yi - refy;
            const bool mv = valid && !(fmaf(my, my, mx * mx) <= moved2);
            if (__builtin_amdgcn_ballot_w64(mv) != 0ull && lane == 0) sred[2] = 1;
        }
        if (!(MODE == kObserve && fresh)) TRACE_COARSE(1);
        if (!EARLY_TABLES) group_sync<WL>();                 // (EARLY_TABLES: taken ahead of the loads' return)
        TRACE_COARSE(2);
        // CACHED_B: workgroup-uniform (every thread reads the same word; agent 0 clears it behind the verdict barrier)
        const bool rebuild = CACHED_B ? (__builtin_amdgcn_readfirstlane(sred[2]) != 0) : true;
        if (FAR && step == 0 && a.far_inm && !uniform && valid) {
#pragma nounroll
            for (int j = 0; j < N; ++j) far_total += (j != agent && dhat <= sconst[j].x) ? 1 : 0;
        }
        if (use_bucket && rebuild) {
            if (valid) {
                atomicOr(&sbt[(agent >> 6) * NCELL + bcx], 1ull << (agent & 63));            // [axis][word][cell]:
                atomicOr(&sbt[(W + (agent >> 6)) * NCELL + bcy], 1ull << (agent & 63));      // lanes spread over banks
            }
            group_sync<WL>();
        }
        TRACE_FINE(1);                                       // (-DDRONESIM_TRACE_FINE: cell tables built)

        // @phase pass2_init
        float zrx[K + 1], zry[K + 1];
        int nbv[K + 1];
        float s_all = sum0_all, s_msk = sum0_msk;
        int ncoll = ncoll0;
        NbrList<K, ASC> list = list0;                         // holds the self entry
        int in_range = in_range0;                             // :346, minus itself
        // FAR: the partners pass 2 has visited (absolute agent index, one bit each), and how many of them a distance
        // clipped to dhat_i would leave inside their Delta mask -- what the dense far tail below subtracts
        unsigned long long vis[FAR ? WMAX : 1];
#pragma unroll
        for (int w = 0; w < (FAR ? WMAX : 1); ++w) vis[w] = 0ull;
        int vis_inm = 0;

        // @phase pass2_visit
        // pass 2 body: the pair (this agent, partner at index jdup of the doubled position array)
        auto visit = [&](int jdup, auto defer, auto uni) __attribute__((always_inline)) {
            const float2 pj = spos_env[jdup];
            const int j = jdup - ((jdup >= N) ? N : 0);
            // (Delta_j, l_j): `uni` = known at the call site to be the kernel-argument scalars (the hot walk is written
            // out once per case: a run-time choice inside the loop costs two register copies and a branch per partner)
            const float2 cj = (decltype(uni)::value || uni_args) ? make_float2(a.delta_u, a.radius_u) : sconst[j];
            const float dx = xi - pj.x, dy = yi - pj.y;
            const float d2 = fmaf(dy, dy, dx * dx);
            if (CACHED && !(d2 < thr)) return;                                // listed but currently far
            const PairTerms<float> pt = pair_terms<float>(d2, li, cj.y, dhat, log2_dhat, cj.x);
            s_all += pt.lg;                                                   // :283
            s_msk += pt.inm ? pt.lg : 0.0f;                                   // :282
            ncoll += pt.coll ? 1 : 0;                                         // :284
            in_range += pt.inm ? 1 : 0;
            if (FAR) vis_inm += (dhat <= cj.x) ? 1 : 0;
            list.template insert<decltype(defer)::value>(pt.d, j, dii);      // :338
        };

        // @phase pair_parallel
        // ---- pair-parallel phase, part 1: compaction.  `up` = this lane's partners j > lane (candidates, or listed partners in
        // the fused rollout).  Lane i's pairs take the slots [off_i, off_i + cnt_i) of the list (exclusive prefix sum of the
        // counts over the wave), in ascending j: the list is sorted by (i, j).  Returns the number of pairs; the list is only
        // written when it fits the wave's 64 pair lanes (the caller falls back to the per-lane walk otherwise).
        auto pp_compact = [&](unsigned long long up) __attribute__((always_inline)) -> int {
            const int cnt = __builtin_popcountll(up);
            const int incl = wave_incl_scan64(cnt);
            const int T = __builtin_amdgcn_readlane(incl, 63);
            pp_cnt = cnt; pp_off = incl - cnt;
            if (T <= kWave) {
                unsigned wa = pp_pairs_a + 2u * (unsigned)pp_off;
                while (up) {
                    const unsigned j = (unsigned)__builtin_ctzll(up);
                    up &= up - 1ull;
                    *(__attribute__((address_space(3))) unsigned short *)(uintptr_t)wa = (unsigned short)(lane | (j << 8));
                    wa += 2u;
                }
            }
            return T;
        };
        // ---- parts 2 and 3.  Pair pass: lane p < T holds pair p = (i, j), i < j, and evaluates it ONCE: with uniform
        // (dhat, Delta, l) the terms of (i, j) and (j, i) are the same numbers (d^2 is formed from squares, the radii commute).
        // A pair inside the early-out radius leaves (d, log term, pair) in its result slot and raises bit p in row j's inbox;
        // row i finds its own pairs through the ballot of the verdicts and its segment of the list.  Fold: a row's partners in
        // ascending index = the set bits of inbox | own segment in ascending slot order (pairs (j, i), j < i, sit in earlier
        // segments, ordered by j; the row's own pairs follow, ordered by j > i) -- the additions and list insertions of the
        // per-lane walk in the same order, without its sqrt / log / compares.
        auto pair_phase = [&](int T, int off, int cnt) __attribute__((always_inline)) {
            typedef __attribute__((address_space(3))) unsigned short lds_u16;
            if (is_rollout(MODE)) pp_in[lane] = 0ull;         // (step / observe: zeroed ahead of the loads' return)
            const unsigned pr = CACHED ? pp_pr : *(const lds_u16 *)(uintptr_t)(pp_pairs_a + 2u * lane);
            const unsigned pi = pr & 63u, pj = (pr >> 8) & 63u;   // (lanes >= T read stale entries: in range, masked below)
            const float2 a0 = spos_env[pi], a1 = spos_env[pj];
            const float dx = a0.x - a1.x, dy = a0.y - a1.y;
            const float d2 = fmaf(dy, dy, dx * dx);
            const bool nr = (int)lane < T && d2 < thr;
            const unsigned long long nb = __builtin_amdgcn_ballot_w64(nr);
            if (nr) {
                const PairTerms<float> pt = pair_terms<float>(d2, a.radius_u, a.radius_u, dhat, log2_dhat, a.delta_u);
                u32x4 r; r.x = __float_as_uint(pt.d); r.y = __float_as_uint(pt.lg); r.z = pr; r.w = 0u;
                *(lds_u32x4 *)(uintptr_t)(pp_res_a + 16u * lane) = r;
                atomicOr(&pp_in[pj], 1ull << lane);
            }
            group_sync<true>();
            const unsigned long long seg = ((1ull << cnt) - 1ull) << (off & 63);   // (cnt <= kBucketMax; off = 64 only with cnt = 0)
            const unsigned long long mine = pp_in[lane] | (nb & seg);
            auto fold = [&](auto defer) __attribute__((always_inline)) {
                unsigned long long m = mine;
                while (m) {
                    const int p = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const u32x4 r = *(const lds_u32x4 *)(uintptr_t)(pp_res_a + 16u * (unsigned)p);
                    const float d = __uint_as_float(r.x), lg = __uint_as_float(r.y);
                    const int j = (int)((p < off ? r.z : (r.z >> 8)) & 63u);          // the OTHER end of the pair
                    const bool inm = d <= a.delta_u;                                  // :328
                    s_all += lg;                                                      // :283
                    s_msk += inm ? lg : 0.0f;                                         // :282
                    ncoll += (d < 0.0f) ? 1 : 0;                                      // :284, :327
                    in_range += inm ? 1 : 0;
                    list.template insert<decltype(defer)::value>(d, j, dii);         // :338
                }
            };
            fold(Defer{});
            if (__builtin_expect(list_degenerate(list), 0)) {
                list.init(dii, agent);
                in_range = in_range0;
                s_all = 0.f; s_msk = 0.f; ncoll = 0;
                fold(NoDefer{});
            }
        };

        // @phase filter_generic2
        if (valid && use_bucket) {
            // ---- generic bucket filter, part 2.  Cells are at least reach_max wide, so every partner inside this
            // agent's radius sits in its cell or a neighbouring one on BOTH axes (cell numbers are hashed mod 64:
            // far-away cells alias, which only adds candidates).  candidates = (3 x masks) & (3 y masks), exact
            // test per candidate, then pass 2 in ascending agent order.  A wave that finds one of its agents
            // crowded tests all partners instead (broadcast reads, same order, same verdicts).
            unsigned long long pool[WMAX];
            int npool = 0;
            const int cxm = (bcx - 1) & (NCELL - 1), cxp = (bcx + 1) & (NCELL - 1);
            const int cym = (bcy - 1) & (NCELL - 1), cyp = (bcy + 1) & (NCELL - 1);
            const float thr_t = CACHED_B ? thr_list : thr;   // CACHED_B: the test below builds the LIST; pass 2 drops listed-but-far pairs
            if (CACHED_B && !rebuild) {
#pragma unroll
                for (int w = 0; w < WMAX; ++w) pool[w] = candw[w];
            } else {
#pragma unroll
            for (int w = 0; w < WMAX; ++w) {
                pool[w] = 0ull;
                if (w < W) {
                    const unsigned long long *tx = sbt + w * NCELL, *ty = sbt + (W + w) * NCELL;
                    unsigned long long m = (tx[cxm] | tx[bcx] | tx[cxp]) & (ty[cym] | ty[bcy] | ty[cyp]);
                    if (w == (agent >> 6)) m &= ~(1ull << (agent & 63));
                    pool[w] = m;
                    npool += __builtin_popcountll(m);
                }
            }
            const bool crowded = __builtin_expect(__builtin_amdgcn_ballot_w64(npool > kBucketMax) != 0ull, 0);
            // the first candidate of every word is tested at once (their position reads share one LDS round trip): a
            // sparse env has about one candidate per lane and word at most, and with two waves per SIMD nothing hides
            // the dependent trip per candidate -- C5: 1.2 of the wave's 4.1 us went into one-at-a-time tests
            unsigned long long first[WMAX];
#pragma unroll
            for (int w = 0; w < WMAX; ++w) first[w] = 0ull;
            if (B256 && !crowded) {
                float2 pf[WMAX];
                int uf[WMAX];
#pragma unroll
                for (int w = 0; w < WMAX; ++w) {
                    uf[w] = pool[w] ? __builtin_ctzll(pool[w]) : 0;
                    pf[w] = spos_env[pool[w] ? 64 * w + uf[w] : 0];
                }
#pragma unroll
                for (int w = 0; w < WMAX; ++w) {
                    const float dx = xi - pf[w].x, dy = yi - pf[w].y;
                    if (pool[w] != 0ull && fmaf(dy, dy, dx * dx) < thr_t) first[w] = 1ull << uf[w];
                    pool[w] &= pool[w] - 1ull;
                }
            }
            if (B256 && !crowded) {
                // the candidates BEHIND every word's first one: ONE loop over all words, every lane taking its own lowest
                // remaining candidate per trip (the shape of the pass-2 walk).  One loop per word (through round 3) cost the
                // wave a dependent LDS round trip per word in which ANY lane had a second candidate -- nearly every word:
                // 1084 of a C5 wave's 7700 cycles (per-wave stamps, profiles/r4_trace_fine_c5.log)
                unsigned long long left = 0ull;
#pragma unroll
                for (int w = 0; w < WMAX; ++w) left |= pool[w];
                while (left != 0ull) {
                    unsigned long long hs = pool[WMAX - 1];
                    int ws = WMAX - 1;
#pragma unroll
                    for (int w = WMAX - 2; w >= 0; --w) { if (pool[w] != 0ull) { hs = pool[w]; ws = w; } }
                    const int u = __builtin_ctzll(hs);
                    hs &= hs - 1ull;
                    const float2 pj = spos_env[64 * ws + u];
                    const float dx = xi - pj.x, dy = yi - pj.y;
                    const bool hit = fmaf(dy, dy, dx * dx) < thr_t;
                    left = 0ull;
#pragma unroll
                    for (int w = 0; w < WMAX; ++w) {
                        if (w == ws) { pool[w] = hs; if (hit) first[w] |= 1ull << u; }
                        left |= pool[w];
                    }
                }
#pragma unroll
                for (int w = 0; w < WMAX; ++w) pool[w] = first[w];       // the verdicts replace the candidates
            } else {
#pragma unroll
            for (int w = 0; w < WMAX; ++w) {
                if (w < W) {
                    unsigned long long hits = first[w];
                    if (crowded) {
                        const int jn = min(64, N - 64 * w);
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const int cnt = jn - c4 * kChunk;
                            if (cnt > 0) {
                                const float4 *pp = reinterpret_cast<const float4 *>(spos_env + 64 * w + c4 * kChunk);
                                unsigned m = 0u;
#pragma unroll
                                for (int u = 0; u < kChunk / 2; ++u) {
                                    const float4 v = pp[u];
                                    const float dx0 = xi - v.x, dy0 = yi - v.y, dx1 = xi - v.z, dy1 = yi - v.w;
                                    m |= (fmaf(dy0, dy0, dx0 * dx0) < thr_t ? 1u : 0u) << (2 * u);
                                    m |= (fmaf(dy1, dy1, dx1 * dx1) < thr_t ? 1u : 0u) << (2 * u + 1);
                                }
                                if (cnt < kChunk) m &= (1u << cnt) - 1u;
                                hits |= (unsigned long long)m << (c4 * kChunk);
                            }
                        }
                        if (w == (agent >> 6)) hits &= ~(1ull << (agent & 63));
                    } else if (__builtin_amdgcn_ballot_w64(pool[w] != 0ull) != 0ull) {
                        unsigned long long m = pool[w];
                        while (m) {
                            const int u = __builtin_ctzll(m);
                            m &= m - 1ull;
                            const float2 pj = spos_env[64 * w + u];
                            const float dx = xi - pj.x, dy = yi - pj.y;
                            if (fmaf(dy, dy, dx * dx) < thr_t) hits |= 1ull << u;
                        }
                    }
                    pool[w] = hits;                          // the verdicts replace the candidates
                }
            }
            }
            if (CACHED_B) {                                  // keep the list and where it was taken
#pragma unroll
                for (int w = 0; w < WMAX; ++w) candw[w] = pool[w];
                refx = xi; refy = yi;
            }
            }   // rebuild
            TRACE_FINE(2);                                   // (-DDRONESIM_TRACE_FINE: candidates tested)
            // pass 2 over the verdicts, ascending agent order.  Workgroup-per-env geometries (ascending-order list):
            // the hot walk defers the general insertion and is written out for the uniform-(Delta, l) case, like kSym64's
            auto walk = [&](auto defer, auto uni) __attribute__((always_inline)) {
                if (B256) {
                    // ONE loop over the verdicts of all words (every lane takes its own lowest partner per trip): a trip
                    // costs the wave a dependent LDS read and ~60 VALU whatever the number of lanes that take part, and
                    // the few partners of a sparse env are spread over the words
                    unsigned long long h[WMAX];
#pragma unroll
                    for (int w = 0; w < WMAX; ++w) h[w] = pool[w];
                    unsigned long long left = 0ull;
#pragma unroll
                    for (int w = 0; w < WMAX; ++w) left |= h[w];
                    while (left != 0ull) {
                        unsigned long long hs = h[WMAX - 1];
                        int ws = WMAX - 1;
#pragma unroll
                        for (int w = WMAX - 2; w >= 0; --w) { if (h[w] != 0ull) { hs = h[w]; ws = w; } }
                        const int u = __builtin_ctzll(hs);
                        hs &= hs - 1ull;
                        left = 0ull;
#pragma unroll
                        for (int w = 0; w < WMAX; ++w) { if (w == ws) h[w] = hs; left |= h[w]; }
                        visit(64 * ws + u, defer, uni);
                    }
                    return;
                }
#pragma unroll
                for (int w = 0; w < WMAX; ++w) {
                    if (w < W) {
                        unsigned long long hits = pool[w];
                        while (hits) {
                            const int u = __builtin_ctzll(hits);
                            hits &= hits - 1ull;
                            visit(64 * w + u, defer, uni);
                        }
                    }
                }
            };
            if (FAR) {
#pragma unroll
                for (int w = 0; w < WMAX; ++w) vis[w] = (w < W) ? pool[w] : 0ull;
            }
            if (ASC) {
                if (uni_args) walk(Defer{}, UniArgs{}); else walk(Defer{}, UniRuntime{});
                if (__builtin_expect(list_degenerate(list), 0)) {
                    list.init(dii, agent);
                    in_range = in_range0;
                    s_all = 0.f; s_msk = 0.f; ncoll = 0;
                    walk(NoDefer{}, UniRuntime{});
                }
            } else {
                walk(NoDefer{}, UniRuntime{});
            }
        }
        // @phase filter_scan
        if (valid && !use_bucket) {
            const int rmax = N - 1;
            for (int r0 = 1; r0 <= rmax; r0 += 64) {
                // ---- pass 1: far filter, 16 partners in flight.  Result: one bit per partner to revisit.
                unsigned long long near = 0ull;
                bool pp_now = false;                          // wave-uniform: this step's pairs go through the pair-parallel phase
                const int left = rmax - r0 + 1;
                if (FAR && N <= kFarAllMaxN) {
                    // a handful of agents: every partner goes through pass 2 (the filter and the far tail cost more than
                    // the few far pairs they would save: N = 5 x 1024 envs 4.0 us this way, 4.66 us filtered)
                    near = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
                } else if (SYM) {
                    // bit u < 32: partner i+1+u ("forward");  bit 32+u: partner i-1-u ("backward", u < 31)
                    bool rebuild = true;
                    if (CACHED) {
                        const float mx = xi - refx, my = yi - refy;
                        rebuild = __builtin_amdgcn_ballot_w64(!(fmaf(my, my, mx * mx) <= moved2)) != 0ull;
                        near = cand;
                        pp_now = PP && pp_T >= 0;
                    }
                    if (rebuild) {
                    pp_now = false;
                    // @phase filter_sym_bucket
                    // (a) bucket filter: cells of width >= the list radius along x and along y; a partner can
                    //     only be inside the radius if it sits in this agent's cell or a neighbouring one on
                    //     BOTH axes.  Each lane ORs its lane bit into the mask of its x cell and of its y cell
                    //     (LDS atomics), then reads the three masks around its own cell per axis:
                    //     candidates = (x masks) & (y masks).  Coordinates beyond the 64 cells clamp to the end
                    //     cells, which only ever adds candidates.  ~35 instructions instead of 32 offsets x 5+.
                    const unsigned long long self = self_bit;
                    if (CACHED || is_rollout(MODE)) { sbx[lane] = 0ull; sby[lane] = 0ull; }   // (step / observe: zeroed ahead of the loads' return)
                    const int cx = (int)fminf(fmaxf(fmaf(xi, inv_cell, 1.0f), 1.0f), 62.0f);
                    const int cy = (int)fminf(fmaxf(fmaf(yi, inv_cell, 1.0f), 1.0f), 62.0f);
                    group_sync<true>();
                    atomicOr(&sbx[cx], self);
                    atomicOr(&sby[cy], self);
                    group_sync<true>();
                    unsigned long long pool = (sbx[cx - 1] | sbx[cx] | sbx[cx + 1]) &
                                              (sby[cy - 1] | sby[cy] | sby[cy + 1]) & ~self;
                    unsigned long long hits = 0ull;          // bit j: agent j is inside the list radius
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(__builtin_popcountll(pool) > kBucketMax) == 0ull, 1)) {
                        // single step: the candidate pairs themselves are compacted, the pair pass does the exact test
                        if (PP && !CACHED) {
                            pp_T = pp_compact(pool & above);
                            pp_now = pp_T <= kWave;            // (more than 64 candidate pairs: the per-lane walk below)
                        }
                        // (b) exact test of the few candidates
                        // (requesting 2 or 4 candidates' positions per LDS round trip was measured: +0.04 / +0.08 us per launch --
                        // this part of a wave is issue-bound, the other three waves of the SIMD cover the round trips)
                        if (!pp_now)
                        while (pool) {
                            const int j = __builtin_ctzll(pool);
                            pool &= pool - 1ull;
                            const float2 pj = spos_env[j];
                            const float dx = xi - pj.x, dy = yi - pj.y;
                            if (fmaf(dy, dy, dx * dx) < thr_list) hits |= 1ull << j;
                        }
                        // fused rollout: the LISTED pairs (inside reach + skin) are compacted once per rebuild; every step's
                        // pair pass drops the listed-but-currently-far ones by its exact test
                        if (PP && CACHED) {
                            pp_T = pp_compact(hits & above);
                            if (pp_T > kWave) pp_T = -1;
                            pp_now = pp_T >= 0;
                            if (pp_now) pp_pr = *(const __attribute__((address_space(3))) unsigned short *)(uintptr_t)(pp_pairs_a + 2u * lane);
                        }
                    } else {
                    if (PP && CACHED) pp_T = -1;              // (a crowded env's list stays in the per-lane masks)
                    // @phase filter_sym_crowded
                    // (c) crowded env: every unordered pair once -- lane i tests partners i+1..i+32 and the
                    //     verdict reaches the other end as a rotated ballot.  The doubled / shifted copies of
                    //     the positions that those windows read are made here, on the rare path only.
                    spos_env[agent + N] = make_float2(xi, yi);
                    spos_env[stride + agent + 1] = make_float2(xi, yi);
                    spos_env[stride + agent + N + 1] = make_float2(xi, yi);
                    group_sync<true>();
                    unsigned mf = 0u, mb = 0u;
                    // (8 partners in flight, not 16: this rare path must not set the register budget of the hot one --
                    // with 16 the 64-register step kernels of the episode layer spilled here)
                    constexpr int kSymChunk = 8;
#pragma unroll
                    for (int c2 = 0; c2 < 32 / kSymChunk; ++c2) {
                        const float4 *pp = reinterpret_cast<const float4 *>(pwin + c2 * kSymChunk);   // dup index agent+1+8*c2
                        float2 pj[kSymChunk];
#pragma unroll
                        for (int u = 0; u < kSymChunk / 2; ++u) {         // ds_read_b128: two partners per read
                            const float4 v = pp[u];
                            pj[2 * u] = make_float2(v.x, v.y); pj[2 * u + 1] = make_float2(v.z, v.w);
                        }
#pragma unroll
                        for (int u = 0; u < kSymChunk; ++u) {
                            const int r = 1 + c2 * kSymChunk + u;             // 1..32
                            const float dx = xi - pj[u].x, dy = yi - pj[u].y;
                            const float d2 = fmaf(dy, dy, dx * dx);
                            const bool f = d2 < thr_list;
                            const unsigned long long fm = __builtin_amdgcn_ballot_w64(f);
                            if (fm) {                                         // wave-uniform
                                mf |= (f ? 1u : 0u) << (r - 1);
                                if (r < 32) {                                 // r = 32: both ends see it as forward
                                    const unsigned long long bm = (fm << r) | (fm >> (64 - r));   // lane i -> lane i+r
                                    mb |= (__builtin_amdgcn_inverse_ballot_w64(bm) ? 1u : 0u) << (r - 1);
                                }
                            }
                        }
                    }
                    // offsets -> agent indices: forward bit u is agent i+1+u, backward bit u is agent i-1-u
                    const unsigned long long fw = (unsigned long long)mf, bw = (unsigned long long)__builtin_bitreverse32(mb);
                    const unsigned sf = (lane + 1) & 63u, sb = (lane + 32) & 63u;
                    hits = ((fw << sf) | (sf ? fw >> (64 - sf) : 0ull)) | ((bw << sb) | (sb ? bw >> (64 - sb) : 0ull));
                    }
                    near = hits;
                    if (CACHED) { cand = near; refx = xi; refy = yi; }
                    }
                } else {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const int cnt = left - c4 * kChunk;
                        if (cnt > 0) {
                            const float4 *pp = reinterpret_cast<const float4 *>(pwin + (r0 - 1) + c4 * kChunk);
                            float2 pj[kChunk];
#pragma unroll
                            for (int u = 0; u < kChunk / 2; ++u) {        // ds_read_b128: two partners per read
                                const float4 v = pp[u];
                                pj[2 * u] = make_float2(v.x, v.y); pj[2 * u + 1] = make_float2(v.z, v.w);
                            }
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
                // @phase pass2_walk
                // ---- pass 2: every lane walks its own surviving partners
                if (SYM) {
                    if (FAR) vis[0] = near;                   // (absolute agent indices already)
                    // hot walk with the deferred neighbour list; a wave that met a partner at or inside its agent's own
                    // entry (coincident agents, a larger partner over a smaller agent's centre) starts over with the
                    // general insertion, out of line
                    const unsigned long long near0 = near;
                    if (PP && pp_now) {
                        pair_phase(pp_T, pp_off, pp_cnt);
                        near = 0ull;
                    } else if (uni_args) {
                        while (near) {
                            const int u = __builtin_ctzll(near);
                            near &= near - 1ull;
                            visit(u, Defer{}, UniArgs{});
                        }
                    } else {
                        while (near) {
                            const int u = __builtin_ctzll(near);
                            near &= near - 1ull;
                            visit(u, Defer{}, UniRuntime{});
                        }
                    }
                    if (__builtin_expect(list_degenerate(list), 0)) {
                        list.init(dii, agent);
                        in_range = in_range0;
                        s_all = 0.f; s_msk = 0.f; ncoll = 0;
                        near = near0;
                        while (near) {
                            const int u = __builtin_ctzll(near);
                            near &= near - 1ull;
                            visit(u, NoDefer{}, UniRuntime{});
                        }
                    }
                } else {
                if (FAR) {
                    // bit u of `near` is partner (agent + r0 + u) mod N: rotate into absolute agent indices.  This scan
                    // runs for packed envs of N < kBucketMinN agents only: one chunk (r0 = 1), all shifts below 64.
                    const int sh = agent + r0, back = N - sh;                         // 1 <= sh <= N - 1 + 1
                    const unsigned long long lo = sh < 64 ? near << sh : 0ull;
                    const unsigned long long hi = back > 0 ? near >> back : near;
                    vis[0] |= (lo | hi) & ((N < 64) ? ((1ull << N) - 1ull) : ~0ull);
                }
                while (near) {
                    const int u = __builtin_ctzll(near);
                    near &= near - 1ull;
                    visit(agent + r0 + u, NoDefer{}, UniRuntime{});
                }
                }
            }
        }
        // @phase far_tail
        if (FAR && valid && !(!use_bucket && N <= kFarAllMaxN)) {
            // ---- Far partners (d^2 >= reach^2: never visited above).  Their clipped distance is dhat_i EXACTLY (:318), so
            // the log term is 0, they are no collision (dhat > 0), and they are inside the column's Delta mask iff
            // dhat_i <= Delta_j -- none of which depends on where the partner is.  The row sums and the collision count are
            // therefore complete as they stand; the Delta count gets (count over all j != i, position-independent) minus
            // (the visited partners' share of it); and since all far partners tie at dhat_i, the stable argsort (:338)
            // lists them by index: the K lowest-index unvisited agents are offered to the list (keyed insertion: a
            // visited partner that was clipped to exactly dhat_i ties with them and sorts by index like them).
            // Round 2 sent every ordered pair of such envs (c = 5, or deltas=None) through sqrt / log / the insertion.
            if (a.far_inm) in_range += far_total - vis_inm;
            unsigned long long fr[WMAX];
#pragma unroll
            for (int w = 0; w < WMAX; ++w) {
                const int nw = N - 64 * w;                                        // agents in this word
                const unsigned long long live = nw >= 64 ? ~0ull : (nw > 0 ? (1ull << nw) - 1ull : 0ull);
                fr[w] = ~vis[w] & live;
                if (w == (agent >> 6)) fr[w] &= ~(1ull << (agent & 63));
            }
#pragma unroll
            for (int s = 0; s < K; ++s) {
                int j = -1;
#pragma unroll
                for (int w = 0; w < WMAX; ++w) {
                    if (j < 0 && fr[w] != 0ull) {
                        j = 64 * w + __builtin_ctzll(fr[w]);
                        fr[w] &= fr[w] - 1ull;
                    }
                }
                if (j >= 0) list.template insert<false>(dhat, j, dii);
            }
        }

        // @phase epilogue_rewards_z
        float r_out = 0.0f, tr_out = 0.0f;                    // this lane's rewards (episode bookkeeping)
        float r_env = 0.0f, tr_env = 0.0f;                    // their sums over the env, valid in its agent-0 lane
        float2 sm_blk = make_float2(0.f, 0.f);                // workgroup-per-env: this wave's partial sums
        int coll_s = 0;                                       // kSym64: the env's collisions / agents outside the goal
        unsigned long long outside_m = 0ull;                  //         disk, wave-uniform (scalar registers)
        // fused rollout: the next step's action (prefetched at the top of this step) is waited for HERE, ahead of the
        // step's first output store -- at the loop's end the same wait would also cover the stores just issued
        if (is_rollout(MODE) && GEO != kPacked) asm volatile("" : "+v"(unext.x), "+v"(unext.y));
        if (valid) {
            TRACE_MARK(3);
            // rewards (:276, :287-288)
            // x - xF with the goal in two float32 parts: the first difference is exact near the goal (Sterbenz), so the
            // offset keeps float32 RELATIVE accuracy where the ghost direction and the arrival test are sensitive to it
            const float zx = (xi - xFx) - xLx, zy = (yi - xFy) - xLy;         // :357
            const float err2 = fmaf(zy, zy, zx * zx);
            const float to_goal = k_q * err2;
            r_out = -nan_to_num_f32(fmaf(k_b, s_msk, to_goal));
            tr_out = -nan_to_num_f32(fmaf(k_b, s_all, to_goal));
            if (w_reward) st_g(p_reward + lane, r_out);
            if (w_true) st_g(p_true + lane, tr_out);
            if (SYM && EPI) {                                 // all 64 lanes are here (one env per wave).  Not under the
                                                              // run-time `has_acc`: a branch would fence the dependent
                const float2 sm = wave_sum64_pair(r_out, tr_out);   // chain off from the row arithmetic below, which fills
                r_env = sm.x; tr_env = sm.y;                  // its wait states
            } else if (!WL && EPI && nval == kWave) {
                // workgroup-per-env, a full wave (wave-uniform test: all 64 lanes are here): the same fixed-order tree, taken
                // HERE so that the row arithmetic below fills the wait states of its dependent chain (round 4; it sat
                // behind the rows, in front of the staging, where nothing overlaps it)
                sm_blk = wave_sum64_pair(r_out, tr_out);
            }

            // localized state rows + neighbour list (:344-397)
            const float gsc = __builtin_amdgcn_rsqf(err2) * delta_i * k_ghost;
            const float ghx = zx * gsc, ghy = zy * gsc;                       // :386 (NaN when on the goal)
            zrx[0] = zx; zry[0] = zy; nbv[0] = agent;
            bool have[K + 1];
            have[0] = true;
            // all K neighbour positions are requested before the first is used (one LDS round trip instead of K; an
            // entry without a neighbour reads the agent's own slot and is discarded)
            float2 pnb[K + 1];
#pragma unroll
            for (int kth = 1; kth <= K; ++kth) {
                const unsigned j = list.index(kth);
                have[kth] = j < (unsigned)N;
                pnb[kth] = spos_env[have[kth] ? j : (unsigned)agent];
            }
#pragma unroll
            for (int kth = 1; kth <= K; ++kth) {
                const unsigned j = list.index(kth);
                const bool real = kth <= in_range && have[kth];               // :362
                zrx[kth] = real ? pnb[kth].x - xi : ghx;                      // :368
                zry[kth] = real ? pnb[kth].y - yi : ghy;
                nbv[kth] = real ? (int)j : -1;
            }
            // c = 5 rows also carry (vx, vy, l) of the row's agent: the agent itself (:355), the listed neighbour or the
            // tie-ordered agent behind a ghost row (:367 / :385), NaN where the list has no agent at all
            float zvx[FAR ? K + 1 : 1], zvy[FAR ? K + 1 : 1], zvl[FAR ? K + 1 : 1];
            if (FAR && zc == 5) {
                zvx[0] = vxi; zvy[0] = vyi; zvl[0] = li;
#pragma unroll
                for (int kth = 1; kth <= K; ++kth) {
                    const unsigned j = list.index(kth);
                    float2 vj = make_float2(__builtin_nanf(""), __builtin_nanf(""));
                    float lj = __builtin_nanf("");
                    if (have[kth]) {
                        if (a.stage5) {
                            const f32x2 v = *(const lds_f32x2 *)(uintptr_t)(svel_a + 8u * j);    // (written next to the positions, same sync)
                            vj = make_float2(v.x, v.y);
                        } else if (rand_act) {                                // counter-based stream: any lane can
                            uint32_t o[4];                                    // restate any agent's action
                            philox4x32_10(j, gid, (uint32_t)(tcur >> 1), epi, a.key0 ^ kRandActKey, a.key1, o);
                            vj = (tcur & 1) ? make_float2(unit_action(o[2]), unit_action(o[3]))
                                            : make_float2(unit_action(o[0]), unit_action(o[1]));
                        } else {
                            vj = reinterpret_cast<const float2 *>(velsrc)[(size_t)env * N + j];
                        }
                        lj = uni_args ? a.radius_u : sconst[j].y;
                    }
                    zvx[kth] = vj.x; zvy[kth] = vj.y; zvl[kth] = lj;
                }
            }
            if (!staged) {                                                    // masked observe / c = 5 rows that do not fit LDS
                const int c = zc;
                float *zr = a.z + (so + wga0 + lane) * (size_t)((K + 1) * c);
                int *nb = a.nbr_idx + (so + wga0 + lane) * (size_t)(K + 1);
#pragma unroll
                for (int kth = 0; kth <= K; ++kth) {
                    nb[kth] = nbv[kth];
                    float *row = zr + kth * c;
                    row[0] = zrx[kth]; row[1] = zry[kth];
                    if (FAR && c == 5) { row[2] = zvx[kth]; row[3] = zvy[kth]; row[4] = zvl[kth]; }
                }
            } else if (FAR && zc == 5) {                                      // this lane's 5 (K+1)-word row -> staging area
                float *zrow5 = reinterpret_cast<float *>(stage_z) + lane * (5 * (K + 1));
                unsigned *nrow5 = stage_n + lane * kNRow;
#pragma unroll
                for (int kth = 0; kth <= K; ++kth) {
                    zrow5[5 * kth + 0] = zrx[kth]; zrow5[5 * kth + 1] = zry[kth];
                    zrow5[5 * kth + 2] = zvx[kth]; zrow5[5 * kth + 3] = zvy[kth]; zrow5[5 * kth + 4] = zvl[kth];
                    nrow5[kth] = (unsigned)nbv[kth];
                }
            }

            // @phase epilogue_state_done
            if (MODE != kObserve) {
                if (!is_rollout(MODE) || step == nsteps - 1) {                 // final state only
                    st_g2(o_pos + 2 * lane, xi, yi);
                    st_g2(o_vel + 2 * lane, vxi, vyi);
                }
                if (SYM) outside_m = __builtin_amdgcn_ballot_w64(!(__builtin_amdgcn_sqrtf(err2) <= k_done_radius));
                else if (!(__builtin_amdgcn_sqrtf(err2) <= k_done_radius)) atomicOr(&sred[2 * slot + 1], 1);   // :249-251
            }
            if (SYM) {
                // one env per wave: the env's collision count is a sum of ballot popcounts on the scalar unit (ballots of
                // ncoll >= 1, >= 2, ...: one or two rounds), the arrival verdict one ballot -- no LDS words, no waits
                unsigned long long m = __builtin_amdgcn_ballot_w64(ncoll > 0);
                for (int lvl = 1; m != 0ull; ++lvl) {
                    coll_s += __builtin_popcountll(m);
                    m = __builtin_amdgcn_ballot_w64(ncoll > lvl);
                }
            } else if (ncoll) atomicAdd(&sred[2 * slot], ncoll);
        }
        // episode bookkeeping: sum of this step's rewards per env, fixed order (bit-reproducible).  Workgroup-per-env
        // geometries leave one partial per wave in LDS ahead of the barrier below; wave-local geometries reduce after
        // their output stores have been issued (the reduction is a dependent chain: nothing should queue behind it)
        if (!WL && has_acc) {
            if (nval != kWave) sm_blk = wave_sum64_pair(r_out, tr_out);   // (a ragged last wave: lanes without an agent carry 0)
            if (lane == 0) { spart[2 * wave] = sm_blk.x; spart[2 * wave + 1] = sm_blk.y; }
        }
        // @phase stage_and_copy_out
        if (staged && valid && zc == 2) {                     // this lane's rows -> the wave's staging area
#pragma unroll
            for (int kth = 0; kth <= K; ++kth) {              // one base each (formed early), immediate offsets
                f32x2 v; v.x = zrx[kth]; v.y = zry[kth];
                *(lds_f32x2 *)(uintptr_t)(zrow_a + 8 * kth) = v;
            }
#pragma unroll
            for (int kth = 0; kth <= K; ++kth) *(lds_u32 *)(uintptr_t)(nrow_a + 4 * kth) = (unsigned)nbv[kth];
        }
        TRACE_MARK(4);
        // the staged rows are written and copied out by the SAME wave: a wave-level fence orders them.  The workgroup
        // barrier the env's verdict words need (n_coll / done / reward partials of all waves) comes behind the copy-out,
        // so that a wave's output stores are issued before it waits for the slower waves of its env
        group_sync<true>();
        if (WL && !SYM && has_acc) {
            r_env = segment_sum(r_out, agent, N); tr_env = segment_sum(tr_out, agent, N);
        }
        if (staged && nval > 0) {
            g_u32 *gzg = p_gz, *gng = p_gn;
            if (FAR && zc == 5) gzg = (g_u32 *)(reinterpret_cast<unsigned *>(a.z) + (so + wga0) * (size_t)(5 * (K + 1)));
            unsigned *gz = (unsigned *)gzg, *gn = (unsigned *)gng;              // (generic views for the ragged copy)
            // wave-uniform row bases: pinned in SGPRs so that the stores below address as scalar base + 32-bit lane
            // offset (the compiler otherwise builds 64-bit per-lane addresses: three v_lshl_add_u64 and two v_mad_i64_i32)
            typedef g_u32x4 gu32x4;
            gu32x4 *gz4 = (gu32x4 *)gzg, *gn4 = (gu32x4 *)gng;
            // fixed-shape copy: a full wave of agents whose rows start 16-byte aligned -- always for kSym64 (checked on
            // the host), every full wave of the workgroup-per-env geometries otherwise (wave-uniform test)
            const bool fixed = (SYM && !(FAR && zc == 5)) || BU || (BLOCKGEO && !(FAR && zc == 5) && nval == kWave &&
                                       ((reinterpret_cast<uintptr_t>(gz) | reinterpret_cast<uintptr_t>(gn)) & 15u) == 0);
            if (!kTrace && (SYM || BLOCKGEO)) asm volatile("" : "+s"(gz4), "+s"(gn4));   // (the trace build's stamps make hipcc lose the uniformity)
            if (fixed) {
                // full wave of one env, 16-byte aligned rows (checked on the host): fixed-shape copy, no loops
                // (a branch-free variant -- surplus lanes of a ragged last round repeating a live lane's 16 bytes, so
                // that the kernel's tail is one basic block -- was 0.14 us slower per launch: the stores got wider)
                // All staged words are read before the first store is issued (one LDS round trip for the whole copy-out;
                // the surplus lanes of a ragged last round read on into the cell tables behind the staging area -- kSym64:
                // the wave's own, part of its block; workgroup-per-env: the env's -- and store nothing).
                constexpr int nz = kWave * kZRow, nn = kWave * kNRow;          // words
                constexpr int rz = (nz + 4 * kWave - 1) / (4 * kWave), rn = (nn + 4 * kWave - 1) / (4 * kWave);
                u32x4 vz[rz], vn[rn];
#pragma unroll
                for (int r = 0; r < rz; ++r) vz[r] = *(const lds_u32x4 *)(uintptr_t)(copy_a + r * 16 * kWave);
#pragma unroll
                for (int r = 0; r < rn; ++r) vn[r] = *(const lds_u32x4 *)(uintptr_t)(copy_a + 4 * nz + r * 16 * kWave);
#pragma unroll
                for (int r = 0; r < rz; ++r)
                    if ((r + 1) * 4 * kWave <= nz || (int)lane * 4 < nz - r * 4 * kWave)
                        __builtin_nontemporal_store(vz[r], gz4 + r * kWave + lane);
#pragma unroll
                for (int r = 0; r < rn; ++r)
                    if ((r + 1) * 4 * kWave <= nn || (int)lane * 4 < nn - r * 4 * kWave)
                        __builtin_nontemporal_store(vn[r], gn4 + r * kWave + lane);
            } else {
                wave_copy_out(gz, stage_z, nval * zrow_w, lane);
                wave_copy_out(gn, stage_n, nval * kNRow, lane);
            }
        }
        if (!WL) group_sync<false>();
        // @phase env_outputs
        bool fin_env = false;                                 // agent 0: this env's episode ended with this step
        if (SYM) {
            // one env per wave: verdicts are wave-uniform, the record update is branch-free (lane 0 holds the two sums,
            // lane 1 the counters), and lane 0's stores are the only masked region
            const int tc = MODE != kObserve ? __builtin_amdgcn_readfirstlane(tcur) : 0;      // lane 0 = agent 0
            const bool fin = MODE != kObserve && (outside_m == 0ull || tc >= k_last_t);   // :251
            fin_env = fin;
            if (has_acc) {                                        // train_problem.py:98-100 and t_iter, every step
                const double nr = __builtin_bit_cast(double, make_uint2(accw.x, accw.y)) + (double)r_env;
                const double ntr = __builtin_bit_cast(double, make_uint2(accw.z, accw.w)) + (double)tr_env;
                const uint2 w0 = __builtin_bit_cast(uint2, nr), w1 = __builtin_bit_cast(uint2, ntr);
                const bool l0 = lane == 0;
                accw = make_uint4(l0 ? w0.x : accw.x + (unsigned)coll_s, l0 ? w0.y : accw.y + 1u,
                                  l0 ? w1.x : accw.z, l0 ? w1.y : accw.w);
            }
            if (lane == 0) {
                if (w_ncoll) *p_ncoll = coll_s;
                if (MODE != kObserve) *p_done = (uint8_t)fin;
            }
        } else
        if (valid && (agent == 0 || (has_acc && agent == 1))) {
            const int2 red = *reinterpret_cast<const int2 *>(sred + 2 * slot);   // (collisions, someone outside the goal disk)
            // up to four waves: all partial sums are requested together with the verdict words (one LDS round trip; the
            // region behind the partials belongs to the episode layer's sampling tables, so 32 bytes are always there)
            float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
            if (B256 && has_acc && agent == 0) {
                pa = reinterpret_cast<const float4 *>(spart)[0];
                pb = reinterpret_cast<const float4 *>(spart)[1];
            }
            const int coll_env = red.x;
            if (agent == 0) {
                const size_t eo = (is_rollout(MODE)) ? (size_t)step * a.E + env : (size_t)env;
                if (a.n_coll) a.n_coll[eo] = coll_env;
                bool fin = false;
                if (MODE != kObserve) {
                    fin = (red.y == 0) || (tcur >= k_last_t);                                     // :251
                    a.done[eo] = (uint8_t)fin;
                    fin_env = fin;
                }
                if (has_acc) {                                    // train_problem.py:98-99, every step
                    if (B256) {                                   // wave order, like the loop below
                        r_env += pa.x; tr_env += pa.y;
                        if (nwaves > 1) { r_env += pa.z; tr_env += pa.w; }
                        if (nwaves > 2) { r_env += pb.x; tr_env += pb.y; }
                        if (nwaves > 3) { r_env += pb.z; tr_env += pb.w; }
                    } else if (!WL) {
                        for (int w = 0; w < nwaves; ++w) { r_env += spart[2 * w]; tr_env += spart[2 * w + 1]; }
                    }
                    const double nr = __builtin_bit_cast(double, make_uint2(accw.x, accw.y)) + (double)r_env;
                    const double ntr = __builtin_bit_cast(double, make_uint2(accw.z, accw.w)) + (double)tr_env;
                    const uint2 w0 = __builtin_bit_cast(uint2, nr), w1 = __builtin_bit_cast(uint2, ntr);
                    accw = make_uint4(w0.x, w0.y, w1.x, w1.y);
                }
                *reinterpret_cast<int2 *>(sred + 2 * slot) = make_int2(0, (auto_reset && fin) ? 1 : 0);   // under auto_reset: "re-sample this env"
                if (CACHED_B) sred[2] = 0;                   // the "moved" flag: every thread has read it (barriers in between)
            } else {                                              // train_problem.py:100 and t_iter, every step
                accw.x += (unsigned)coll_env; accw.y += 1u;
            }
        }
        if (MODE != kObserve && (rand_act || agent == 0)) tcur += 1;                              // :256
        if (EPI && auto_reset) {
            // @phase auto_reset
            // ---- in-kernel reset (drone_env.py:98-102 after the `while not finished` loop, train_problem.py:132).
            // Every lane learns whether its env finished.  Finished envs retire their episode record, draw N distinct
            // lattice nodes exactly as reset_kernel does (same Philox stream, same acceptance rule) and are observed
            // again (drone_env.py:208-210) by a plain all-partner pass: this runs once per episode and env, so it is
            // written for size, not speed; its arithmetic is the hot path's own `visit`, so z / Ni are bit-identical
            // to what dronesim_reset + dronesim_observe produce.
            bool rs, any_rs;
            if (SYM) {                                        // one env per wave: agent 0 is lane 0, its verdict is the wave's
                any_rs = __builtin_amdgcn_readfirstlane((int)fin_env) != 0;
                rs = any_rs;
            } else {
                group_sync<WL>();
                const bool flag = sred[2 * slot + 1] != 0;
                rs = valid && flag;
                // one env per workgroup: every thread has just read the same word -- no second barrier to agree on it
                any_rs = WL ? (__builtin_amdgcn_ballot_w64(rs) != 0ull) : (__builtin_amdgcn_readfirstlane((int)flag) != 0);
            }
            if (__builtin_expect(any_rs, 0)) {                // once per episode and env: out of line
                // (`s_nop 13` ... `s_nop 14`: where this cold block begins and ends in the ISA, for tools/spill_sites.py --
                // which scratch accesses of an episode-layer kernel sit on the per-step path and which in here)
                asm volatile("s_nop 13");
                // This block's own kernel arguments are read through the kernel-argument segment behind an opaque pointer,
                // and its per-lane addresses are formed from an opaque copy of the lane id: read as fields of `a` / formed
                // from `lane`, the compiler hoists those scalar loads and 64-bit additions in front of the branch, i.e.
                // onto the critical path of EVERY step (six s_load and four vector instructions at C3)
                const __attribute__((address_space(4))) char *kseg =
                    (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
                unsigned lane_c = lane;
                asm volatile("" : "+s"(kseg), "+v"(lane_c));
                const __attribute__((address_space(4))) KArgs &ca =
                    *(const __attribute__((address_space(4))) KArgs *)(kseg + kKArgsOffset);
                int *const c_episode = ca.episode;
                if (!rand_act && rs)                          // the episode counter is only needed here: read it now (past
                    epi = (uint32_t)__builtin_nontemporal_load(c_episode + env);   // L1: an earlier reset of this launch wrote it)
                // (the counter is stored back -- by agent 0 -- only behind the first barrier of the sampling loop, and every
                // wave waits for its own read first: written right here, a wave that ran ahead could hand a slower wave of
                // the same env the INCREMENTED counter, i.e. another Philox stream for its 64 agents.  Round 4: found by the
                // rollout fuzz of tests/test_gpu_fuzz.py as a 1-in-5 flake of multi-wave envs, present since round 2.)
                if (!WL && !rand_act) {
                    __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0): epi has arrived
                    asm volatile("" : "+v"(epi) :: "memory");                      // ... and is a register value HERE: the load cannot
                }                                                                 // sink behind the barrier where agent 0 stores epi + 1
                if (WL && rs && agent == 0) c_episode[env] = (int)(epi + 1u);      // (one wave per env: nobody else reads it)
                // terminal state of the finished episode (drone_env.py:258 returns it; the reset below overwrites it)
                float *const c_pos_final = ca.pos_final, *const c_z_final = ca.z_final;
                int *const c_nbr_final = ca.nbr_final;
                if (rs && c_pos_final != nullptr) st_out2(c_pos_final + 2 * (so + wga0 + lane_c), xi, yi);
                if (rs && has_acc && agent < 2) {                 // retire the finished episode (train_problem.py:118-121)
                    double *tot = ca.acc + 8 * (size_t)env + 4 + 2 * agent;       // agent 0: done_return, done_true_return
                    if (agent == 0) {                                             // agent 1: done_collisions, done_len
                        const double2 dn = *reinterpret_cast<const double2 *>(tot);
                        *reinterpret_cast<double2 *>(tot) =
                            make_double2(dn.x + __builtin_bit_cast(double, make_uint2(accw.x, accw.y)),
                                         dn.y + __builtin_bit_cast(double, make_uint2(accw.z, accw.w)));
                        accw = make_uint4(0u, 0u, 0u, 0u);
                    } else {
                        const longlong2 dl = *reinterpret_cast<const longlong2 *>(tot);
                        *reinterpret_cast<longlong2 *>(tot) = make_longlong2(dl.x + (int)accw.x, dl.y + (int)accw.y);
                        accw = make_uint4(0u, 0u, accw.z + 1u, accw.w);           // episodes += 1
                    }
                }
                // ---- N distinct lattice nodes, the acceptance rule of reset_kernel (an unsettled agent settles on its
                // proposal iff no settled agent holds that node and no lower-index agent proposed it this round), found
                // through an open-addressing table in LDS instead of an O(N) scan per agent: settled agents enter their
                // node as blockers (owner -1), proposers enter theirs with owner = min(agent index); a proposer wins iff
                // it owns its entry.  Keys are compared exactly, so the draw is the one reset_kernel / the oracle make.
                const int nt = ca.samp_tbl;                       // entries per env slot, a power of two >= 2 N
                const uint32_t c_key0 = ca.key0, c_key1 = ca.key1, c_lat_M = ca.lat_M;
                const int c_shift = ca.samp_shift, c_div_y = ca.div_y;
                const float c_pitch = ca.pitch;
                // (the sampling tables sit behind the per-wave partial sums, see the LDS carve-up)
                // (workgroup-per-env geometries, single-step launches: the table shares the cell tables' region -- dead since pass 1;
                // sized on the host as max(cell tables, sampling table).  The fused rollouts keep their own: sharing measured +2 % there)
                int2 *tbl = (BLOCKGEO && !is_rollout(MODE)) ? reinterpret_cast<int2 *>(sbt_all)
                                     : reinterpret_cast<int2 *>(reinterpret_cast<float *>(smem + ca.lds_tail) + 2 * ((nwaves + 1) & ~1)) + (size_t)slot * nt;
                const uint32_t gid_c = ca.gid_base + (uint32_t)env;
                int node = -1;
                uint32_t round = 0;
                bool more;
                do {
                    if (rs)
                        for (int o = agent; o < nt; o += N) tbl[o] = make_int2(-1, 0x7fffffff);
                    group_sync<WL>();
                    // the "re-sample this env" word is cleared only now: every wave of the workgroup has read it (the
                    // barrier above) -- clearing it right behind agent 0's own read let a late wave see 0, skip this
                    // block and miss its barriers
                    if (!SYM && round == 0 && rs && agent == 0) sred[2 * slot + 1] = 0;
                    if (!WL && round == 0 && rs && agent == 0) c_episode[env] = (int)(epi + 1u);   // (every wave has read the old value)
                    int prop = node, h = 0;
                    if (rs) {
                        if (node < 0)
                            prop = (int)__umulhi(philox4x32_10_word0((uint32_t)agent, round, gid_c, epi, c_key0, c_key1), c_lat_M);
                        h = (int)(((uint32_t)prop * 0x9E3779B1u) >> c_shift);
                        for (;;) {                                                // linear probing, load factor <= 1/2
                            const int old = atomicCAS(&tbl[h].x, -1, prop);
                            if (old == -1 || old == prop) break;
                            h = (h + 1) & (nt - 1);
                        }
                        atomicMin(&tbl[h].y, node >= 0 ? -1 : agent);
                    }
                    group_sync<WL>();
                    if (rs && node < 0 && tbl[h].y == agent) node = prop;
                    const bool left = rs && node < 0;
                    more = WL ? (__builtin_amdgcn_ballot_w64(left) != 0ull) : (__syncthreads_or(left ? 1 : 0) != 0);
                    if (WL) group_sync<true>();                   // this round's reads before the next round's clearing
                    ++round;
                } while (more && round < (1u << 20));
                if (rs) {
                    if (node >= 0) {                              // (an agent still unsettled after 2^20 rounds keeps its place)
                        const int idx = node / c_div_y, jdx = node - idx * c_div_y;
                        xi = (float)idx * c_pitch; yi = (float)jdx * c_pitch;     // drone_env.py:196-205
                    }
                    vxi = 0.f; vyi = 0.f;                                         // :189
                    tcur = 0;                                                     // :100
                    epi += 1u;
                    __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0): the counter store above has landed
                    if (CACHED) refx = __builtin_nanf("");                        // the candidate list is stale
                    spos_env[agent] = make_float2(xi, yi);
                }
                group_sync<WL>();
                __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0): this wave's earlier z / Ni / state stores
                if (rs && (c_z_final != nullptr || c_nbr_final != nullptr)) {
                    // terminal observation (the `new_z` of the episode's last transition, utils.py:244-249): this lane's
                    // rows of z / Ni as the hot path has just written them -- all of them by THIS wave, whose stores have
                    // been acknowledged above -- are read back past L1 and kept before the re-observation replaces them
                    const size_t row = so + wga0 + lane_c;
                    if (c_z_final != nullptr) {
                        const float *src = a.z + row * (size_t)((K + 1) * zc);
                        float *dst = c_z_final + row * (size_t)((K + 1) * zc);
                        for (int w = 0; w < (K + 1) * zc; ++w) st_out(dst + w, __builtin_nontemporal_load(src + w));
                    }
                    if (c_nbr_final != nullptr) {
                        const int *src = a.nbr_idx + row * (size_t)(K + 1);
                        int *dst = c_nbr_final + row * (size_t)(K + 1);
#pragma unroll
                        for (int w = 0; w <= K; ++w) st_out(dst + w, __builtin_nontemporal_load(src + w));
                    }
                    __builtin_amdgcn_s_waitcnt(0x0f70);           // the read-back has returned before the rows are rewritten
                }
                if (rs) {                                         //           have landed before they are overwritten
                    list.init(dii, agent);
                    in_range = in_range0;
                    if (WMAX <= 4) {
                        // one cheap scan builds this agent's partner mask, then it walks its own bits: the expensive
                        // pair arithmetic runs max-over-lanes(partners) times, not once per agent of the env
                        unsigned long long bits[WMAX];
#pragma unroll
                        for (int w = 0; w < WMAX; ++w) {
                            bits[w] = 0ull;
                            const int jn = min(64, N - 64 * w);
#pragma nounroll
                            for (int u = 0; u < jn; ++u) {
                                const float2 pj = spos_env[64 * w + u];
                                const float dx = xi - pj.x, dy = yi - pj.y;
                                const bool hit = FAR || fmaf(dy, dy, dx * dx) < thr;
                                bits[w] |= (hit && 64 * w + u != agent) ? (1ull << u) : 0ull;
                            }
                        }
                        // ONE loop over the bits of all words, every lane taking its own lowest partner per trip (ascending
                        // order, like every other path).  Not one loop per word: hipcc (ROCm 7.2) lowered the four
                        // exec-masked per-word loops of the k = 3 rollout of the episode layer, at its 128-register budget,
                        // with the copy that merges `in_range` behind the LAST word's loop placed ahead of the exec
                        // restore -- lanes without partners in that word (all of them at N = 128) kept a stale temporary
                        // there and the re-observed rows marked partners outside their Delta disk as neighbours
                        // (tools/fuzz_rollout.py seeds 7 / 9).  A single loop has a single merge point, the shape
                        // of the hot walk.
                        unsigned long long left = 0ull;
#pragma unroll
                        for (int w = 0; w < WMAX; ++w) left |= bits[w];
                        while (left != 0ull) {
                            unsigned long long hs = bits[WMAX - 1];
                            int ws = WMAX - 1;
#pragma unroll
                            for (int w = WMAX - 2; w >= 0; --w) { if (bits[w] != 0ull) { hs = bits[w]; ws = w; } }
                            const int u = __builtin_ctzll(hs);
                            hs &= hs - 1ull;
                            left = 0ull;
#pragma unroll
                            for (int w = 0; w < WMAX; ++w) { if (w == ws) bits[w] = hs; left |= bits[w]; }
                            visit(64 * ws + u, NoDefer{}, UniRuntime{});
                        }
                    } else {
#pragma nounroll
                        for (int j = 0; j < N; ++j) {             // ascending order, like every other path
                            if (j == agent) continue;
                            if (!FAR) {
                                const float2 pj = spos_env[j];
                                const float dx = xi - pj.x, dy = yi - pj.y;
                                if (!(fmaf(dy, dy, dx * dx) < thr)) continue;
                            }
                            visit(j, NoDefer{}, UniRuntime{});
                        }
                    }
                    const float zx = (xi - xFx) - xLx, zy = (yi - xFy) - xLy;
                    const float gsc = __builtin_amdgcn_rsqf(fmaf(zy, zy, zx * zx)) * delta_i * a.ghost_factor;
                    float *zr = a.z + (so + wga0 + lane_c) * (size_t)((K + 1) * zc);
                    int *nb = a.nbr_idx + (so + wga0 + lane_c) * (size_t)(K + 1);
#pragma unroll
                    for (int kth = 0; kth <= K; ++kth) {
                        const unsigned j = list.index(kth);
                        const bool hv = kth == 0 || j < (unsigned)N;
                        const bool real = kth == 0 || (kth <= in_range && hv);
                        float rx = kth == 0 ? zx : zx * gsc, ry = kth == 0 ? zy : zy * gsc;
                        if (kth > 0 && real) {
                            const float2 pj = spos_env[j];
                            rx = pj.x - xi; ry = pj.y - yi;
                        }
                        // streaming stores like every other output: plain ones would leave the rows dirty in L2 and the
                        // launch would end with their write-back (a launch in which every env restarts: 47 -> see DESIGN 3.3)
                        st_out(nb + kth, kth == 0 ? agent : (real ? (int)j : -1));
                        float *row = zr + kth * zc;
                        if (zc == 5) {
                            st_out(row + 0, rx); st_out(row + 1, ry);          // 20-byte rows: 4-byte aligned only
                            if (kth == 0) { st_out(row + 2, 0.f); st_out(row + 3, 0.f); st_out(row + 4, li); }
                            else if (hv) { st_out(row + 2, 0.f); st_out(row + 3, 0.f); st_out(row + 4, uni_args ? a.radius_u : sconst[j].y); }
                            else { const float qn = __builtin_nanf(""); st_out(row + 2, qn); st_out(row + 3, qn); st_out(row + 4, qn); }
                        } else {
                            st_out2(row, rx, ry);
                        }
                    }
                    if (!is_rollout(MODE) || step == nsteps - 1) {
                        st_g2(o_pos + 2 * lane_c, xi, yi);
                        st_g2(o_vel + 2 * lane_c, 0.f, 0.f);
                    }
                }
                asm volatile("s_nop 14");
            }
        }
        if (is_rollout(MODE)) {
            group_sync<WL>();                                 // staging / sred reuse by the next step
            p_reward += step_agents; p_true += step_agents;
            p_gz += step_agents * kZRow; p_gn += step_agents * kNRow;
            p_ncoll += a.E; p_done += a.E;
        }
    }
    if (SYM) {                                                // scalar bases + lane offsets
        if (MODE != kObserve && lane == 0) *o_t = tcur;
        if (has_acc && lane < 2) { u32x4 w; w.x = accw.x; w.y = accw.y; w.z = accw.z; w.w = accw.w; o_acc[lane] = w; }
    } else {
        if (MODE != kObserve && valid && agent == 0) a.t[env] = tcur;
        if (has_acc && valid && agent < 2) *reinterpret_cast<uint4 *>(a.acc + 8 * (size_t)env + 2 * agent) = accw;
    }
#undef has_acc
#undef auto_reset
    TRACE_MARK(5);
    if (kTraceSpan && a.trace && (threadIdx.x & 63) == 0)        // one word per wave: entry (low) | exit (high), stores NOT waited for
        a.trace[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] =
            (trace_rt0 & 0xffffffffll) | ((long long)__builtin_amdgcn_s_memrealtime() << 32);
    if (kTrace) {
        __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0): all stores acknowledged
        TRACE_MARK(6);
        if (a.trace && (threadIdx.x & 63) == 0)
            a.trace[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + 7] =    // entry and exit on the global clock
                (trace_rt0 & 0xffffffffll) | ((long long)__builtin_amdgcn_s_memrealtime() << 32);
    }
}


struct Geometry {
    int P, epb, threads, blocks, geo;
    size_t lds;
};

// more than 48 KiB of dynamic LDS (envs of several hundred agents) has to be opted into once per kernel
template <int K, bool FAR, int MODE, int GEO, bool EPI>
hipError_t launch_one(const KArgs &a, const Geometry &g, hipStream_t s)
{
    if (g.lds > 48 * 1024) {
        // once per (kernel, device): remembered in a per-instantiation bit mask of device ordinals under a mutex
        static std::mutex mu;
        static unsigned long long opted[4] = {0ull, 0ull, 0ull, 0ull};   // device ordinals 0..255
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 255) dev = 0;
        std::lock_guard<std::mutex> lock(mu);
        if (!(opted[dev >> 6] >> (dev & 63) & 1ull)) {
            // the dynamic allowance is what the CU's 160 KiB leave next to the kernel's STATIC LDS: hipcc may promote a
            // small per-lane array to LDS (256 B in the K = 7 episode-layer instantiations), and asking for all 160 KiB
            // on top of that is refused
            const void *fn = reinterpret_cast<const void *>(drone_kernel<K, FAR, MODE, GEO, EPI>);
            hipFuncAttributes fa{};
            hipError_t e = hipFuncGetAttributes(&fa, fn);
            if (e != hipSuccess) return e;   // reported by launch() through dronesim_last_error()
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
            if (e != hipSuccess) return e;
            opted[dev >> 6] |= 1ull << (dev & 63);
        }
    }
    hipLaunchKernelGGL((drone_kernel<K, FAR, MODE, GEO, EPI>), dim3(g.blocks), dim3(g.threads), g.lds, s,
                       a.pos, MODE == kObserve ? static_cast<const float *>(a.vel) : a.act,
                       (int)((unsigned)a.P | ((unsigned)g.blocks & ~255u)),   // P <= 64; whole groups of 256 workgroups (XCD map)
                       EPI ? (a.epb | ((a.acc != nullptr ? 1 : 0) | (a.auto_reset ? 2 : 0) | (a.rand_act ? 4 : 0)) << 16) : a.epb,
                       a.E, a.N, a);
    return hipSuccess;
}

template <int K, bool FAR, int GEO>
hipError_t launch_mode(int mode, const KArgs &a, const Geometry &g, hipStream_t s)
{
    bool epi = a.acc != nullptr || a.auto_reset != 0 || a.rand_act != 0;   // DroneEpisodeCtl in use
    switch (mode) {
    case kStep: return epi ? launch_one<K, FAR, kStep, GEO, true>(a, g, s) : launch_one<K, FAR, kStep, GEO, false>(a, g, s);
    case kObserve: return launch_one<K, FAR, kObserve, GEO, false>(a, g, s);
    default:
        if (!epi) return launch_one<K, FAR, kRollout, GEO, false>(a, g, s);
        if constexpr (GEO == kSym64)   // the action source at compile time (see Mode)
            return a.rand_act ? launch_one<K, FAR, kRolloutRand, GEO, true>(a, g, s) : launch_one<K, FAR, kRolloutPool, GEO, true>(a, g, s);
        else
            return launch_one<K, FAR, kRollout, GEO, true>(a, g, s);
    }
}

template <int K>
hipError_t launch_k(int mode, bool far, const KArgs &a, const Geometry &g, hipStream_t s)
{
    switch (g.geo) {
    case kSym64: return far ? launch_mode<K, true, kSym64>(mode, a, g, s) : launch_mode<K, false, kSym64>(mode, a, g, s);
    case kPacked:
        return far ? launch_mode<K, true, kPacked>(mode, a, g, s) : launch_mode<K, false, kPacked>(mode, a, g, s);
    case kBlock256:
        return far ? launch_mode<K, true, kBlock256>(mode, a, g, s) : launch_mode<K, false, kBlock256>(mode, a, g, s);
    case kBlockU256: {                              // rollouts only (launch() picks it for mode == kRollout)
        const bool epi = a.acc != nullptr || a.auto_reset != 0 || a.rand_act != 0;
        if (epi) return a.rand_act ? launch_one<K, false, kRolloutRand, kBlockU256, true>(a, g, s) : launch_one<K, false, kRolloutPool, kBlockU256, true>(a, g, s);
        return launch_one<K, false, kRollout, kBlockU256, false>(a, g, s);
    }
    default:
        return far ? launch_mode<K, true, kBlock1024>(mode, a, g, s) : launch_mode<K, false, kBlock1024>(mode, a, g, s);
    }
}


}   // namespace
