// policy.hip -- batched per-agent 3-layer MLP forward + action sampling on gfx950 matrix cores.
//
// The reference evaluates one small torch MLP per agent per step in a Python loop
// (SAC_agents.py:170-180 -> utils.py:304-309 / 110-117 / 40-53): with the environment on the device this
// is the whole rollout time.  Here ALL agents' networks run in one launch over the
// batched observation z[E][N][d_in]:
//     h1 = relu(x W1_i + b1_i)        utils.py:291-292 / 91-92 / 42-43
//     h2 = relu(h1 W2_i + b2_i)       utils.py:295-296 / 95-99 / 46-47
//     y  = h2 W3_i + b3_i             utils.py:299 / 102-106 / 50
//     out = softmax(y) | (tanh, sigmoid) | y       utils.py:300 / 103,106 / --
// plus the sampling of sample_action (categorical over unit-circle actions utils.py:262-269,304-309;
// Gaussian utils.py:110-117) from a counter-based Philox stream.
//
// Arithmetic: exact float32 on the matrix cores -- v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain
// (no reduced precision), so results match a float32 torch reference to round-off.
// Since round 6 the host class hands the weights over as ONE packed stream per agent (DroneMlp.w2_layout = 2) and the kernel is
// mlp3_rt_kernel further down: a wave owns 32 env rows and every output chunk, the layers meet in registers.  The kernel described
// here (mlp3_kernel: w2_layout = 0 / 1, activations staged through LDS) is the one of rounds 2-5.
// Decomposition: ceil(E/32) x N workgroups (XCD-aware order, see xcd_work_item): one workgroup = 32 env rows of ONE agent, 4 waves.
//   wave w owns every fourth 32-column chunk of the hidden layers (one 32x32 accumulator tile).
//   32 rows keep the workgroup at ~57 KiB of LDS, so two workgroups share a CU and one's prologue, barriers
//   and output stage overlap the other's MFMAs (64-row workgroups -- one per CU -- measured 6-12 % slower).
//   layer 1: x tile (LDS) x W1 (global/L2)                   -> relu -> h1 tile in LDS [32][ld1]
//   layer 2 chunk (32 columns): h1 (LDS) x W2 (fragment-packed, L2) -> relu -> per-wave LDS staging [32][36]
//   layer 3 partial: staged chunk x W3 rows of the chunk (v_mfma_f32_16x16x4_f32 when nout <= 16) -> registers
//   the four waves' partials are summed through LDS, then activation + sampling.
// With packed W2 and nout <= 16 (what the host class passes for every network of the reference) layers 1 and 2 are
// computed transposed -- weights as the A operand -- so that tiles leave the accumulators as 16-byte row pieces
// (store_tile_tr); the plain-layout path (w2_layout = 0, or nout > 16) keeps the row-major tiles and odd LDS strides
// (h1 + 1, 33) of rounds 2-3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>

#include "common.hpp"
#include "dronesim.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kRows = 32;                // env rows per workgroup (32-row tiles x 4 feature waves each)
constexpr int kThreadsF = kRows * 8;
constexpr int kMaxOut = 32;

long long *g_policy_trace = nullptr;     // developer trace builds (kTrace, common.hpp) only: [workgroups][4 waves][8] timestamps
// (mlp3_rt16_kernel: [workgroups][4 waves][64] stamps -- tools/trace_rt16.py)
#define PT64(k) do { if (kTrace && a.trace && lane == 0) a.trace[((size_t)blockIdx.x * 4 + wave) * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PT(k) do { if (kTrace && a.trace && lane == 0) a.trace[((size_t)blockIdx.x * 4 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

struct FinishArgs {
    int N, nout, out_kind, sample_kind;
    float *out, *act;
    int *act_idx;
    uint32_t key0, key1, ctr2, ctr3;
    long long env_base;
    const int *t_dev, *episode_dev;
};

struct MArgs {
    int E, N, d_in, h1, h2, nout;
    const float *x, *w1, *b1, *w2, *b2, *w3, *b3;
    FinishArgs fin;
    long long *trace;                    // developer trace builds only (NULL otherwise)
    unsigned rb_magic;                   // xcd_work_item: ceil(2^32 / row blocks), or 0
};

// LDS row stride (floats) of the h1 tile for the packed layer 2: whole 32-column chunks (the k padding is written as
// zeros by layer 1), a multiple of 4 (16-byte aligned rows) whose quotient is odd (ds_read_b128 phases conflict-free)
__host__ __device__ __forceinline__ int packed_row_stride(int h1)
{
    const int w = ((h1 + 31) >> 5) * 32;
    return ((w >> 2) & 1) ? w : w + 4;
}

// Workgroup -> (agent, row block).  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs,
// each with its own 4 MiB L2; all agents' weights together (12.8 MiB in bf16 / 23 MiB in f32 at N = 64,
// h = 300) do not fit one L2, a few agents' do.  So the work list is ordered agent-major and cut into 8
// contiguous pieces, one per XCD: XCD x walks its own agents one after the other, the ~64 workgroups resident
// on it at any time share one or two agents' weights, and every weight byte leaves HBM once per launch.
// `magic` = ceil(2^32 / row_blocks) from the host when total x row_blocks < 2^32 (then umulhi(v, magic) = v / row_blocks
// exactly for every v < total), else 0: a run-time integer division is ~25 instructions of this prologue, each of which
// waits for an issue slot next to the other workgroup's matrix stream.
__host__ inline unsigned div_magic(unsigned long long total, unsigned d)
{
    return (d > 1 && total * d < (1ull << 32)) ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u;
}
__device__ __forceinline__ void xcd_work_item(int row_blocks, int &agent, int &row_block, unsigned magic = 0u)
{
    const int total = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q = total >> 3, r = total & 7;
    const int v = xcd * q + min(xcd, r) + slot;            // XCD x owns q + (x < r) items
    agent = magic ? (int)__umulhi((unsigned)v, magic) : v / row_blocks;
    row_block = v - agent * row_blocks;
}

// C/D layout of v_mfma_f32_32x32x2_f32: element reg r of lane l is (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31)
__device__ __forceinline__ int cd_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Output activation + sampling of ONE env row by the 4 adjacent lanes of a quad: lane `part` holds the
// pre-activation outputs j = part + 4 i (i < 8) in y[i].  Reductions over the row (softmax max / sum, the
// categorical cdf) run on DPP quad permutes, so the serial tail of the kernel is a quarter as long and the
// probabilities leave as 16-byte segments.  tval / epval: the env's step and episode counters (0 if absent).
constexpr int kQ = kMaxOut / 4;

template <int CTRL> __device__ __forceinline__ float quad_perm(float v)     // CTRL = quad_perm:[a,b,c,d] = a | b<<2 | c<<4 | d<<6
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kQuadUp1 = 0x90, kQuadUp2 = 0x40, kQuadLast = 0xFF;

__device__ __forceinline__ void finish_quad(const FinishArgs &a, float (&y)[kQ], int e, int agent, int part,
                                            uint32_t tval, uint32_t epval)
{
    const int nout = a.nout;
    if (a.out_kind == 1) {                                   // softmax (utils.py:286, dim = 0 of one sample)
        float m = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < kQ; ++i) if (part + 4 * i < nout) m = fmaxf(m, y[i]);
        m = fmaxf(m, quad_perm<kQuadXor1>(m));
        m = fmaxf(m, quad_perm<kQuadXor2>(m));
        float ssum = 0.0f;
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            // (v_exp_f32 = 2^x, 1 ulp: the library expf is ~15 instructions of range reduction for arguments that are <= 0
            // here, and every vector instruction of this tail waits for an issue slot next to the other workgroup's
            // matrix stream -- about one per 64 cycles, DESIGN_LOG round 5)
            if (4 * i < nout) { y[i] = part + 4 * i < nout ? __builtin_amdgcn_exp2f((y[i] - m) * 1.4426950408889634f) : 0.0f; ssum += y[i]; }
            else y[i] = 0.0f;
        }
        ssum += quad_perm<kQuadXor1>(ssum);
        ssum += quad_perm<kQuadXor2>(ssum);
        const float inv = __builtin_amdgcn_rcpf(ssum);             // ssum in [1, nout]: v_rcp_f32, 1 ulp
#pragma unroll
        for (int i = 0; i < kQ; ++i) y[i] *= inv;
    } else if (a.out_kind == 2) {                            // tanh means, sigmoid variances (utils.py:74-77)
        const int half = nout / 2;
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = part + 4 * i;
            if (j < nout) {                                      // tanh = 1 - 2 / (e^2y + 1), sigmoid = 1 / (1 + e^-y): absolute 1e-7
                const float ex = __builtin_amdgcn_exp2f(y[i] * (j < half ? 2.8853900817779268f : -1.4426950408889634f));
                const float rc = __builtin_amdgcn_rcpf(ex + 1.0f);
                y[i] = j < half ? fmaf(-2.0f, rc, 1.0f) : rc;
            }
        }
    }
    const size_t row = (size_t)e * a.N + agent;
    if (a.out) {
#pragma unroll
        for (int i = 0; i < kQ; ++i) if (part + 4 * i < nout) a.out[row * nout + part + 4 * i] = y[i];
    }
    if (a.sample_kind != 0) {
        uint32_t rnd[4];
        philox4x32_10((uint32_t)agent, (uint32_t)(a.env_base + e), a.ctr2 + tval, a.ctr3 + epval, a.key0, a.key1, rnd);
        if (a.sample_kind == 1) {                            // categorical -> unit vector (utils.py:262-269, 304-309)
            const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
            float base = 0.0f;                               // cdf up to the previous group of 4 outputs
            int below = 0;                                   // outputs j with cdf_j <= u: the pick is the first j with u < cdf_j
#pragma unroll
            for (int i = 0; i < kQ; ++i) {
                if (4 * i < nout) {
                    const bool valid = part + 4 * i < nout;
                    float incl = valid ? y[i] : 0.0f;        // inclusive scan over the quad
                    const float n1 = quad_perm<kQuadUp1>(incl);
                    if (part >= 1) incl += n1;
                    const float n2 = quad_perm<kQuadUp2>(incl);
                    if (part >= 2) incl += n2;
                    const float cdf = base + incl;
                    if (valid && !(u < cdf)) ++below;
                    base = quad_perm<kQuadLast>(cdf);
                }
            }
            below += __builtin_amdgcn_update_dpp(0, below, kQuadXor1, 0xF, 0xF, true);
            below += __builtin_amdgcn_update_dpp(0, below, kQuadXor2, 0xF, 0xF, true);
            const int pick = min(below, nout - 1);
            if (part == 0) {
                if (a.act_idx) a.act_idx[row] = pick;
                if (a.act) {                                    // v_cos_f32 / v_sin_f32 take REVOLUTIONS: cos(2 pi pick / nout) directly
                    const float rev = (float)pick * __builtin_amdgcn_rcpf((float)nout);
                    *reinterpret_cast<float2 *>(a.act + row * 2) = make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
                }
            }
        } else {                                             // Gaussian, Box-Muller (utils.py:110-117); nout = 4:
            const float var = quad_perm<kQuadXor2>(y[0]);    // lane d < 2 holds mu_d, lane d + 2 its variance
            if (part < 2 && a.act) {
                const uint32_t r0 = part == 0 ? rnd[0] : rnd[2], r1 = part == 0 ? rnd[1] : rnd[3];
                const float u1 = ((float)(r0 >> 8) + 1.0f) * (1.0f / 16777216.0f);            // (0, 1]
                const float u2 = (float)(r1 >> 8) * (1.0f / 16777216.0f);
                // Box-Muller on the hardware's log2 / sqrt / cos(2 pi x): v_log_f32, v_sqrt_f32, v_cos_f32 (input in revolutions)
                const float n01 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
                a.act[row * 2 + part] = fmaf(__builtin_amdgcn_sqrtf(var), n01, y[0]);
            }
        }
    }
}

// acc += A[32 x K] * B[K x 32].  A row-major in LDS (lda floats per row, odd stride -> conflict-free), B row-major
// in global / L2 (ldb floats per row); B columns >= ncols_valid read a clamped column (never stored), k >= K reads
// as zero.  Operand loads of the next 8 k-steps are issued before the 8 MFMAs of the current ones.
constexpr int kU = 8;                  // k-steps (of 2) per pipeline stage

struct Frag { float b[kU], a[kU]; };

template <bool CHECK>                                   // CHECK: k >= K reads as zero (the ragged last stage)
__device__ __forceinline__ void load_frag(Frag &f, const float *Arow, const float *__restrict__ Bcol, int ldb, int kbase, int K)
{
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int k = kbase + 2 * u;
        const int kc = CHECK ? min(k, K - 1) : k;          // clamped address, value masked below: no branches
        const float b = Bcol[(size_t)kc * ldb], av = Arow[kc];
        f.b[u] = (!CHECK || k < K) ? b : 0.0f;
        f.a[u] = (!CHECK || k < K) ? av : 0.0f;
    }
}

__device__ __forceinline__ void tile_gemm(f32x16 &acc, const float *A, int lda, const float *__restrict__ B, int ldb,
                                          int K, int ncols_valid, int lane)
{
    const int ar = lane & 31, kk = lane >> 5;
    const float *Bcol = B + min(ar, ncols_valid - 1);
    const float *Arow = A + ar * lda;
    const int Kmain = K - K % (2 * kU);                    // whole pipeline stages, no bounds checks inside
    if (Kmain > 0) {
        Frag cur, nxt;
        load_frag<false>(cur, Arow, Bcol, ldb, kk, K);
        // The first stage's operands are waited for HERE, once, ahead of the loop.  Left pending into the loop, they make
        // hipcc place counted waits (vmcnt(7) ... vmcnt(0)) in front of the eight MFMAs of the loop body -- needed on
        // the first trip, where the MFMAs' operands are those loads, but the same instructions then also wait on every
        // later trip, where the only loads in flight are the NEXT stage's: the prefetch distance collapsed from a full
        // stage (512 cycles of MFMAs) to the position inside the stage, and every stage stalled on the L2 latency
        // (round 2: 0.46 of the matrix peak).  With nothing pending at the loop's entry the only wait left is the one
        // in front of the `cur = nxt` copies, a whole stage after the loads were issued.
        __builtin_amdgcn_s_waitcnt(0x0070);                // vmcnt(0) lgkmcnt(0)
        for (int k0 = 0; k0 < Kmain; k0 += 2 * kU) {
            const bool more = k0 + 2 * kU < Kmain;         // wave-uniform
            if (more) load_frag<false>(nxt, Arow, Bcol, ldb, k0 + 2 * kU + kk, K);
#pragma unroll
            for (int u = 0; u < kU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u], cur.b[u], acc, 0, 0, 0);
            if (more) cur = nxt;
        }
    }
    if (Kmain < K) {                                       // ragged rest (and all of a K < 16 layer) as ONE masked stage:
        Frag t;                                            // its loads are in flight together instead of one per MFMA
        load_frag<true>(t, Arow, Bcol, ldb, Kmain + kk, K);
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (Kmain + 2 * u < K) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.a[u], t.b[u], acc, 0, 0, 0);   // wave-uniform
    }
}

// ---- layer 2 on fragment-packed weights (DroneMlp.w2_layout = 1, what the host class passes) ------------------------
// One pipeline stage = 16 k-values = 8 MFMAs.  Lane (col = lane & 31, half = lane >> 5) feeds MFMA u of stage s with
// k = 16 s + 8 half + u: its eight A values are CONSECUTIVE floats of its h1 row in LDS (two ds_read_b128; the row
// stride is a multiple of 4 floats with an odd quotient, so the 16 lanes of a read phase hit 16 different bank
// quads), its eight B values two 16-byte pieces of the packed chunk
//     w2p[agent][chunk c][stage s][q][lane][4] = W2[16 s + 8 half + 4 q + jj][32 c + col]       (zero beyond h1 / h2)
// which the wave reads as two fully coalesced 1 KiB loads (scalar base + lane * 16 + immediate).  The reference-layout
// loop above needs 8 dword loads, 4 LDS reads, 16 64-bit address additions and 16 register copies per stage (4 VALU
// per MFMA -- round-3 counters: 7 VALU instructions per MFMA over the kernel, the matrix pipe 54-60 % busy with two
// waves per SIMD); this one 2 + 2 loads, no copies (R register sets, the loop unrolled R times) and scalar address
// updates, with the loads R - 1 stages ahead of their use.  Measured at the C5 shard (Gaussian actor, h = 400):
// reference layout 521 us; R = 1 / 2 / 3 / 4: 505 / 417-424 / 436-445 / 460 us -- one stage (512 matrix cycles) of
// distance is enough with a second wave on the SIMD, deeper only keeps more loads and registers in flight.  Requesting
// the next chunk's first stages and the chunk's W3 rows early (across the layer-3 part) was measured too: +-1 %.
// Per-wave trace of this kernel (tools/trace_policy.py gaussian c5 f32): layer 2 is 72 % of a wave's life, layer 1 20 %
// (8.4k ticks waiting for W1 / b1 / x, 13.7k for ~200 instructions: while the OTHER workgroup's wave on the SIMD streams
// matrix instructions, this one gets about one issue slot per matrix instruction); raising the issue priority of the
// waves outside their layer-2 loop (s_setprio 2 / 3) costs 2.4 % instead of helping.  Matrix pipe busy 70 % (54 %).
#define POLICY_SETS 2         // register sets of the pipeline (see tile_gemm_packed)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct FragSet { f32x4 a0, a1, b0, b1; };

__device__ __forceinline__ void load_set(FragSet &t, const float *Arow, const f32x4 *Bp, int s)
{
    const f32x4 *ap = reinterpret_cast<const f32x4 *>(Arow + 16 * s);
    const f32x4 *bp = Bp + (size_t)s * 128;
    t.b0 = bp[0]; t.b1 = bp[64];
    t.a0 = ap[0]; t.a1 = ap[1];
}
template <bool TR>                        // TR: the transposed product D^T[feature][row] (the weights as the A operand)
__device__ __forceinline__ void mfma_set(f32x16 &acc, const FragSet &t)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(TR ? t.b0[u] : t.a0[u], TR ? t.a0[u] : t.b0[u], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(TR ? t.b1[u] : t.a1[u], TR ? t.a1[u] : t.b1[u], acc, 0, 0, 0);
}

// Transposed tiles (TR).  D[i][j] of v_mfma_f32_32x32x2_f32 puts FOUR CONSECUTIVE i (registers 4 q .. 4 q + 3 = rows
// 8 q + 4 (lane >> 5) + 0..3) of column j = lane & 31 into a lane.  With the weights as the A operand i is the feature and
// j the env row, so a lane's registers are contiguous pieces of its row of the next layer's input: the relu'd tile goes
// to LDS as four ds_write_b128 instead of sixteen ds_write_b32, and the bias -- one value per feature, i.e. per A row
// -- rides on one more matrix instruction (A = bias in the k slot of lanes 0..31, B = 1 there, 0 in the other k slot:
// fmaf(bias, 1, acc), rounded exactly like acc + bias) instead of sixteen v_add.  Per 32 x 32 tile: 16 v_max + 4 wide
// writes instead of 16 x (v_add, v_max, ds_write_b32).  Every instruction saved here is an issue slot the OTHER
// workgroup's wave on the SIMD gets for its matrix stream (the float32 matrix instructions run on the vector ALUs).
__device__ __forceinline__ f32x16 bias_mfma(f32x16 acc, float bias, int lane)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(lane < 32 ? bias : 0.0f, lane < 32 ? 1.0f : 0.0f, acc, 0, 0, 0);
}
template <bool RELU>
__device__ __forceinline__ void store_tile_tr(float *rowp, const f32x16 &acc, int lane)     // rowp: this lane's row + chunk offset
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = RELU ? fmaxf(acc[4 * q + t], 0.0f) : acc[4 * q + t];
        *reinterpret_cast<f32x4 *>(rowp + 8 * q + 4 * (lane >> 5)) = v;
    }
}

// acc += (h1 tile, stages [sb, sb + n)) x (packed chunk).  Arow: this lane's LDS row + 8 half; Bp: chunk base + lane.
template <int R, bool TR>
__device__ __forceinline__ void tile_gemm_packed(f32x16 &acc, const float *Arow, const f32x4 *Bp, int sb, int n)
{
    FragSet set[R];
    if (n < R - 1) {                                         // (a hidden layer of <= 16 (R - 2) units)
        for (int s1 = 0; s1 < n; ++s1) { load_set(set[0], Arow, Bp, sb + s1); mfma_set<TR>(acc, set[0]); }
        return;
    }
#pragma unroll
    for (int r = 0; r < R - 1; ++r) load_set(set[r], Arow, Bp, sb + r);
    int s = 0;
    // steady state: R stages per trip, every load unconditional (a conditional one makes hipcc wait for one stage more
    // than needed at the join, which costs a whole stage of prefetch distance)
    for (const int last = n - (2 * R - 1); s <= last; s += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            load_set(set[(r + R - 1) % R], Arow, Bp, sb + s + r + R - 1);
            mfma_set<TR>(acc, set[r]);
        }
    }
    // the last <= 2 R - 2 stages
#pragma unroll
    for (int r = 0; r < 2 * R - 2; ++r) {
        if (s + r < n) {                                     // wave-uniform
            if (s + r + R - 1 < n) load_set(set[(r + R - 1) % R], Arow, Bp, sb + s + r + R - 1);
            mfma_set<TR>(acc, set[r % R]);
        }
    }
}

// ---- layer 3 of a chunk for nout <= 16 (every network of the reference: 16 action probabilities, 4 Gaussian moments, 1
// value) on v_mfma_f32_16x16x4_f32.  The 32x32x2 form spends 16 matrix instructions of 64 cycles per chunk on 32 output
// columns of which at most 16 exist (8 % of the kernel's matrix time at h = 400); two 16-row tiles x 8 k-steps of the
// 16-column instruction are 16 x 32 cycles.  Lane (i = lane & 15, g = lane >> 4) feeds k-step ks with k = 8 g + ks (any
// assignment of the chunk's 32 k values to (g, ks) is a valid contraction order; this one makes a lane's eight A values
// CONSECUTIVE floats of its staged row: two ds_read_b128 per tile instead of eight ds_read_b32; kStN = 36 floats per
// row = a multiple of 4 with an odd quotient, so the 8 lanes of a read phase hit different bank quads) and reads its
// eight W3 values once for both tiles.  D: reg r of lane l = (row 4 (l >> 4) + r, col l & 15).
constexpr int kStW = 33, kStN = 36;      // floats per row of the per-wave staging tile (wide / narrow layer 3)

__device__ __forceinline__ void layer3_narrow(f32x4 (&y)[2], const float *st, const float *__restrict__ w3c, int nout, int kvalid, int lane)
{
    const int g = lane >> 4, i = lane & 15, c = min(i, nout - 1);
    float b[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int k = 8 * g + ks;
        const float w = w3c[(size_t)min(k, kvalid - 1) * nout + c];                 // clamped address, masked value
        b[ks] = k < kvalid ? w : 0.0f;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const f32x4 *ap = reinterpret_cast<const f32x4 *>(st + (mt * 16 + i) * kStN + 8 * g);
        const f32x4 a0 = ap[0], a1 = ap[1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) y[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks], b[ks], y[mt], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) y[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], b[4 + ks], y[mt], 0, 0, 0);
    }
}

// kRows env rows of one agent per workgroup, 4 waves per 32-row tile: wave w owns feature chunks (w & 3),
// (w & 3) + 4, ... of the rows of tile (w >> 2).
// NU: k-steps (of 2) of layer 1 that are loaded and issued when d_in <= 16 (3 for the reference's simplified observation,
// d_in = 6; compile-time so that the loads stay unconditional -- wave-uniform `if (u < nu)` around them was measured +2 %:
// hipcc drains the loads at every join)
template <bool PACKED, bool NARROW, int NU = kU>      // PACKED: W2 in the fragment layout of tile_gemm_packed; NARROW: nout <= 16
__global__ void __launch_bounds__(kThreadsF, 2) mlp3_kernel(const float *x, int E, int N, int d_in, const MArgs rest)
{
    MArgs a = rest;                      // leading scalars are preloaded into SGPRs at wave launch (csrc/Makefile)
    a.x = x; a.E = E; a.N = N; a.d_in = d_in;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int agent, row_block;
    xcd_work_item((a.E + kRows - 1) / kRows, agent, row_block, a.rb_magic);
    // (rotating which wave owns the chunks 0, 4, 8, ... -- one more chunk than the others at h = 400 -- with the row block,
    // so that the heavy waves of the two workgroups on a CU sit on different SIMDs, was measured in round 3: +-1 %)
    const int cw = wave & 3, rh = wave >> 2;
    const int e0 = row_block * kRows;
    const int ldx = a.d_in + 1, ld1 = PACKED ? packed_row_stride(a.h1) : a.h1 + 1;
    float *sx = reinterpret_cast<float *>(smem);                 // [rows][d_in+1]
    float *sh1 = sx + kRows * ldx;                               // [rows][h1+1]
    constexpr int kSt = NARROW ? kStN : kStW;
    constexpr bool TR = PACKED && NARROW;                        // transposed tiles (see store_tile_tr)
    float *sst = sh1 + kRows * ld1;                              // [waves][32][kSt] layer-2 chunk staging,
                                                                 // reused for the layer-3 partials
    const float *w1 = a.w1 + (size_t)agent * a.d_in * a.h1, *b1 = a.b1 + (size_t)agent * a.h1;
    const int nst = (a.h1 + 15) >> 4;                            // PACKED: 16-k stages of layer 2
    const float *w2 = a.w2 + (PACKED ? (size_t)agent * ((a.h2 + 31) >> 5) * nst * 512 : (size_t)agent * a.h1 * a.h2);
    const float *b2 = a.b2 + (size_t)agent * a.h2;
    const float *w3 = a.w3 + (size_t)agent * a.h2 * a.nout, *b3 = a.b3 + (size_t)agent * a.nout;

    PT(0);
    const unsigned long long rt0 = kTrace ? __builtin_amdgcn_s_memrealtime() : 0ull;   // trace builds: 100 MHz clock at entry
    const int col = lane & 31;
    constexpr int kL1 = 4;
    float wb[kL1][NU], bias[kL1];
    const bool small_k = NU < kU || a.d_in <= 2 * kU;            // (the NU = 3 instance is only launched with d_in <= 6)
    if (small_k) {                                               // layer 1's weights travel together with the x tile
#pragma unroll
    for (int i = 0; i < kL1; ++i) {
        // no branch around these loads, not even the wave-uniform `chunk exists`: hipcc drains the loads of a
        // conditional block at its join (s_waitcnt vmcnt(0) per chunk: four round trips in series, 10k cycles of a wave's
        // 100k); a chunk that does not exist re-reads the last column and is never used
        const int c0 = cw * 32 + 128 * i, cc = min(c0 + col, a.h1 - 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int k = (lane >> 5) + 2 * u;
            const float w = w1[(size_t)min(k, a.d_in - 1) * a.h1 + cc];            // clamped address, masked value
            wb[i][u] = k < a.d_in ? w : 0.0f;
        }
        const float bv = b1[cc];
        bias[i] = c0 + col < a.h1 ? bv : 0.0f;
    }
    }
    // ---- x tile -> LDS (rows beyond E are zero)
    if (a.d_in <= 8) {                                       // eight lanes per row: no per-thread division
        const int r = tid >> 3, c = tid & 7, e = e0 + r;
        if (c < a.d_in) sx[r * ldx + c] = e < a.E ? a.x[((size_t)e * a.N + agent) * a.d_in + c] : 0.0f;
    } else
    for (int idx = tid; idx < kRows * a.d_in; idx += kThreadsF) {
        const int r = idx / a.d_in, c = idx - r * a.d_in;
        const int e = e0 + r;
        sx[r * ldx + c] = e < a.E ? a.x[((size_t)e * a.N + agent) * a.d_in + c] : 0.0f;
    }
    __syncthreads();
    PT(1);

    // ---- layer 1: K = d_in is tiny, so the operands of ALL of this wave's chunks (<= 4: h1 <= 512) and their
    //      biases are requested together -- one global round trip for the layer instead of two per chunk
    if (small_k) {
        const float *Arow = sx + (rh * 32 + (lane & 31)) * ldx;
        float xa[NU];                                            // the x operand is the same for every chunk
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int k = (lane >> 5) + 2 * u;
            const float v = Arow[min(k, a.d_in - 1)];
            xa[u] = k < a.d_in ? v : 0.0f;
        }
        if (kTrace) {
            __builtin_amdgcn_s_waitcnt(0x0070);                  // trace builds: stamp 7 = layer 1's operands have arrived
            PT(7);
        }
#pragma unroll
        for (int i = 0; i < kL1; ++i) {
            const int c0 = cw * 32 + 128 * i;
            if (c0 < a.h1) {
                f32x16 acc = {0};
                if (TR) {
                    const bool ok = c0 + col < a.h1;             // features beyond h1 (the k padding of layer 2) come out as zero
#pragma unroll
                    for (int u = 0; u < NU; ++u)
                        if (2 * u < a.d_in) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? wb[i][u] : 0.0f, xa[u], acc, 0, 0, 0);
                    acc = bias_mfma(acc, bias[i], lane);
                    store_tile_tr<true>(sh1 + (rh * 32 + col) * ld1 + c0, acc, lane);
                    continue;
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (2 * u < a.d_in) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u], wb[i][u], acc, 0, 0, 0);
                if (c0 + col < a.h1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sh1[(rh * 32 + cd_row(r, lane)) * ld1 + c0 + col] = fmaxf(acc[r] + bias[i], 0.0f);
                } else if (PACKED) {                             // the k padding of the last stage reads as zero
#pragma unroll
                    for (int r = 0; r < 16; ++r) sh1[(rh * 32 + cd_row(r, lane)) * ld1 + c0 + col] = 0.0f;
                }
            }
        }
    } else {
        for (int c0 = cw * 32; c0 < a.h1; c0 += 128) {
            const bool ok = c0 + col < a.h1;
            const float bias = ok ? b1[c0 + col] : 0.0f;
            f32x16 acc = {0};
            tile_gemm(acc, sx + rh * 32 * ldx, ldx, w1 + c0, a.h1, a.d_in, a.h1 - c0, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ok) sh1[(rh * 32 + cd_row(r, lane)) * ld1 + c0 + col] = fmaxf(acc[r] + bias, 0.0f);
                else if (PACKED) sh1[(rh * 32 + cd_row(r, lane)) * ld1 + c0 + col] = 0.0f;
            }
        }
    }
    PT(2);
    __syncthreads();
    PT(3);

    // the output stage's inputs (4 lanes per row), requested now so that their latency hides behind layers 2 + 3
    uint32_t tval = 0u, epval = 0u;
    float b3v[kQ];
    {
        const int e = e0 + (tid >> 2);
        if (tid < 4 * kRows && e < a.E && a.fin.sample_kind != 0) {
            if (a.fin.t_dev) tval = (uint32_t)a.fin.t_dev[e];
            if (a.fin.episode_dev) epval = (uint32_t)a.fin.episode_dev[e];
        }
#pragma unroll
        for (int i = 0; i < kQ; ++i) b3v[i] = (tid & 3) + 4 * i < a.nout ? b3[(tid & 3) + 4 * i] : 0.0f;
    }

    // ---- layers 2 + 3 fused over this wave's column chunks.  The chunk count is rarely a multiple of four (13 at
    // h = 400, 10 at h = 300, 7 at h = 200): dealt whole, one wave gets a chunk more than the others, runs 4 : 3 longer and
    // the other three wait for it holding their slots (round 3 trace: 140k against 110-116k cycles per wave, the matrix
    // pipe 54 % busy).  So only whole rounds of four chunks are dealt, and the leftover chunks are SPLIT BY K in one more
    // trip: one leftover chunk over the four waves (a quarter of the h1 range each), two leftover chunks over two pairs of
    // waves (half the range each); the partial tiles meet in LDS and one wave per chunk finishes it.
    f32x16 y = {0};
    f32x4 yn[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    float *st = sst + wave * 32 * kSt;
    const int nch = (a.h2 + 31) >> 5;
    // Measured at the C5 shard: one leftover (13 chunks at h = 400) -5.5 %, -4.3 % at C3; two leftovers (10 chunks at h = 300)
    // as pairs -8.7 % (round 4; as two quarter-split trips, i.e. four barriers, +1.5 %); three leftovers (7 chunks at
    // h = 200) lose either way (three quarter-split trips +19 %, a pair trip + a quarter trip +4 %) and are dealt whole
    const int rem = (nch & 3) == 3 ? 0 : (nch & 3);              // leftover chunks that are split (3: dealt whole)
    const int nch_even = rem ? (nch & ~3) : nch;
    const int split = rem == 2 ? 2 : 4;                          // waves sharing a leftover chunk
    const int ntrips = rem ? 1 : 0;
    // ONE loop over both kinds of trips (one inlined copy of each GEMM: a second copy of the layer-2 loop took the kernel
    // from 244 to 272 registers, i.e. from two workgroups per CU to one): first the whole rounds, then the leftover chunks
    const int rounds = (nch_even + 3) >> 2;                      // (whole dealing: the last round may be ragged)
    const int kpad = PACKED ? 16 * nst : a.h1;
    for (int it = 0; it < rounds + ntrips; ++it) {               // wave-uniform trip count (barriers inside)
        const bool left = it >= rounds;                          // leftover trip: this wave's K part of a leftover chunk
        const int part = cw & (split - 1);                       // this wave's K part
        const int lch = nch_even + (rem == 2 ? cw >> 1 : 0);
        const int c0 = (left ? lch : cw + 4 * it) * 32;
        if (!left && c0 >= nch_even * 32) continue;              // ragged last round of the whole dealing (no barrier in it)
        const int kq = PACKED ? 16 * ((nst + split - 1) / split) : (((a.h1 + split - 1) / split) + 1) & ~1;   // 1/split of K (even; PACKED: whole stages)
        const int kb = left ? min(part * kq, kpad) : 0, kn = left ? min(kq, kpad - kb) : kpad;
        const bool ok = c0 + col < a.h2;
        const float bias = ok ? b2[c0 + col] : 0.0f;             // issued before the k-loop, needed after it
        f32x16 acc = {0};
        // A single leftover chunk with at most 16 features (h2 = 400: the 13th chunk holds 16) is computed on the 16-column
        // instruction: D[16 features][16 rows] x two row tiles x 4 k per step = 8 instructions of 32 cycles per 16 k
        // instead of 8 of 64 on a tile whose other 16 feature rows are padding.  Lane (i = lane & 15, g = lane >> 4) takes
        // k = 16 s + 4 g + t at step t of stage s: its four weights are ONE 16-byte piece of the chunk's ordinary packed
        // stage ([q = g & 1][lane i + 32 (g >> 1)]: no other layout needed), its four h1 values one ds_read_b128 per row tile.
        const bool half16 = TR && left && split == 4 && a.h2 - c0 <= 16;     // wave-uniform
        if (half16) {
            const int i16 = lane & 15, g4 = lane >> 4;
            const f32x4 *Wp = reinterpret_cast<const f32x4 *>(w2 + (size_t)(c0 >> 5) * nst * 512) + (g4 & 1) * 64 + i16 + 32 * (g4 >> 1);
            const float *hr0 = sh1 + i16 * ld1 + 4 * g4, *hr1 = hr0 + 16 * ld1;
            const float bh = (part == 0 && c0 + i16 < a.h2) ? b2[c0 + i16] : 0.0f;
            f32x4 ha[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
            const int s0 = kb >> 4, sn = kn >> 4;
            if (sn > 0) {
                f32x4 wv = Wp[(size_t)s0 * 128];
                f32x4 x0 = *reinterpret_cast<const f32x4 *>(hr0 + 16 * s0), x1 = *reinterpret_cast<const f32x4 *>(hr1 + 16 * s0);
                for (int s1 = 0; s1 < sn; ++s1) {
                    const int sx2 = s0 + min(s1 + 1, sn - 1);                     // next stage (the last one re-reads itself: no branch)
                    const f32x4 wn = Wp[(size_t)sx2 * 128];
                    const f32x4 y0 = *reinterpret_cast<const f32x4 *>(hr0 + 16 * sx2), y1 = *reinterpret_cast<const f32x4 *>(hr1 + 16 * sx2);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        ha[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], x0[t], ha[0], 0, 0, 0);
                        ha[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], x1[t], ha[1], 0, 0, 0);
                    }
                    wv = wn; x0 = y0; x1 = y1;
                }
            }
            if (part == 0) {                                     // bias on the matrix pipe (k slot of lanes 0..15), as in bias_mfma
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    ha[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(g4 == 0 ? bh : 0.0f, g4 == 0 ? 1.0f : 0.0f, ha[rt], 0, 0, 0);
            }
            // D: register r of lane (i, g) = (feature 4 g + r, row 16 rt + i) -> this wave's partial tile [row][feature]
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) *reinterpret_cast<f32x4 *>(st + (rt * 16 + i16) * kSt + 4 * g4) = ha[rt];
            __syncthreads();
            if (part == 0) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    f32x4 *p = reinterpret_cast<f32x4 *>(st + (rt * 16 + i16) * kSt + 4 * g4);
                    f32x4 v = ((p[0] + p[32 * kSt / 4]) + p[2 * 32 * kSt / 4]) + p[3 * 32 * kSt / 4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.0f);
                    p[0] = v;
                    p[4] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};      // features 16 .. 31 of the staged chunk read as zero in layer 3
                }
            }
        } else
        if (PACKED) {
            if (kn > 0)
                tile_gemm_packed<POLICY_SETS, TR>(acc, sh1 + (rh * 32 + col) * ld1 + 8 * (lane >> 5),
                                    reinterpret_cast<const f32x4 *>(w2 + (size_t)(c0 >> 5) * nst * 512) + lane, kb >> 4, kn >> 4);
        } else if (kn > 0)
            tile_gemm(acc, sh1 + rh * 32 * ld1 + kb, ld1, w2 + c0 + (size_t)kb * a.h2, a.h2, kn, a.h2 - c0, lane);
        bool l3 = true;                                          // this wave feeds the chunk to layer 3
        if (half16) l3 = part == 0;
        else if (TR) {                                           // (packed W2 and the masked bias are zero beyond h2)
            if (!left || part == 0) acc = bias_mfma(acc, bias, lane);
            if (!left) store_tile_tr<true>(st + col * kSt, acc, lane);
            else {
                store_tile_tr<false>(st + col * kSt, acc, lane);                     // this wave's partial tile (part 0: + bias)
                __syncthreads();
                l3 = part == 0;                                  // one wave per chunk adds the partials in a fixed order
                if (l3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 *p = reinterpret_cast<f32x4 *>(st + col * kSt + 8 * q + 4 * (lane >> 5));
                        f32x4 v = p[0] + p[32 * kSt / 4];
                        if (split == 4) v = (v + p[2 * 32 * kSt / 4]) + p[3 * 32 * kSt / 4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.0f);
                        p[0] = v;
                    }
                }
            }
        } else if (!left) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[cd_row(r, lane) * kSt + col] = ok ? fmaxf(acc[r] + bias, 0.0f) : 0.0f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[cd_row(r, lane) * kSt + col] = acc[r];     // this wave's partial tile
            __syncthreads();
            l3 = part == 0;                                      // one wave per chunk adds the partials in a fixed order
            if (l3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = cd_row(r, lane) * kSt + col;
                    float v = st[o] + st[32 * kSt + o];
                    if (split == 4) v = (v + st[2 * 32 * kSt + o]) + st[3 * 32 * kSt + o];
                    st[o] = ok ? fmaxf(v + bias, 0.0f) : 0.0f;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (l3) {
            if (NARROW) layer3_narrow(yn, st, w3 + (size_t)c0 * a.nout, a.nout, min(32, a.h2 - c0), lane);
            else tile_gemm(y, st, kStW, w3 + (size_t)c0 * a.nout, a.nout, min(32, a.h2 - c0), a.nout, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (left) __syncthreads();                               // the partial regions are free again
    }
    PT(4);
    if (NARROW) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(mt * 16 + 4 * (lane >> 4) + r) * kSt + (lane & 15)] = yn[mt][r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[cd_row(r, lane) * kSt + col] = y[r];   // this wave's partial outputs
    }
    __syncthreads();
    PT(5);
    if (kTrace && a.trace && lane == 0 && wave >= 2)             // (slot 6 of waves 2, 3 is free: the finish stamp is waves 0, 1)
        a.trace[((size_t)blockIdx.x * 4 + wave) * 8 + 6] = (long long)(__builtin_amdgcn_s_memrealtime() - rt0);

    // ---- output activation + sampling: four lanes per env row
    if (tid < 4 * kRows) {
        const int row = tid >> 2, part = tid & 3;
        const int e = e0 + row;
        if (e >= a.E) return;
        const int rhh = row >> 5, rr = row & 31;
        float yv[kQ];
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = part + 4 * i;
            float v = 0.0f;
            if (j < a.nout) {
                v = b3v[i];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += sst[((rhh * 4 + w) * 32 + rr) * kSt + j];
            }
            yv[i] = v;
        }
        finish_quad(a.fin, yv, e, agent, part, tval, epval);
        PT(6);
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 variant (opt-in): weights and activations in bfloat16, float32 accumulation, on
// v_mfma_f32_32x32x16_bf16 (16x the float32 matrix rate).  Formulated transposed -- D[feature][env row] =
// W^T (A operand, pre-packed per fragment on the host, one 16-byte load per lane) x activations (B operand,
// [row][k]) -- so that a lane's accumulator registers are features of ONE env row.
// One workgroup = 64 env rows (2 row tiles sharing every weight fragment) of one agent, 4 waves; wave w owns
// feature chunks w, w+4, ... of every layer.  <= 168 VGPRs and ~48 KiB of LDS at h = 300: three workgroups per CU.
//   layer 1: B = x tile (LDS), relu -> h1 tile in LDS as bf16 [64][h1 pad + 8]
//   layer 2: B = h1 tile (LDS, one ds_read_b128 per lane and fragment, fetched one k-step ahead), A = weight
//            fragments streaming from L2 through a register ring that runs kRing k-steps ahead and across chunk
//            borders (pinned with sched_barrier: the scheduler otherwise sinks every load next to its use).
//            The kernel is occupancy-bound, not bandwidth-bound: at 3 workgroups per CU the MFMA pipe, the LDS
//            and the L2 each sit near 30 %; keeping part of the h1 tile in registers (measured: up to 128 VGPRs)
//            cut LDS traffic but cost the third workgroup and lost 20 %.
//   layer 3: the accumulator layout of a finished layer-2 chunk (lane = env row, registers = features
//            (r&3) + 8 (r>>2) + 4 (lane>>5)) IS a valid B operand if the k order of W3's fragments is permuted
//            to match (done once on the host, see dronesim.h) -- bias + relu + convert in registers, no LDS.
//   the four waves' layer-3 partials are summed through LDS, then activation + sampling, one thread per row.
// LDS row strides are odd multiples of 16 bytes (conflict-free b128 reads).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kRowsB = 64, kTiles = 2;   // 64 env rows (2 row tiles) per workgroup
constexpr int kLdx = 24;                 // bf16 per row of the x tile

struct MArgsB {
    int E, N, d_in, h1, h2, nc1, nc2, ks1;
    const float *x, *b1, *b2, *b3;
    const bf16x8 *w1p, *w2p, *w3p;       // [agent][chunk][k-step][64 lanes] fragments
    FinishArgs fin;
    long long *trace;                    // developer trace builds only (NULL otherwise)
};

// relu(acc + bias) of one 32-feature chunk -> bf16 rows [row][feature].  `bias` points at the chunk's 32 biases
// in LDS (zero beyond the layer width; the padded weights are zero there too, so those features come out 0).
__device__ __forceinline__ void store_chunk(const f32x16 (&acc)[kTiles], const float *bias, __bf16 *dst, int ld, int lane)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int fl = 8 * q + 4 * (lane >> 5);            // local feature of register 4q (C/D layout rows)
        const float4 bv = *reinterpret_cast<const float4 *>(bias + fl);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            bf16x4 p;
            p[0] = (__bf16)fmaxf(acc[t][4 * q + 0] + bv.x, 0.0f);
            p[1] = (__bf16)fmaxf(acc[t][4 * q + 1] + bv.y, 0.0f);
            p[2] = (__bf16)fmaxf(acc[t][4 * q + 2] + bv.z, 0.0f);
            p[3] = (__bf16)fmaxf(acc[t][4 * q + 3] + bv.w, 0.0f);
            *reinterpret_cast<bf16x4 *>(dst + (t * 32 + (lane & 31)) * ld + fl) = p;
        }
    }
}

template <int NC1>                       // 32-feature chunks of the first hidden layer (h1 <= 32 NC1)
__global__ void __launch_bounds__(256, NC1 <= 8 ? 4 : 3) mlp3_bf16_kernel(const float *x, int E, int N, int d_in, const MArgsB rest)
{
    MArgsB a = rest;                     // leading scalars are preloaded into SGPRs at wave launch (csrc/Makefile)
    a.x = x; a.E = E; a.N = N; a.d_in = d_in;
    // occupancy first: h1 <= 256 leaves LDS for four workgroups per CU, which needs <= 128 VGPRs (ring of 4);
    // wider layers fit three (two beyond h1 = 352), where 168 VGPRs allow weight fragments 8 k-steps ahead
    constexpr int kRing = NC1 <= 8 ? 4 : 8;
    constexpr int KS2 = 2 * NC1;                         // k-steps (of 16) of layer 2
    constexpr int L1C = (NC1 + 3) / 4;                   // layer-1 chunks per wave (upper bound)
    constexpr int kMaxChunks = 4;                        // layer-2 chunks per wave: h2 <= 512
    constexpr int ld1 = NC1 * 32 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int agent, row_block;
    xcd_work_item((a.E + kRowsB - 1) / kRowsB, agent, row_block);
    const int e0 = row_block * kRowsB;
    const int nb = (NC1 + a.nc2) * 32;
    __bf16 *sx = reinterpret_cast<__bf16 *>(smem);                 // [64][24]
    float *sbias = reinterpret_cast<float *>(sx + kRowsB * kLdx);  // b1 | b2 (zero padded to chunks) | b3 (32)
    __bf16 *sh1 = reinterpret_cast<__bf16 *>(sbias + nb + 32);     // [64][ld1]; later f32 partials [4][64][33]
    float *spart = reinterpret_cast<float *>(sh1);
    PT(0);

    // ---- everything that does not depend on LDS is requested up front: x, biases, the sampling counters of
    //      this thread's row, this wave's layer-1 fragments and the head of its layer-2 weight stream
    float xv[4];                                                   // x tile: thread = (row, 4 of 16 k slots)
    {
        const int r = tid >> 2, c0 = (tid & 3) * 4, e = e0 + r;
        const float *xr = a.x + ((size_t)e * a.N + agent) * a.d_in;
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = (e < a.E && c0 + j < a.d_in) ? xr[c0 + j] : 0.0f;   // zero padded to k = 16
    }
    uint32_t tval = 0, epval = 0;
    if (e0 + (tid >> 2) < a.E && a.fin.sample_kind != 0) {          // of the row this thread finishes (4 lanes per row)
        if (a.fin.t_dev) tval = (uint32_t)a.fin.t_dev[e0 + (tid >> 2)];
        if (a.fin.episode_dev) epval = (uint32_t)a.fin.episode_dev[e0 + (tid >> 2)];
    }
    float bv[5];                                                   // (16 + 16) * 32 + 32 <= 5 * 256 bias words
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int idx = tid + 256 * u;
        float v = 0.0f;
        if (idx < NC1 * 32) { if (idx < a.h1) v = a.b1[(size_t)agent * a.h1 + idx]; }
        else if (idx < nb) { if (idx - NC1 * 32 < a.h2) v = a.b2[(size_t)agent * a.h2 + idx - NC1 * 32]; }
        else if (idx - nb < a.fin.nout) v = a.b3[(size_t)agent * a.fin.nout + idx - nb];
        bv[u] = v;
    }
    bf16x8 w1f[L1C];
#pragma unroll
    for (int i = 0; i < L1C; ++i)
        if (wave + 4 * i < NC1) w1f[i] = a.w1p[((size_t)agent * NC1 + wave + 4 * i) * 64 + lane];
    const bf16x8 *w2a = a.w2p + (size_t)agent * a.nc2 * KS2 * 64 + lane;
    const bf16x8 *w3a = a.w3p + (size_t)agent * a.nc2 * 2 * 64 + lane;
    bf16x8 ring[kRing];
#pragma unroll
    for (int g = 0; g < kRing; ++g) {                              // stream position g = chunk slot * KS2 + k-step
        const int c = wave + 4 * (g / KS2);
        if (g / KS2 < kMaxChunks && c < a.nc2) ring[g] = w2a[((size_t)c * KS2 + g % KS2) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);                             // all requests are out before anything waits
    {
        bf16x4 p;
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = (__bf16)xv[j];
        *reinterpret_cast<bf16x4 *>(sx + (tid >> 2) * kLdx + (tid & 3) * 4) = p;
    }
#pragma unroll
    for (int u = 0; u < 5; ++u)
        if (tid + 256 * u < nb + 32) sbias[tid + 256 * u] = bv[u];
    __syncthreads();
    PT(1);

    // ---- layer 1 -> sh1 (bf16): wave w owns feature chunks w, w+4, ...
    {
        const __bf16 *xrow = sx + (lane & 31) * kLdx + 8 * (lane >> 5);
        bf16x8 bx[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) bx[t] = *reinterpret_cast<const bf16x8 *>(xrow + t * 32 * kLdx);
#pragma unroll
        for (int i = 0; i < L1C; ++i) {
            const int c = wave + 4 * i;
            if (c < NC1) {
                f32x16 acc[kTiles] = {};
#pragma unroll
                for (int t = 0; t < kTiles; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[i], bx[t], acc[t], 0, 0, 0);
                store_chunk(acc, sbias + c * 32, sh1 + c * 32, ld1, lane);
            }
        }
    }
    PT(2);
    __syncthreads();
    PT(3);

    // ---- layers 2 + 3 fused over this wave's feature chunks
    const __bf16 *brow = sh1 + (lane & 31) * ld1 + 8 * (lane >> 5);
    f32x16 y[kTiles] = {};
    PT(4);
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = wave + 4 * i;
        if (c < a.nc2) {                                           // wave-uniform
            bf16x8 w3f[2];
            w3f[0] = w3a[(size_t)(2 * c) * 64];
            w3f[1] = w3a[(size_t)(2 * c + 1) * 64];
            f32x16 acc[kTiles] = {};
            bf16x8 bl[2][kTiles];                                   // h1 fragments, fetched one k-step ahead
#pragma unroll
            for (int t = 0; t < kTiles; ++t) bl[0][t] = *reinterpret_cast<const bf16x8 *>(brow + t * 32 * ld1);
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const int g = i * KS2 + s;
                if (s + 1 < KS2) {
#pragma unroll
                    for (int t = 0; t < kTiles; ++t)
                        bl[(s + 1) & 1][t] = *reinterpret_cast<const bf16x8 *>(brow + t * 32 * ld1 + (s + 1) * 16);
                }
#pragma unroll
                for (int t = 0; t < kTiles; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[g % kRing], bl[s & 1][t], acc[t], 0, 0, 0);
                const int g2 = g + kRing, i2 = g2 / KS2;           // refill the slot just consumed
                if (i2 < kMaxChunks) {
                    const int c2 = wave + 4 * i2;
                    if (c2 < a.nc2) ring[g % kRing] = w2a[((size_t)c2 * KS2 + g2 % KS2) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);                 // keep the ring kRing steps ahead: the scheduler
            }                                                      // otherwise sinks each load next to its use
            // bias + relu in registers; registers 8 s .. 8 s + 7 of a lane are the k slots of layer-3 k-step s
            const float *bc = sbias + (NC1 + c) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float4 b0 = *reinterpret_cast<const float4 *>(bc + 16 * s);
                const float4 b1 = *reinterpret_cast<const float4 *>(bc + 16 * s + 8);
#pragma unroll
                for (int t = 0; t < kTiles; ++t) {
                    bf16x8 p;
                    p[0] = (__bf16)fmaxf(acc[t][8 * s + 0] + b0.x, 0.0f);
                    p[1] = (__bf16)fmaxf(acc[t][8 * s + 1] + b0.y, 0.0f);
                    p[2] = (__bf16)fmaxf(acc[t][8 * s + 2] + b0.z, 0.0f);
                    p[3] = (__bf16)fmaxf(acc[t][8 * s + 3] + b0.w, 0.0f);
                    p[4] = (__bf16)fmaxf(acc[t][8 * s + 4] + b1.x, 0.0f);
                    p[5] = (__bf16)fmaxf(acc[t][8 * s + 5] + b1.y, 0.0f);
                    p[6] = (__bf16)fmaxf(acc[t][8 * s + 6] + b1.z, 0.0f);
                    p[7] = (__bf16)fmaxf(acc[t][8 * s + 7] + b1.w, 0.0f);
                    y[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3f[s], p, y[t], 0, 0, 0);
                }
            }
        }
    }
    PT(5);
    __syncthreads();                                               // everyone is done reading sh1
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            spart[((size_t)wave * kRowsB + t * 32 + (lane & 31)) * 33 + cd_row(r, lane)] = y[t][r];
    __syncthreads();
    PT(6);

    {
        const int row = tid >> 2, part = tid & 3;
        const int e = e0 + row;
        if (e >= a.E) return;
        float yv[kQ];
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = part + 4 * i;
            float v = 0.0f;
            if (j < a.fin.nout) {
                v = sbias[nb + j];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += spart[((size_t)w * kRowsB + row) * 33 + j];
            }
            yv[i] = v;
        }
        finish_quad(a.fin, yv, e, agent, part, tval, epval);
        PT(7);
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 variant: float32-ACCURATE results on the bf16 matrix instructions.  Every float32 operand is split by
// truncation into three bf16 parts, x = hi + mid + lo EXACTLY (8 + 8 + 8 significant bits), and a product is the sum
// of the six partial products of weight at least 2^-16:  hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid  (the dropped
// ones are below 2^-24 of the product), accumulated in float32, smallest first.  Six v_mfma_f32_32x32x16_bf16 per
// k-step of 16 cost 6/16 of the matrix time of the eight v_mfma_f32_32x32x2_f32 they replace; the result meets the
// reference's own torch modules to the 1e-5 bar like the exact-f32 kernel (tools/split_bf16_emulation.py: a two-part
// split with three products does NOT -- 4-5x over the bar on the Gaussian and critic heads).
// Weights are split and packed per fragment on the host (policies.py) into one consumption-ordered stream per
// (agent, wave) (include/dronesim.h).
// The three layers are fused in REGISTERS, transposed like the bf16 kernel (D[feature][env row]): one workgroup =
// 64 env rows (two 32-row tiles) of one agent, 4 waves; wave w owns the output chunks w, w+4, ... of layer 2 and keeps
// their accumulators for the whole launch while it streams over the first hidden layer chunk by chunk -- each chunk of
// h1 is computed where it is consumed (layer 1 has K = d_in <= 16: one k-step), relu + split happen on the accumulator
// registers (the bias went in as their initial value), whose layout IS a valid B operand once the k order of W2 / W3
// is permuted to match on the host ("accumulator" order, include/dronesim.h).  No activation ever touches LDS; the
// only barriers are around the final sum of the four waves' layer-3 partials.  Non-finite activations are outside the
// split's domain (inf - inf).
// The kernel is software-pipelined by hand inside each wave (mlp3_bf16x3_kernel: `stage`): between the twelve matrix
// instructions of a stage sit the LDS reads of the NEXT stage's fragments, the DMA requests of the stage four ahead and
// one row tile's share of the relu + split that the stages after the next layer-1 step will consume -- a wave never
// waits on L2, LDS or its own vector work with the matrix pipe idle.  Measured on the MI355X (tools/trace_x3.py,
// tools/micro/mfma_dma.hip): the matrix pipe is ~70 % busy at the ACTUAL shader clock, and that clock is what gives:
// 2.07 GHz with the weight stream switched off, 1.6 GHz with it on (power management), against 2.4 GHz nominal --
// ring depth 3 / 4 / 5, spreading the roles over the SIMDs and L1-resident weights all leave the time unchanged.
// TILES row tiles of 32 env rows per wave: every weight fragment loaded feeds all of them.  2 = 64-row workgroups, two per
// CU (256 registers per wave) -- the instantiated one; 4 = 128-row workgroups, one per CU, would halve the weight bytes
// per matrix instruction with the 256 accumulator registers of a wave in the AGPR half of its 512 (see launch below).
constexpr int kStreamPadX = 8;             // zero stages behind every stream: the run-ahead requests of the deepest ring
__host__ __device__ constexpr int split_ring_depth(int tiles) { return tiles <= 2 ? 4 : 8; }   // stages of the per-wave weight ring in LDS
template <int N, int I = 0, class F> __device__ __forceinline__ void for_each_slot(F &f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); for_each_slot<N, I + 1>(f); }
}

struct MArgsX {
    int E, N, d_in, h1, h2, nc1, nc2;
    const float *x, *b1, *b2, *b3;
    const float *wscale;           // f16x2: [N][3] power-of-two factors the packed weights were multiplied by, or NULL
    const char *ws;                // the per-(agent, wave) fragment streams
    int stages;                    // stages per stream (padded)
    FinishArgs fin;
    long long *trace;              // developer trace builds only (NULL otherwise)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int P> struct Parts { u32x4 p[P]; };     // the P parts of a lane's 8 k-slots (two 16-bit values per dword, low half first)

__device__ __forceinline__ unsigned upper_halves(unsigned odd, unsigned even)      // -> {even.hi16 (low), odd.hi16 (high)}
{
    return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}

// ---- the two split schemes.  A scheme names its parts (part 0 = the leading one), the partial products it keeps, in
//      issue order -- smallest first as far as the register hand-over of the weight fragments allows: a fragment part is
//      re-loaded with the NEXT stage's bytes right behind its last product -- and how two float32 values become one
//      dword of every part.
struct SchemeBf16x3 {                                  // v = hi + mid + lo exactly (truncation), products >= 2^-16
    static constexpr int kParts = 3, kProducts = 6;
    static constexpr bool kScaled = false;             // (an exact split: the weights' magnitude does not matter)
    __device__ static constexpr int w_part(int q) { constexpr int t[6] = {2, 1, 0, 1, 0, 0}; return t[q]; }   // lo.hi mid.mid hi.lo
    __device__ static constexpr int b_part(int q) { constexpr int t[6] = {0, 1, 2, 0, 1, 0}; return t[q]; }   // mid.hi hi.mid hi.hi
    __device__ static constexpr int last_use(int p) { constexpr int t[3] = {5, 3, 0}; return t[p]; }
    __device__ static constexpr int request_slot(int p) { constexpr int t[3] = {3, 6, 9}; return t[p]; }
    __device__ static __forceinline__ f32x16 mfma(const u32x4 &w, const u32x4 &b, const f32x16 &acc)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    // the upper 16 bits of a float32 ARE a bf16: 5 vector instructions per value with the relu, 3 per pair to pack
    template <bool RELU> __device__ static __forceinline__ void split_pair(float v0, float v1, unsigned (&d)[3])
    {
        if (RELU) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
        const unsigned h0 = __float_as_uint(v0), h1 = __float_as_uint(v1);
        const float r0 = v0 - __uint_as_float(h0 & 0xffff0000u), r1 = v1 - __uint_as_float(h1 & 0xffff0000u);
        const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1);
        const float l0 = r0 - __uint_as_float(m0 & 0xffff0000u), l1 = r1 - __uint_as_float(m1 & 0xffff0000u);
        d[0] = upper_halves(h1, h0); d[1] = upper_halves(m1, m0); d[2] = upper_halves(__float_as_uint(l1), __float_as_uint(l0));
    }
};
struct SchemeF16x2 {                                   // v = hi + lo to 2^-22 (float16 parts, subnormals honoured by the
    static constexpr int kParts = 2, kProducts = 3;    // matrix unit), products hi.hi, hi.lo, lo.hi; |v| < 65504
    // the packed weights of a layer carry a power-of-two factor (DroneMlpBf16.wscale) that keeps their low parts out of the
    // float16 subnormals; a layer's accumulators are multiplied by its inverse where they are split for the next layer
    static constexpr bool kScaled = true;
    __device__ static constexpr int w_part(int q) { constexpr int t[3] = {1, 0, 0}; return t[q]; }            // lo.hi hi.lo hi.hi
    __device__ static constexpr int b_part(int q) { constexpr int t[3] = {0, 1, 0}; return t[q]; }
    __device__ static constexpr int last_use(int p) { constexpr int t[2] = {2, 0}; return t[p]; }
    __device__ static constexpr int request_slot(int p) { constexpr int t[2] = {2, 4}; return t[p]; }
    __device__ static __forceinline__ f32x16 mfma(const u32x4 &w, const u32x4 &b, const f32x16 &acc)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    template <bool RELU> __device__ static __forceinline__ void split_pair(float v0, float v1, unsigned (&d)[2])
    {
        if (RELU) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
        const f32x2v v = {v0, v1};                                             // round to nearest twice (v_cvt_pk_f16_f32):
        const f16x2v h = __builtin_convertvector(v, f16x2v);                   // |v - hi| <= 2^-11 |v|, v - hi exact in float32,
        const f16x2v l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2v), f16x2v);   // |v - hi - lo| <= 2^-22 |v|
        d[0] = __builtin_bit_cast(unsigned, h); d[1] = __builtin_bit_cast(unsigned, l);
    }
};

// relu + split of half an accumulator tile (registers 8 HALF .. 8 HALF + 7 = the k slots of k-step HALF of the next
// layer's B operand), dealt out over the slots between a stage's matrix instructions: one call per slot.
template <class S, int HALF, int TILES>
struct SplitJob {
    static constexpr int kSlots = TILES * S::kProducts;
    static constexpr int kStride = (kSlots - 2) / 4 > 0 ? (kSlots - 2) / 4 : 1;    // pairs behind slots 1, 1 + stride, ...
    const f32x16 &src;
    Parts<S::kParts> &dst;
    const float mul;                                                               // S::kScaled: 1 / (the producing layer's weight factor)
    __device__ __forceinline__ SplitJob(const f32x16 &s, Parts<S::kParts> &d, float m = 1.0f) : src(s), dst(d), mul(m) {}
    template <int Q> __device__ __forceinline__ void pair()                        // values 2 Q, 2 Q + 1 -> dword Q
    {
        unsigned d[S::kParts];
        if constexpr (S::kScaled) S::template split_pair<true>(src[8 * HALF + 2 * Q] * mul, src[8 * HALF + 2 * Q + 1] * mul, d);
        else
        S::template split_pair<true>(src[8 * HALF + 2 * Q], src[8 * HALF + 2 * Q + 1], d);
#pragma unroll
        for (int p = 0; p < S::kParts; ++p) dst.p[p][Q] = d[p];
    }
    template <int SLOT> __device__ __forceinline__ void slot()
    {
        if constexpr (SLOT >= 1 && (SLOT - 1) % kStride == 0 && (SLOT - 1) / kStride < 4) pair<(SLOT - 1) / kStride>();
    }
    __device__ __forceinline__ void all() { pair<0>(); pair<1>(); pair<2>(); pair<3>(); }
};
struct NoJob { template <int SLOT> __device__ __forceinline__ void slot() {} };
#define SplitJobInStage SplitJob

// LDS reads the compiler must NOT see: hipcc orders every LDS read it can see behind ALL pending global_load_lds of
// the wave (s_waitcnt vmcnt(0)), which would drain the weight ring; the bytes read this way (biases, the x operand)
// were written before the first DMA was issued.
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// The 32 biases of a chunk in accumulator layout (register 4 q + j of a lane = feature 8 q + 4 (lane >> 5) + j): the
// initial value of the chunk's accumulators.  `bias` = the chunk's 32 floats in LDS.
__device__ __forceinline__ f32x16 bias_tile(const float *bias, int lane)
{
    const uint32_t addr = lds_addr(bias + 4 * (lane >> 5));
    float4 q0, q1, q2, q3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                 "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr) : "memory");
    return f32x16{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
}
template <int P> __device__ __forceinline__ Parts<P> parts_from_lds(const char *p)      // parts 1 KiB apart
{
    Parts<P> r;
    if constexpr (P == 3)
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]) : "v"(lds_addr(p)) : "memory");
    else
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r.p[0]), "=&v"(r.p[1]) : "v"(lds_addr(p)) : "memory");
    return r;
}

#define X3_PIN() __builtin_amdgcn_sched_barrier(0)
#define X3_RING_READ(dst, p) dst = *reinterpret_cast<const u32x4 *>(p)

template <class S, int TILES>
__global__ void __launch_bounds__(256, TILES <= 2 ? 2 : 1) mlp3_split_kernel(const float *x, int E, int N, int d_in, const MArgsX rest)
{
    constexpr int P = S::kParts, kStageBytes = P * 1024;
    constexpr int kTilesX = TILES, kRowsX = 32 * TILES, kRingX = split_ring_depth(TILES);
    MArgsX a = rest;
    a.x = x; a.E = E; a.N = N; a.d_in = d_in;
    constexpr int kMaxChunks = 4;                        // layer-2 chunks per wave: h2 <= 512
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int agent, row_block;
    xcd_work_item((a.E + kRowsX - 1) / kRowsX, agent, row_block);
    const int e0 = row_block * kRowsX;
    const int NC1 = a.nc1;
    const int nb = (a.nc1 + a.nc2) * 32;
    // S::kScaled: the power-of-two factors of this agent's packed layers and their (exact) inverses
    float ws1 = 1.0f, ws2 = 1.0f, wi1 = 1.0f, wi2 = 1.0f, wi3 = 1.0f;
    if (S::kScaled && a.wscale != nullptr) {
        const float *wsp = a.wscale + 3 * (size_t)agent;
        const float ws3 = wsp[2];
        ws1 = wsp[0]; ws2 = wsp[1];
        // 1 / factor by v_rcp_f32: EXACT for the powers of two include/dronesim.h asks for (a device-side pointer cannot be checked by
        // the host entry point; exponent arithmetic on the bit pattern, as in round 5, was only right for normal powers of two)
        wi1 = __builtin_amdgcn_rcpf(ws1);
        wi2 = __builtin_amdgcn_rcpf(ws2);
        wi3 = __builtin_amdgcn_rcpf(ws3);
    }
    if (kTrace && a.trace && lane == 0) {                          // shader clock and the 100 MHz clock at entry
        a.trace[((size_t)blockIdx.x * 4 + wave) * 8 + 0] = __builtin_amdgcn_s_memtime();
        a.trace[((size_t)blockIdx.x * 4 + wave) * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    float *sbias = reinterpret_cast<float *>(smem);                // b1 | b2 (zero padded to chunks) | b3 (32)
    char *sxb = reinterpret_cast<char *>(sbias + nb + 32);         // x operand [tile][part][lane] x 16 B
    float *spart = reinterpret_cast<float *>(sxb + kTilesX * kStageBytes);   // [4 waves][kRowsX rows][33], shares LDS with the rings

    // ---- the x operand: row 32 t + (lane & 31), inputs 8 (lane >> 5) .. + 7, split once, kept in LDS (every wave uses it)
    if (wave < kTilesX) {
        const int t = wave;
        const int e = e0 + 32 * t + (lane & 31), k0 = 8 * (lane >> 5);
        const float *xr = a.x + ((size_t)min(e, a.E - 1) * a.N + agent) * a.d_in;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // clamped address, value masked with an AND: written as `cond ? xv : 0` hipcc turns the select into a branch
            // around the load and waits for each of the eight loads in turn (eight round trips in series at the head of
            // every workgroup; found in the ISA in round 4)
            const float xv = xr[min(k0 + j, a.d_in - 1)];
            v[j] = __uint_as_float(__float_as_uint(xv) & ((e < a.E && k0 + j < a.d_in) ? 0xffffffffu : 0u));
        }
        Parts<P> xp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned d[P];
            S::template split_pair<false>(v[2 * q], v[2 * q + 1], d);              // (no relu on the inputs)
#pragma unroll
            for (int p = 0; p < P; ++p) xp.p[p][q] = d[p];
        }
        u32x4 *dp = reinterpret_cast<u32x4 *>(sxb + t * kStageBytes) + lane;
#pragma unroll
        for (int p = 0; p < P; ++p) dp[64 * p] = xp.p[p];
    }
    uint32_t tval[kRowsX / 64], epval[kRowsX / 64];                // step / episode counters of the rows this thread finishes
#pragma unroll
    for (int r = 0; r < kRowsX / 64; ++r) {                        // (the sampling stream's position; 4 lanes per row)
        const int e = e0 + 64 * r + (tid >> 2);
        tval[r] = epval[r] = 0;
        if (e < a.E && a.fin.sample_kind != 0) {
            if (a.fin.t_dev) tval[r] = (uint32_t)a.fin.t_dev[e];
            if (a.fin.episode_dev) epval[r] = (uint32_t)a.fin.episode_dev[e];
        }
    }
    {                                                              // biases -> LDS, zero padded: nb + 32 <= 1056 floats.  All
        constexpr int kIt = 5;                                     // loads first, none behind a branch (same reason as above)
        float bv[kIt];
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int idx = tid + 256 * it, j2 = idx - a.nc1 * 32, j3 = idx - nb;
            const bool in1 = idx < a.nc1 * 32, in2 = idx < nb;
            const float *p = in1 ? a.b1 + (size_t)agent * a.h1 + min(idx, a.h1 - 1)
                           : in2 ? a.b2 + (size_t)agent * a.h2 + min(j2, a.h2 - 1)
                                 : a.b3 + (size_t)agent * a.fin.nout + min(max(j3, 0), a.fin.nout - 1);
            const bool ok = in1 ? idx < a.h1 : in2 ? j2 < a.h2 : j3 < a.fin.nout;
            bv[it] = __uint_as_float(__float_as_uint(*p) & (ok ? 0xffffffffu : 0u));
            if (S::kScaled) bv[it] *= in1 ? ws1 : in2 ? ws2 : 1.0f;    // b1, b2 start accumulators of SCALED products; b3 is added at the end
        }
#pragma unroll
        for (int it = 0; it < kIt; ++it)
            if (tid + 256 * it < nb + 32) sbias[tid + 256 * it] = bv[it];
    }
    __syncthreads();

    const int nmine = (a.nc2 - wave + 3) / 4;                      // output chunks of this wave (wave-uniform): wave + 4 i
    f32x16 acc2[kMaxChunks][kTilesX];
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {                         // start from the layer-2 biases (zero padded)
        const f32x16 b = bias_tile(sbias + (a.nc1 + min(wave + 4 * i, a.nc2 - 1)) * 32, lane);
#pragma unroll
        for (int t = 0; t < kTilesX; ++t) acc2[i][t] = b;
    }
    f32x16 y[kTilesX];
#pragma unroll
    for (int t = 0; t < kTilesX; ++t) y[t] = f32x16{};

    // ---- The weight fragments of this wave form ONE linear stream of stages (the P parts of one 32-feature chunk x
    //      16 k slots, 1 KiB each), laid out by the host in the order the matrix instructions want them
    //      (include/dronesim.h):
    //          W1(0), W2(0,0,*) | W1(1), W2(0,1,*), W2(1,0,*) | W1(2), W2(1,1,*), W2(2,0,*) | ... | W3(*)
    //      (W2(c1, ss, i): k-step 2 c1 + ss of this wave's i-th output chunk).  They travel global -> LDS by DMA
    //      (global_load_lds, 1 KiB per instruction) into a ring PRIVATE to the wave, requested kRingX stages before
    //      their matrix instructions and picked up part by part during the PREVIOUS stage's matrix instructions, into
    //      the registers that stage has just finished with: no copies, no exposed LDS or L2 latency.  The ring needs no
    //      barrier (one wave writes and reads it); a read is ordered behind its DMA by the counted s_waitcnt vmcnt.
    //      The stream is padded by kRingX stages, so the producer never needs to know where it ends.
    char *ring = reinterpret_cast<char *>(spart) + (size_t)wave * kRingX * kStageBytes;
    const char *gp = a.ws + (((size_t)agent * 4 + wave) * a.stages * P * 64 + lane) * 16;
    int pslot = 0;
    auto request_part = [&](int p) {                               // 1 KiB of the stage kRingX ahead
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + p * 1024),
                                         (__attribute__((address_space(3))) void *)(ring + pslot * kStageBytes + p * 1024), 16, 0, 0);
    };
    auto request_done = [&]() {
        gp += kStageBytes;
        pslot = pslot + 1 == kRingX ? 0 : pslot + 1;
    };
    u32x4 wf[P];                                                   // the current stage's fragments, part 0 first
    int nslot = 1;                                                 // ring slot of the NEXT stage
    // one stage: acc[t] += W * B[t], the scheme's products on both tiles; slot s = the gap behind matrix instruction s:
    // `job` is offered every slot, the ring reads and the DMA requests sit where the scheme puts them
    auto stage = [&](f32x16 (&acc)[kTilesX], const Parts<P> (&b)[kTilesX], auto &job) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P * (kRingX - 2)) : "memory");    // the NEXT stage has landed
        const char *np = ring + nslot * kStageBytes + lane * 16;
        X3_PIN();
        auto slot_work = [&](auto SLOT) {
            constexpr int s = decltype(SLOT)::value, q = s / TILES, t = s % TILES;
            acc[t] = S::mfma(wf[S::w_part(q)], b[t].p[S::b_part(q)], acc[t]);
            X3_PIN();
            if constexpr (t == TILES - 1) {
#pragma unroll
                for (int p = 0; p < P; ++p)
                    if (S::last_use(p) == q) X3_RING_READ(wf[p], np + p * 1024);
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
                if (S::request_slot(p) * TILES / 2 == s) { request_part(p); if (p == P - 1) request_done(); }
            job.template slot<s>();
            X3_PIN();
        };
        for_each_slot<S::kProducts * TILES>(slot_work);
        nslot = nslot + 1 == kRingX ? 0 : nslot + 1;
        X3_PIN();
    };

    if (nmine > 0) {
#pragma unroll 1
        for (int j = 0; j < kRingX; ++j) {
#pragma unroll
            for (int p = 0; p < P; ++p) request_part(p);
            request_done();
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P * (kRingX - 1)) : "memory");
#pragma unroll
        for (int p = 0; p < P; ++p) wf[p] = *reinterpret_cast<const u32x4 *>(ring + p * 1024 + lane * 16);

        NoJob nojob;
        f32x16 a1[kTilesX];
        Parts<P> hB0[kTilesX], hB1[kTilesX];                       // layer-2 operands of k-steps 2 c1 and 2 c1 + 1
        {                                                          // layer 1, chunk 0
            Parts<P> xB[kTilesX];
#pragma unroll
            for (int t = 0; t < kTilesX; ++t) { a1[t] = bias_tile(sbias, lane); xB[t] = parts_from_lds<P>(sxb + t * kStageBytes + lane * 16); }
            stage(a1, xB, nojob);
#pragma unroll
            for (int t = 0; t < kTilesX; ++t) { SplitJob<S, 0, TILES> j(a1[t], hB0[t], wi1); j.all(); }
        }
        for (int c1 = 0; c1 < NC1; ++c1) {
            // k-step 2 c1 of every chunk of mine; meanwhile the other half of a1 becomes hB1 (tile i in stage i)
#pragma unroll
            for (int i = 0; i < kMaxChunks; ++i) {
                if (i < nmine) {                                   // wave-uniform
                    if (i < kTilesX) { SplitJobInStage<S, 1, TILES> j(a1[i], hB1[i], wi1); stage(acc2[i], hB0, j); }
                    else stage(acc2[i], hB0, nojob);
                }
            }
#pragma unroll
            for (int t = 0; t < kTilesX; ++t)                     // tiles no stage of mine has dealt with (fewer chunks than tiles)
                if (t >= nmine) { SplitJob<S, 1, TILES> j(a1[t], hB1[t], wi1); j.all(); }
            if (c1 + 1 < NC1) {                                    // layer 1 of the next chunk
                Parts<P> xB[kTilesX];
                a1[0] = bias_tile(sbias + (c1 + 1) * 32, lane);
#pragma unroll
                for (int t = 0; t < kTilesX; ++t) { a1[t] = a1[0]; xB[t] = parts_from_lds<P>(sxb + t * kStageBytes + lane * 16); }
                stage(a1, xB, nojob);
            }
            // k-step 2 c1 + 1; meanwhile the first half of the next chunk becomes hB0 (stale and unused after the last chunk)
#pragma unroll
            for (int i = 0; i < kMaxChunks; ++i) {
                if (i < nmine) {
                    if (i < kTilesX) { SplitJobInStage<S, 0, TILES> j(a1[i], hB0[i], wi1); stage(acc2[i], hB1, j); }
                    else stage(acc2[i], hB1, nojob);
                }
            }
#pragma unroll
            for (int t = 0; t < kTilesX; ++t)
                if (t >= nmine) { SplitJob<S, 0, TILES> j(a1[t], hB0[t], wi1); j.all(); }
        }

        // ---- layer 3 from the finished layer-2 accumulators, same stream
#pragma unroll
        for (int i = 0; i < kMaxChunks; ++i) {
            if (i < nmine) {
                Parts<P> pB[kTilesX];
#pragma unroll
                for (int t = 0; t < kTilesX; ++t) { SplitJob<S, 0, TILES> j(acc2[i][t], pB[t], wi2); j.all(); }
                stage(y, pB, nojob);
#pragma unroll
                for (int t = 0; t < kTilesX; ++t) { SplitJob<S, 1, TILES> j(acc2[i][t], pB[t], wi2); j.all(); }
                stage(y, pB, nojob);
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the partial sums reuse the rings: no DMA may land late
    if (kTrace && a.trace && lane == 0) {                          // ... and when this wave's stream is done
        a.trace[((size_t)blockIdx.x * 4 + wave) * 8 + 2] = __builtin_amdgcn_s_memtime();
        a.trace[((size_t)blockIdx.x * 4 + wave) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kTilesX; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            spart[((size_t)wave * kRowsX + t * 32 + (lane & 31)) * 33 + cd_row(r, lane)] = y[t][r];
    __syncthreads();

#pragma unroll
    for (int r = 0; r < kRowsX / 64; ++r) {                        // activation + sampling: four lanes per env row
        const int row = 64 * r + (tid >> 2), part = tid & 3;
        const int e = e0 + row;
        if (e >= a.E) return;
        float yv[kQ];
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = part + 4 * i;
            float v = 0.0f;
            if (j < a.fin.nout) {
                if (S::kScaled) {                                  // the waves' partials carry layer 3's weight factor
                    float pv = 0.0f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) pv += spart[((size_t)w * kRowsX + row) * 33 + j];
                    v = fmaf(pv, wi3, sbias[nb + j]);
                } else {
                    v = sbias[nb + j];
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += spart[((size_t)w * kRowsX + row) * 33 + j];
                }
            }
            yv[i] = v;
        }
        finish_quad(a.fin, yv, e, agent, part, tval[r], epval[r]);
    }
}
#undef X3_PIN
#undef X3_RING_READ

// ---------------------------------------------------------------------------------------------------------
// Exact float32, ROW-TILE ownership (round 6; DroneMlp.w2_layout = 2): the register-fused formulation of the split kernels on
// v_mfma_f32_32x32x2_f32, with the work split the other way round.  A wave owns 32 env rows of one agent and ALL output chunks
// of layer 2 for them (up to kRtChunks = 7 accumulator tiles = 112 registers per pass; h2 > 224 takes two passes and recomputes
// layer 1 in the second: 52 of 2900 matrix instructions at h = 400), so
//   * nothing is computed twice across waves and every wave issues the same number of matrix instructions (whole chunks dealt
//     to waves leave 13 / 10 / 7 chunks at 4-3-3-3 / 3-3-2-2 / 2-2-2-1: 0.70 / 0.70 / 0.62 of the matrix peak at best);
//   * layers 1 -> 2 -> 3 meet in REGISTERS: D^T[feature][row] of v_mfma_f32_32x32x2_f32 puts feature 8 (r >> 2) + 4 (lane >> 5)
//     + (r & 3) of row (lane & 31) into register r, and register r of the two lane halves IS the B operand of k-step r of the
//     next layer once the weights' k order is permuted to match on the host (policies.py: pack_f32_rowtile_stream) -- no
//     activation touches LDS, no barrier after the prologue, layer 3 is complete inside the wave;
//   * the vector ALU -- which the float32 matrix instructions run on -- sees 16 v_max per 32-feature chunk and nothing else in the
//     loop: weights travel global -> LDS by DMA into a ring PRIVATE to the wave (4 blocks of 4 KiB = the sixteen A operands of
//     one (in-chunk, out-chunk) pair, four blocks ahead) and LDS -> registers as one ds_read_b128 per four matrix instructions.
// Matrix instructions per wave and 32 rows: 840 / 1740 / 2912 at h = 200 / 300 / 400 = 0.77 / 0.87 / 0.88 of the peak if none stalls.
// One agent's stream, in consumption order (blocks of 4 pieces of 1 KiB = [64 lanes][4 floats]):
//   per pass p (output chunks S_p):  for c1: L1(c1), L2(c1, c2) for c2 in S_p;  then L3(c2) for c2 in S_p;  kRtPad zero blocks.
//   L1(c1): piece 0 = W1[2 r + half][32 c1 + i], r = 0..3; piece 1 = the same for r = 4..6, then b1[32 c1 + i] (lanes < 32);
//   L2(c1, c2): piece q = W2[32 c1 + 8 q + 4 half + j][32 c2 + i], j = 0..3;   L3(c2): piece q = W3[32 c2 + 8 q + 4 half + j][i]
//   (lane = 32 half + i; zero beyond d_in / h1 / h2 / nout).
constexpr int kRtChunks = 7, kRtRing = 4, kRtPad = kRtRing, kRtRows = 128;
__host__ __device__ constexpr int rt_passes(int nc2) { return (nc2 + kRtChunks - 1) / kRtChunks; }
__host__ __device__ constexpr int rt_per_pass(int nc2) { return (nc2 + rt_passes(nc2) - 1) / rt_passes(nc2); }

struct MArgsR {
    int E, N, d_in, h1, h2, nout, nc1, nc2, blocks;
    const float *x, *ws, *b2, *b3;
    const float *w3;                     // VL3: the plain [N][h2][nout] output layer (nout <= 4)
    FinishArgs fin;
    unsigned rb_magic;
};

#define RT_PIN() __builtin_amdgcn_sched_barrier(0)

// VL3 (nout <= 4: the Gaussian actor's 4 moments, the critic's value): layer 3 on the VECTOR ALU from the same registers -- a lane
// multiplies its 16 features of a chunk with their nout weights (LDS table, one ds_read_b128 per feature) and the two lane halves'
// partial sums meet through one permute per output: 64 fused multiply-adds per chunk instead of 16 matrix instructions of 64 cycles
// whose 32 output rows hold nout <= 4 values (7 % of the kernel's matrix time at h = 400, 13 % at h = 200).  The stream then holds
// no L3 blocks.
template <bool VL3>
__global__ void __launch_bounds__(256, 2) mlp3_rt_kernel(const float *x, int E, int N, int d_in, const MArgsR rest)
{
    MArgsR a = rest;
    a.x = x; a.E = E; a.N = N; a.d_in = d_in;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int agent, row_block;
    xcd_work_item((a.E + kRtRows - 1) / kRtRows, agent, row_block, a.rb_magic);
    const int e0 = row_block * kRtRows + 32 * wave;                // this wave's 32 env rows
    const int half = lane >> 5;
    const int nc1 = a.nc1, nc2 = a.nc2, ks1 = (a.d_in + 1) >> 1;
    float *sb2 = reinterpret_cast<float *>(smem);                  // b2, zero padded to whole chunks
    f32x4 *sw3 = reinterpret_cast<f32x4 *>(smem + nc2 * 32 * 4);   // VL3: W3[f][0..3] (zero beyond h2 / nout)
    char *ring = smem + nc2 * 32 * 4 * (VL3 ? 5 : 1) + wave * (kRtRing * 4096);

    // ---- the weight stream.  Everything about it is scalar except the lane's 16-byte slot: the float32 matrix instructions run
    // on the vector ALUs, so every vector instruction in the loop is matrix time lost -- the DMA requests are issued by name with a
    // scalar base and a constant 32-bit lane offset (hipcc forms 64-bit per-lane addresses with two v_lshl_add_u64 per request), and
    // the ring reads by name with counted waits (hipcc waits for ALL outstanding LDS reads in front of a block's first instruction).
    const unsigned long long sbase0 = reinterpret_cast<unsigned long long>(a.ws) + (unsigned long long)agent * a.blocks * 4096ull;
    const unsigned voff = (unsigned)lane * 16u;                    // this lane's slot of a 1-KiB piece (global and LDS alike)
    const unsigned ring_a = lds_addr(ring);
    const unsigned rd_a = ring_a + voff;                           // LDS address of this lane's slot in ring block 0, piece 0
    int cur = 0;                                                   // block being consumed; its ring slot = cur & 3
    auto dma = [&](int blk, int q) {                               // piece q of stream block blk -> its ring slot
        const unsigned long long src = sbase0 + (unsigned long long)(unsigned)blk * 4096ull + (unsigned)q * 1024u;
        const unsigned dst = ring_a + (unsigned)(blk & (kRtRing - 1)) * 4096u + (unsigned)q * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(src) : "memory", "m0");
    };
    f32x4 w[4];                                                    // the current block's sixteen A operands
    // the ring is primed FIRST, ahead of the prologue's own loads: the first four blocks travel while b2 / x / W3 are fetched and
    // the workgroup meets at its barrier (one global round trip less at the head of every wave: 5-10 % of a wave's life at h = 200)
    if (e0 < a.E) {
#pragma unroll 1
        for (int b = 0; b < kRtRing; ++b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma(b, q);
        }
    }
    for (int i = tid; i < nc2 * 32; i += 256) {                    // (clamped address, masked value: no branch around the load)
        const float v = a.b2[(size_t)agent * a.h2 + min(i, a.h2 - 1)];
        sb2[i] = __uint_as_float(__float_as_uint(v) & (i < a.h2 ? 0xffffffffu : 0u));
        if (VL3) {
            f32x4 wv;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float t = a.w3[((size_t)agent * a.h2 + min(i, a.h2 - 1)) * a.nout + min(o, a.nout - 1)];
                wv[o] = __uint_as_float(__float_as_uint(t) & ((i < a.h2 && o < a.nout) ? 0xffffffffu : 0u));
            }
            sw3[i] = wv;
        }
    }
    // the x operand of layer 1: k-step r = inputs 2 r + half of row (lane & 31)
    float xb[7];
    {
        const int e = min(e0 + (lane & 31), a.E - 1);
        const float *xr = a.x + ((size_t)e * a.N + agent) * a.d_in;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int k = 2 * r + half;
            const float v = xr[min(k, a.d_in - 1)];
            xb[r] = __uint_as_float(__float_as_uint(v) & (k < a.d_in ? 0xffffffffu : 0u));
        }
    }
    // the output stage's inputs: four lanes per env row, two rounds of 16 rows
    uint32_t tval[2] = {0u, 0u}, epval[2] = {0u, 0u};
    float b3v[2][kQ];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = e0 + 16 * it + (lane >> 2);
        if (e < a.E && a.fin.sample_kind != 0) {
            if (a.fin.t_dev) tval[it] = (uint32_t)a.fin.t_dev[e];
            if (a.fin.episode_dev) epval[it] = (uint32_t)a.fin.episode_dev[e];
        }
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = (lane & 3) + 4 * i;
            const float v = a.b3[(size_t)agent * a.nout + min(j, a.nout - 1)];
            b3v[it][i] = j < a.nout ? v : 0.0f;
        }
    }
    __syncthreads();                                               // b2 is in LDS; the only barrier of the kernel
    if (e0 >= a.E) return;                                         // a wave without rows (ragged last workgroup)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every load above AND the ring's first four blocks have landed:
                                                                   // from here vmcnt counts the DMA pieces requested in the loop only

    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                 "ds_read_b128 %3, %4 offset:3072" : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(rd_a) : "memory");
    // one group = the four matrix instructions of piece q (`mf`) -- the piece was requested from LDS a block ago: at most three
    // younger reads may still be out -- then the piece's ring slot is refilled with block cur + 4, and piece q of block cur + 1
    // (landed: twelve DMA requests were issued behind it) takes its place in the registers
    auto group = [&](auto Q, auto &&mf) {
        constexpr int q = decltype(Q)::value;
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w[q]) :: "memory");
        mf(w[q]);
        RT_PIN();
        dma(cur + kRtRing, q);
        const unsigned ra = rd_a + (unsigned)((cur + 1) & (kRtRing - 1)) * 4096u;
        asm volatile("s_waitcnt vmcnt(12)\n\tds_read_b128 %0, %1 offset:%2" : "=v"(w[q]) : "v"(ra), "n"(q * 1024) : "memory");
        RT_PIN();
    };
    typedef std::integral_constant<int, 0> Q0; typedef std::integral_constant<int, 1> Q1;
    typedef std::integral_constant<int, 2> Q2; typedef std::integral_constant<int, 3> Q3;
    auto nothing = [](const f32x4 &) {};

    f32x16 y = {0}, y1 = {0};
    const float one = lane < 32 ? 1.0f : 0.0f;
    const int passes = rt_passes(nc2), per = rt_per_pass(nc2);
    for (int p = 0; p < passes; ++p) {
        const int c2_0 = p * per, npc = min(per, nc2 - c2_0);      // this pass's output chunks (wave-uniform)
        f32x16 acc2[kRtChunks];
#pragma unroll
        for (int i = 0; i < kRtChunks; ++i) acc2[i] = bias_tile(sb2 + min(c2_0 + i, nc2 - 1) * 32, lane);
        for (int c1 = 0; c1 < nc1; ++c1) {
            // layer 1 of chunk c1: K = d_in <= 14, then the bias on one more matrix instruction (A = b1 in lanes 0..31, B = 1 there)
            f32x16 a1 = {0};
            group(Q0{}, [&](const f32x4 &v) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < ks1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], xb[j], a1, 0, 0, 0);
            });
            group(Q1{}, [&](const f32x4 &v) {
#pragma unroll
                for (int j = 0; j < 3; ++j) if (4 + j < ks1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], xb[4 + j], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3], one, a1, 0, 0, 0);
            });
            group(Q2{}, nothing);
            group(Q3{}, nothing);
            ++cur;
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[r] = fmaxf(a1[r], 0.0f);
            // layer 2: every output chunk of the pass takes this chunk's 32 features -- 16 k-steps, B = register r of a1.  Group q holds
            // the features 8 q .. 8 q + 7 of the chunk: the groups of a ragged last chunk that hold none are skipped (h = 400 / 300 /
            // 200 end in chunks of 16 / 12 / 8 features: 8 / 8 / 12 of the 16 instructions of each of that chunk's blocks)
            const int kv = a.h1 - 32 * c1;                     // features of this in-chunk (wave-uniform)
#pragma unroll
            for (int i = 0; i < kRtChunks; ++i) {
                if (i < npc) {
                    group(Q0{}, [&](const f32x4 &v) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], a1[j], acc2[i], 0, 0, 0);
                    });
                    group(Q1{}, [&](const f32x4 &v) {
                        if (kv > 8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], a1[4 + j], acc2[i], 0, 0, 0);
                        }
                    });
                    group(Q2{}, [&](const f32x4 &v) {
                        if (kv > 16) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], a1[8 + j], acc2[i], 0, 0, 0);
                        }
                    });
                    group(Q3{}, [&](const f32x4 &v) {
                        if (kv > 24) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], a1[12 + j], acc2[i], 0, 0, 0);
                        }
                    });
                    ++cur;
                }
            }
        }
        // layer 3 from the finished chunks of this pass: two accumulation chains (even / odd chunks) that are added at the end --
        // half the roundings in a row on the outputs' own scale (every instruction rounds once; the 16-column instruction of the
        // LDS-staged kernel takes four products per rounding, this one two)
        auto layer3 = [&](f32x16 &yy, f32x16 &h, int kv3, int c2) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = fmaxf(h[r], 0.0f);
            if constexpr (VL3) {                             // registers 0..3 of yy = this lane's partial sums of the nout <= 4 outputs
                const f32x4 *wp = sw3 + 32 * c2 + 4 * half;  // this lane's features: 8 q + 4 half + j
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 wv = wp[8 * q + j];
#pragma unroll
                        for (int o = 0; o < 4; ++o) yy[o] = fmaf(wv[o], h[4 * q + j], yy[o]);
                    }
                }
                return;
            }
            group(Q0{}, [&](const f32x4 &v) {
#pragma unroll
                for (int j = 0; j < 4; ++j) yy = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], h[j], yy, 0, 0, 0);
            });
            group(Q1{}, [&](const f32x4 &v) {
                if (kv3 > 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) yy = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], h[4 + j], yy, 0, 0, 0);
                }
            });
            group(Q2{}, [&](const f32x4 &v) {
                if (kv3 > 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) yy = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], h[8 + j], yy, 0, 0, 0);
                }
            });
            group(Q3{}, [&](const f32x4 &v) {
                if (kv3 > 24) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) yy = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], h[12 + j], yy, 0, 0, 0);
                }
            });
            ++cur;
        };
#pragma unroll
        for (int i = 0; i < kRtChunks; ++i) {
            if (i < npc) {
                const int kv3 = a.h2 - 32 * (c2_0 + i);      // features of this chunk (the same skipping as in layer 2)
                if (i & 1) layer3(y1, acc2[i], kv3, c2_0 + i); else layer3(y, acc2[i], kv3, c2_0 + i);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the output tile reuses the ring: no DMA may land late

    // ---- output activation + sampling: the wave's 32 x nout tile through its own LDS region, four lanes per env row
    float *st = reinterpret_cast<float *>(ring);                   // [32 rows][33]
    if constexpr (VL3) {                                           // the two lane halves of a row hold the two halves of its features
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float p = y[o] + y1[o];
            const float other = __shfl_xor(p, 32, 64);
            if (half == 0) st[(lane & 31) * 33 + o] = p + other;
        }
    } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[(lane & 31) * 33 + cd_row(r, lane)] = y[r] + y1[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = 16 * it + (lane >> 2), part = lane & 3;
        const int e = e0 + row;
        if (e < a.E) {
            float yv[kQ];
#pragma unroll
            for (int i = 0; i < kQ; ++i) {
                const int j = part + 4 * i;
                yv[i] = j < a.nout ? st[row * 33 + j] + b3v[it][i] : 0.0f;
            }
            finish_quad(a.fin, yv, e, agent, part, tval[it], epval[it]);
        }
    }
}
#undef RT_PIN

// ---------------------------------------------------------------------------------------------------------
// Two-part float16 split (f16x2), ROW-TILE ownership with ONE weight ring per workgroup (round 6; dronesim_mlp_forward_f16x2_rt).
// The formulation of mlp3_rt_kernel on v_mfma_f32_32x32x16_f16: a wave owns 32 env rows of one agent and every output chunk of layer 2
// for them (kRtChunks accumulator tiles per pass, layer 1 recomputed in the second pass), layers meet in registers through the
// float16 split of the accumulator tile, layer 3 runs on the vector ALU in exact float32 (VL3: nout <= 4) or takes the split, relu'd
// layer-2 tiles as the B operand of its own blocks.  Against mlp3_split_kernel
// (wave w owns output chunks w, w + 4, ... of two row tiles): no layer-1 work and no relu + split repeated by four waves, no
// 4-3-3-3 dealing of 13 chunks -- 1053 instead of 1500 matrix instructions per 32 rows at h = 400 -- and HALF the weight bytes per
// matrix instruction, because the four waves of a workgroup consume the SAME stream: it travels global -> LDS once per workgroup,
// into a ring of kR16Depth = 4 super-stages of four 4-KiB blocks; wave w requests piece w of every block.  One s_barrier per
// super-stage (24 matrix instructions) orders it: a wave that is about to read the first block of super-stage S has waited for its own
// pieces of S (counted vmcnt), so behind the barrier S is complete in LDS; and every wave has consumed all of S - 2 (its reads of
// S - 1's last block may still be in flight: nobody drains its LDS queue for the barrier), so the slot of S - 2 takes super-stage
// S + 2.  No other barrier after the prologue.
// One agent's stream (blocks of four 1-KiB pieces [64 lanes][8 float16]; policies.py: pack_f16_rowtile_stream):
//   per pass p (output chunks S_p):  for c1:  L1(c1) = (W1 hi, W1 lo, 0, 0) of chunk c1 (one 16-wide k-step, linear k order),
//                                             L2(c1, c2) = (hi, lo of k-step 2 c1), (hi, lo of k-step 2 c1 + 1) for c2 in S_p
//                                             (accumulator k order: 16 s + 8 (j >> 2) + 4 half + (j & 3));
//                                    nout > 4 (layer 3 on the matrix cores), after the pass's in-chunks:  L3(c2) = (hi, lo of k-step 2 c2),
//                                             (hi, lo of k-step 2 c2 + 1) of W3^T, outputs zero-padded to 32, for c2 in S_p;
//   padded to whole super-stages, then three empty super-stages (the requests run two super-stages ahead, the reads one block).  Weights carry DroneMlpBf16.wscale like the split kernel's.
constexpr int kR16Depth = 4;

struct MArgsR16 {
    int E, N, d_in, h1, h2, nout, nc1, nc2, blocks;
    const float *x, *b1, *b2, *b3, *w3, *wscale;
    const char *ws;
    FinishArgs fin;
    unsigned rb_magic;
    long long *trace;                                             // developer trace builds only (tools/trace_rt16.py)
};

template <bool VL3>
__global__ void __launch_bounds__(256, 2) mlp3_rt16_kernel(const float *x, int E, int N, int d_in, const MArgsR16 rest)
{
    typedef SchemeF16x2 S;
    constexpr int P = 2;
    MArgsR16 a = rest;
    a.x = x; a.E = E; a.N = N; a.d_in = d_in;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int agent, row_block;
    xcd_work_item((a.E + kRtRows - 1) / kRtRows, agent, row_block, a.rb_magic);
    const int e0 = row_block * kRtRows + 32 * wave;                // this wave's 32 env rows (a wave without rows runs on clamped ones:
    const int half = lane >> 5;                                    // every wave takes part in every barrier)
    const int nc1 = a.nc1, nc2 = a.nc2;
    PT64(0);
    if (kTrace && a.trace && lane == 0) a.trace[((size_t)blockIdx.x * 4 + wave) * 64 + 32] = __builtin_amdgcn_s_memrealtime();
    if (kTrace && a.trace && lane == 0) {                          // where the workgroup runs: HW_ID (wave, simd, cu, sh, se ...) and the XCC
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        a.trace[((size_t)blockIdx.x * 4 + wave) * 64 + 34] = (long long)hw | ((long long)xcc << 32);
    }
    float ws1 = 1.0f, ws2 = 1.0f, wi1 = 1.0f, wi2 = 1.0f, wi3 = 1.0f;
    if (a.wscale != nullptr) {
        ws1 = a.wscale[3 * (size_t)agent]; ws2 = a.wscale[3 * (size_t)agent + 1];
        wi1 = __builtin_amdgcn_rcpf(ws1); wi2 = __builtin_amdgcn_rcpf(ws2);
        if constexpr (!VL3) wi3 = __builtin_amdgcn_rcpf(a.wscale[3 * (size_t)agent + 2]);
    }
    float *sb1 = reinterpret_cast<float *>(smem);                  // b1 * ws1 | b2 * ws2, zero padded to whole chunks
    float *sb2 = sb1 + nc1 * 32;
    f32x4 *sw3 = reinterpret_cast<f32x4 *>(sb2 + nc2 * 32);        // VL3: W3[f][0..3] / layer 2's weight factor (zero beyond h2 / nout)
    char *ring = reinterpret_cast<char *>(sw3 + (VL3 ? nc2 * 32 : 0));   // [kR16Depth][4 blocks][4 pieces][1 KiB], shared by the workgroup

    // ---- the weight stream: wave w requests piece w of every block (scalar base + this lane's 16-byte slot, by name)
    const unsigned long long sbase0 = reinterpret_cast<unsigned long long>(a.ws) + (unsigned long long)agent * a.blocks * 4096ull +
                                      (unsigned long long)wave * 1024ull;
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned ring_a = lds_addr(ring);
    const unsigned rd_a = ring_a + voff;
    auto dma_super = [&](int sst) {                                // this wave's four pieces of super-stage sst
        const unsigned slot = ring_a + (unsigned)(sst & (kR16Depth - 1)) * 16384u + (unsigned)wave * 1024u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned long long src = sbase0 + (unsigned long long)(unsigned)(4 * sst + b) * 4096ull;
            const unsigned dst = slot + (unsigned)b * 4096u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(src) : "memory", "m0");
        }
    };
    dma_super(0);                                                  // primed ahead of the prologue's own loads

    // Prologue loads, all requested before anything waits: biases and W3 (at most four table entries per thread: h1, h2 <= 512; clamped
    // addresses, masked values: no branch around the loads), the x rows, the sampling counters, b3 -- then super-stages 1 and 2, so
    // that ONE counted wait (all but those eight requests) covers what the prologue needs and the first super-stage.
    const int total = (nc1 + nc2) * 32;
    float bv[4];
    f32x4 wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = min(tid + 256 * u, total - 1);
        const bool l1 = i < nc1 * 32;
        const int j = l1 ? i : i - nc1 * 32, h = l1 ? a.h1 : a.h2;
        bv[u] = (l1 ? a.b1 : a.b2)[(size_t)agent * h + min(j, h - 1)];
        if constexpr (VL3) {
            const size_t row = (size_t)agent * a.h2 + (l1 ? 0 : min(j, a.h2 - 1));
#pragma unroll
            for (int o = 0; o < 4; ++o) wv[u][o] = a.w3[row * a.nout + min(o, a.nout - 1)];
        }
    }
    float xv[8];
    {
        const int e = min(e0 + (lane & 31), a.E - 1);
        const float *xr = a.x + ((size_t)e * a.N + agent) * a.d_in;
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = xr[min(8 * half + j, a.d_in - 1)];
    }
    uint32_t tval[2] = {0u, 0u}, epval[2] = {0u, 0u};
    float b3v[2][kQ];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = e0 + 16 * it + (lane >> 2);
        if (e < a.E && a.fin.sample_kind != 0) {
            if (a.fin.t_dev) tval[it] = (uint32_t)a.fin.t_dev[e];
            if (a.fin.episode_dev) epval[it] = (uint32_t)a.fin.episode_dev[e];
        }
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
            const int j = (lane & 3) + 4 * i;
            const float v = a.b3[(size_t)agent * a.nout + min(j, a.nout - 1)];
            b3v[it][i] = j < a.nout ? v : 0.0f;
        }
    }
    dma_super(1);
    dma_super(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // the tables -> LDS (the W3 rows carry 1 / (layer 2's weight factor), a power of two)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + 256 * u;
        if (i < total) {
            const bool l1 = i < nc1 * 32;
            const int j = l1 ? i : i - nc1 * 32, h = l1 ? a.h1 : a.h2;
            sb1[i] = __uint_as_float(__float_as_uint(bv[u]) & (j < h ? 0xffffffffu : 0u)) * (l1 ? ws1 : ws2);
            if constexpr (VL3) {
                if (!l1) {
                    f32x4 w;
#pragma unroll
                    for (int o = 0; o < 4; ++o) w[o] = __uint_as_float(__float_as_uint(wv[u][o]) & ((j < a.h2 && o < a.nout) ? 0xffffffffu : 0u)) * wi2;
                    sw3[j] = w;
                }
            }
        }
    }
    // the x operand of layer 1 (one 16-wide k-step, linear order: inputs 8 half + j of row lane & 31), split once
    Parts<P> xB;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) v2[t] = __uint_as_float(__float_as_uint(xv[2 * q + t]) & (8 * half + 2 * q + t < a.d_in ? 0xffffffffu : 0u));
        unsigned d[P];
        S::template split_pair<false>(v2[0], v2[1], d);
#pragma unroll
        for (int p = 0; p < P; ++p) xB.p[p][q] = d[p];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this wave's table entries are in LDS (its pieces of super-stage 0 as well)
    __builtin_amdgcn_s_barrier();                                  // ... and everybody else's
    PT64(1);

    // ---- consumption.  wf[0..3] = the current block's pieces (k-step 0 hi, lo; k-step 1 hi, lo); a piece is re-read with the NEXT
    // block's bytes right behind its last product.  `cur` = index of the block whose pieces are being (re)loaded.
    u32x4 wf[4];
    int cur = 0;
    unsigned ra = rd_a;                                            // this lane's address of block `cur` in the ring
    // (requested in the order the pieces are consumed -- lo, hi of k-step 0, then lo, hi of k-step 1 -- so that the piece about to
    // be used is always the OLDEST of at most four reads in flight: every wait below is lgkmcnt(3))
    asm volatile("ds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %0, %4\n\tds_read_b128 %3, %4 offset:3072\n\tds_read_b128 %2, %4 offset:2048"
                 : "=&v"(wf[0]), "=&v"(wf[1]), "=&v"(wf[2]), "=&v"(wf[3]) : "v"(rd_a) : "memory");
    // the first read of a block: when it opens super-stage sst >= 1, the workgroup meets first (see the header)
    auto open_block = [&]() {
        ++cur;
        ra = rd_a + ((unsigned)cur & (4u * kR16Depth - 1u)) * 4096u;
        if ((cur & 3) == 0) {
            const int sst = cur >> 2;
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // this wave's pieces of sst have landed (sst + 1's four may be out)
            __builtin_amdgcn_s_barrier();
            dma_super(sst + 2);                                    // the slot of sst - 2: every wave is past its last block
        }
    };
    // (immediate offsets per piece: four variants by name)
    auto ring_read_p = [&](u32x4 &dst, auto PIECE) {
        constexpr int piece = decltype(PIECE)::value;
        const unsigned at = ra;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(at), "n"(piece * 1024) : "memory");
    };
    typedef std::integral_constant<int, 0> P0; typedef std::integral_constant<int, 1> P1;
    typedef std::integral_constant<int, 2> P2; typedef std::integral_constant<int, 3> P3;
    // one k-step of a block: the scheme's three products (lo.hi, hi.lo, hi.hi) of pieces (hiP, loP) with the B parts `b`; each piece is
    // reloaded with the next block's bytes behind its last product; FIRST: this k-step opens the next block (k-step 0)
    // `live` (wave-uniform): false = the k-step holds no feature (ragged last chunk): its pieces only make way for the next block's.
    // NOTE the reads are UNCONDITIONAL and only the matrix instructions sit under the branch: a register that an asynchronous read
    // issued by name is still filling must never meet a control-flow join -- hipcc resolves the join with register copies and does
    // not know that the value is not there yet (the first cut of this kernel copied stale pieces that way).
    auto kstep = [&](f32x16 &acc, const Parts<P> &b, auto HI, auto LO, auto FIRST, bool live) {
        constexpr int hi = decltype(HI)::value, lo = decltype(LO)::value;
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[lo]) :: "memory");
        if (live) acc = S::mfma(wf[lo], b.p[0], acc);              // lo . hi
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(FIRST)::value) open_block();
        ring_read_p(wf[lo], LO);
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[hi]) :: "memory");
        if (live) {
            acc = S::mfma(wf[hi], b.p[1], acc);                    // hi . lo
            acc = S::mfma(wf[hi], b.p[0], acc);                    // hi . hi
        }
        __builtin_amdgcn_sched_barrier(0);
        ring_read_p(wf[hi], HI);
    };

    // layer 3 (vector ALU): this lane's partial sums of the outputs as two packed pairs (v_pk_fma_f32: two outputs per instruction),
    // two chains (even / odd chunks)
    f32x2 ya[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, yb[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    // nout > 4: layer 3 on the matrix cores, out^T[output][row] in two accumulator tiles (even / odd chunks)
    f32x16 y3a, y3b;
#pragma unroll
    for (int r = 0; r < 16; ++r) { y3a[r] = 0.0f; y3b[r] = 0.0f; }
    const int passes = rt_passes(nc2), per = rt_per_pass(nc2);
    for (int p = 0; p < passes; ++p) {
        const int c2_0 = p * per, npc = min(per, nc2 - c2_0);
        f32x16 acc2[kRtChunks];
#pragma unroll
        for (int i = 0; i < kRtChunks; ++i) acc2[i] = bias_tile(sb2 + min(c2_0 + i, nc2 - 1) * 32, lane);
        // in-chunk c1: layer 1 (block L1: pieces 0, 1 = W1 hi, lo; 2, 3 unused), relu + split -> the two k-steps' B operands, then its
        // two k-steps of every output chunk of the pass.  FULL: the chunk holds 32 features (every chunk but a ragged last one, whose
        // second k-step is empty when it has <= 16: the test is compiled only into the last chunk's copy of the code)
        auto in_chunk = [&](int c1, auto FULL) {
            f32x16 a1 = bias_tile(sb1 + c1 * 32, lane);
            kstep(a1, xB, P0{}, P1{}, std::true_type{}, true);
            kstep(a1, xB, P2{}, P3{}, std::false_type{}, false);
            Parts<P> hB0, hB1;
            { SplitJob<S, 0, 1> j(a1, hB0, wi1); j.all(); }
            { SplitJob<S, 1, 1> j(a1, hB1, wi1); j.all(); }
            const bool second = decltype(FULL)::value ? true : a.h1 - 32 * c1 > 16;
#pragma unroll
            for (int i = 0; i < kRtChunks; ++i) {
                if (i < npc) {
                    kstep(acc2[i], hB0, P0{}, P1{}, std::true_type{}, true);
                    kstep(acc2[i], hB1, P2{}, P3{}, std::false_type{}, second);
                }
            }
        };
        for (int c1 = 0; c1 < nc1 - 1; ++c1) { in_chunk(c1, std::true_type{}); PT64(2 + 14 * min(p, 1) + min(c1, 12)); }
        in_chunk(nc1 - 1, std::false_type{});
        PT64(2 + 14 * min(p, 1) + min(nc1 - 1, 12));
        // layer 3 on the vector ALU, exact float32: relu (the weight factor of layer 2 is undone by the table), 16 features x nout <= 4
        // per chunk and lane: per feature pair two v_max and four packed multiply-adds (the pair's relu'd values are the low / high
        // half of one 64-bit operand)
        if constexpr (!VL3) {
            // the pass's blocks L3(c2): the relu'd, split chunk is the B operand of its two k-steps, W3^T (outputs zero-padded to 32) the A
#pragma unroll
            for (int i = 0; i < kRtChunks; ++i) {
                if (i < npc) {
                    Parts<P> gB0, gB1;
                    { SplitJob<S, 0, 1> j(acc2[i], gB0, wi2); j.all(); }
                    { SplitJob<S, 1, 1> j(acc2[i], gB1, wi2); j.all(); }
                    const bool second = a.h2 - 32 * (c2_0 + i) > 16;
                    if (i & 1) {
                        kstep(y3b, gB0, P0{}, P1{}, std::true_type{}, true);
                        kstep(y3b, gB1, P2{}, P3{}, std::false_type{}, second);
                    } else {
                        kstep(y3a, gB0, P0{}, P1{}, std::true_type{}, true);
                        kstep(y3a, gB1, P2{}, P3{}, std::false_type{}, second);
                    }
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < kRtChunks; ++i) {
            if (i < npc) {
                const f32x4 *wp = sw3 + 32 * (c2_0 + i) + 4 * half;
                f32x2 (&y)[2] = (i & 1) ? yb : ya;
                f32x4 w3r[16];                                    // the chunk's 16 table rows of this lane, requested together
#pragma unroll
                for (int k = 0; k < 16; ++k) w3r[k] = wp[8 * (k >> 2) + (k & 3)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        f32x2 hv;
                        asm("v_max_f32 %0, 0, %1" : "=v"(hv.x) : "v"(acc2[i][4 * q + j]));
                        asm("v_max_f32 %0, 0, %1" : "=v"(hv.y) : "v"(acc2[i][4 * q + j + 1]));
                        const f32x4 w0 = w3r[4 * q + j], w1 = w3r[4 * q + j + 1];
                        const f32x2 w0a = {w0[0], w0[1]}, w0b = {w0[2], w0[3]}, w1a = {w1[0], w1[1]}, w1b = {w1[2], w1[3]};
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(y[0]) : "v"(w0a), "v"(hv));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(y[1]) : "v"(w0b), "v"(hv));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(y[0]) : "v"(w1a), "v"(hv));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(y[1]) : "v"(w1b), "v"(hv));
                    }
                }
            }
        }
        }
        PT64(15 + 14 * min(p, 1));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // no DMA may land late: the output tile reuses the ring
    __syncthreads();                                               // ... and no wave may still be reading it
    PT64(30);

    float *st = reinterpret_cast<float *>(ring) + wave * (32 * 33);
    if constexpr (VL3) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float pv = ya[o >> 1][o & 1] + yb[o >> 1][o & 1];
            const float other = __shfl_xor(pv, 32, 64);
            if (half == 0) st[(lane & 31) * 33 + o] = pv + other;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[(lane & 31) * 33 + cd_row(r, lane)] = (y3a[r] + y3b[r]) * wi3;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = 16 * it + (lane >> 2), part = lane & 3;
        const int e = e0 + row;
        if (e < a.E) {
            float yv[kQ];
#pragma unroll
            for (int i = 0; i < kQ; ++i) {
                const int j = part + 4 * i;
                yv[i] = j < a.nout ? st[row * 33 + j] + b3v[it][i] : 0.0f;
            }
            finish_quad(a.fin, yv, e, agent, part, tval[it], epval[it]);
        }
    }
    PT64(31);
    if (kTrace && a.trace && lane == 0) a.trace[((size_t)blockIdx.x * 4 + wave) * 64 + 33] = __builtin_amdgcn_s_memrealtime();   // (100 MHz)
}

// > 64 KiB of dynamic LDS must be opted into once per (kernel, device): a bit mask of device ordinals per kernel,
// guarded by a mutex (the library may be driven from several host threads / devices of one process)
int enable_big_lds(const void *kernel, unsigned long long (&opted)[4], std::mutex &mu, const char *what)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 255) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (opted[dev >> 6] >> (dev & 63) & 1ull) return DRONESIM_OK;
    hipFuncAttributes fa{};                                        // 160 KiB per CU, minus what the kernel holds statically
    hipError_t e = hipFuncGetAttributes(&fa, kernel);
    if (e == hipSuccess) e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
    if (e != hipSuccess) {
        char msg[160];
        snprintf(msg, sizeof(msg), "cannot enable 160 KiB of dynamic LDS for %s: %s", what, hipGetErrorString(e));
        return dronesim_fail(DRONESIM_ELAUNCH, msg);
    }
    opted[dev >> 6] |= 1ull << (dev & 63);
    return DRONESIM_OK;
}

template <bool VL3>
int launch_rt16(const MArgsR16 &r, dim3 grid, size_t lds, hipStream_t stream)
{
    static std::mutex mu;
    static unsigned long long opted[4] = {0ull, 0ull, 0ull, 0ull};
    const int rc = enable_big_lds(reinterpret_cast<const void *>(mlp3_rt16_kernel<VL3>), opted, mu, "mlp3_rt16_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(mlp3_rt16_kernel<VL3>, grid, dim3(256), lds, stream, r.x, r.E, r.N, r.d_in, r);
    return DRONESIM_OK;
}

template <int NC1>
int launch_bf16(const MArgsB &a, size_t lds, hipStream_t stream)
{
    if (lds > 48 * 1024) {
        static std::mutex mu;
        static unsigned long long opted[4] = {0ull, 0ull, 0ull, 0ull};
        const int rc = enable_big_lds(reinterpret_cast<const void *>(mlp3_bf16_kernel<NC1>), opted, mu, "mlp3_bf16_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(mlp3_bf16_kernel<NC1>, dim3(((a.E + kRowsB - 1) / kRowsB) * a.N), dim3(256), lds, stream,
                       a.x, a.E, a.N, a.d_in, a);
    return DRONESIM_OK;
}

FinishArgs make_finish(int N, int nout, int out_kind, int sample_kind, float *out, float *act, int32_t *act_idx,
                       uint64_t seed, uint64_t counter, int64_t env_base, const int32_t *t, const int32_t *episode)
{
    FinishArgs f{};
    f.N = N; f.nout = nout; f.out_kind = out_kind; f.sample_kind = sample_kind;
    f.out = out; f.act = act; f.act_idx = act_idx;
    f.key0 = (uint32_t)seed; f.key1 = (uint32_t)(seed >> 32);
    f.ctr2 = (uint32_t)counter; f.ctr3 = (uint32_t)(counter >> 32);
    f.env_base = env_base; f.t_dev = t; f.episode_dev = episode;
    return f;
}

int check_mlp(const char *who, int N, int d_in, int h1, int h2, int nout, int out_kind, int sample_kind, int E)
{
    (void)who;
    if (N < 1 || d_in < 1 || d_in > 64 || h1 < 1 || h1 > 512 || h2 < 1 || h2 > 512 || nout < 1 || nout > kMaxOut)
        return dronesim_fail(DRONESIM_EUNSUPPORTED, "mlp forward: need d_in<=64, h1,h2<=512, nout<=32");
    if (out_kind < 0 || out_kind > 2 || sample_kind < 0 || sample_kind > 2)
        return dronesim_fail(DRONESIM_EINVAL, "mlp forward: bad out_kind / sample_kind");
    if (sample_kind == 2 && (out_kind != 2 || nout != 4))
        return dronesim_fail(DRONESIM_EINVAL, "Gaussian sampling needs out_kind 2 with nout = 4 (mu_x, mu_y, var_x, var_y)");
    if (E < 0) return dronesim_fail(DRONESIM_EINVAL, "E < 0");
    return DRONESIM_OK;
}

}   // namespace

// developer hook (tools/trace_policy.py / trace_x3.py with a -DDRONESIM_TRACE build; not declared in include/dronesim.h)
extern "C" void dronesim_debug_set_policy_trace(long long *p) { g_policy_trace = p; }

extern "C" int dronesim_mlp_forward_bf16(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                                         uint64_t seed, uint64_t counter, int64_t env_base,
                                         const int32_t *t, const int32_t *episode, int E, void *stream)
{
    if (!m || !x) return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward_bf16: NULL argument");
    const int rc = check_mlp("bf16", m->N, m->d_in, m->h1, m->h2, m->nout, m->out_kind, m->sample_kind, E);
    if (rc) return rc;
    if (m->d_in > 16) return dronesim_fail(DRONESIM_EUNSUPPORTED, "bf16 path: d_in <= 16");
    if (!m->w1p || !m->w2p || !m->w3p || !m->b1 || !m->b2 || !m->b3)
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward_bf16: NULL weight array");
    if (E == 0) return DRONESIM_OK;
    MArgsB a{};
    a.trace = kTrace ? g_policy_trace : nullptr;
    a.E = E; a.N = m->N; a.d_in = m->d_in; a.h1 = m->h1; a.h2 = m->h2;
    a.nc1 = (m->h1 + 31) / 32; a.nc2 = (m->h2 + 31) / 32; a.ks1 = 1;
    a.x = x; a.b1 = m->b1; a.b2 = m->b2; a.b3 = m->b3;
    a.w1p = reinterpret_cast<const bf16x8 *>(m->w1p); a.w2p = reinterpret_cast<const bf16x8 *>(m->w2p);
    a.w3p = reinterpret_cast<const bf16x8 *>(m->w3p);
    a.fin = make_finish(m->N, m->nout, m->out_kind, m->sample_kind, out, act, act_idx, seed, counter, env_base, t, episode);
    const size_t sh1_bytes = sizeof(__bf16) * kRowsB * ((size_t)a.nc1 * 32 + 8);
    const size_t part_bytes = sizeof(float) * 4 * kRowsB * 33;
    const size_t lds = sizeof(__bf16) * kRowsB * kLdx + sizeof(float) * (32 * (a.nc1 + a.nc2) + 32) +
                       (sh1_bytes > part_bytes ? sh1_bytes : part_bytes);
    if (lds > 160 * 1024) return dronesim_fail(DRONESIM_EUNSUPPORTED, "hidden layer too wide for the LDS tile");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int lrc = DRONESIM_OK;
    switch (a.nc1) {                                     // the h1 tile's register residency is compile-time
#define DRONESIM_BF16_CASE(n) case n: lrc = launch_bf16<n>(a, lds, s); break;
        DRONESIM_BF16_CASE(1) DRONESIM_BF16_CASE(2) DRONESIM_BF16_CASE(3) DRONESIM_BF16_CASE(4)
        DRONESIM_BF16_CASE(5) DRONESIM_BF16_CASE(6) DRONESIM_BF16_CASE(7) DRONESIM_BF16_CASE(8)
        DRONESIM_BF16_CASE(9) DRONESIM_BF16_CASE(10) DRONESIM_BF16_CASE(11) DRONESIM_BF16_CASE(12)
        DRONESIM_BF16_CASE(13) DRONESIM_BF16_CASE(14) DRONESIM_BF16_CASE(15) DRONESIM_BF16_CASE(16)
#undef DRONESIM_BF16_CASE
        default: return dronesim_fail(DRONESIM_EUNSUPPORTED, "bf16 path: h1 <= 512");
    }
    if (lrc) return lrc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dronesim_fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

// stages per (agent, wave) stream: wave 0 owns the most chunks; + padding for the run-ahead requests
extern "C" int dronesim_mlp_bf16x3_stages(int h1, int h2)
{
    const int nc1 = (h1 + 31) / 32, nc2 = (h2 + 31) / 32, nm = (nc2 + 3) / 4;
    return nc1 * (1 + 2 * nm) + 2 * nm + kStreamPadX;
}

template <class S, int TILES>
int launch_split(const MArgsX &a, int N, hipStream_t stream)
{
    constexpr size_t stage_bytes = S::kParts * 1024, rows = 32 * TILES;
    const size_t part_bytes = sizeof(float) * 4 * rows * 33, ring_bytes = (size_t)4 * split_ring_depth(TILES) * stage_bytes;   // share LDS
    const size_t lds = sizeof(float) * (32 * (size_t)(a.nc1 + a.nc2) + 32) + TILES * stage_bytes +
                       (part_bytes > ring_bytes ? part_bytes : ring_bytes);
    if (lds > 48 * 1024) {
        static std::mutex mu;
        static unsigned long long opted[4] = {0ull, 0ull, 0ull, 0ull};
        const int rc = enable_big_lds(reinterpret_cast<const void *>(mlp3_split_kernel<S, TILES>), opted, mu, "mlp3_split_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL((mlp3_split_kernel<S, TILES>), dim3(((a.E + rows - 1) / rows) * N), dim3(256), lds, stream,
                       a.x, a.E, a.N, a.d_in, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dronesim_fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

template <class S>
int mlp_forward_split(const char *what, const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                      uint64_t seed, uint64_t counter, int64_t env_base, const int32_t *t, const int32_t *episode, int E,
                      void *stream)
{
    char msg[200];
    if (!m || !x) { snprintf(msg, sizeof(msg), "dronesim_mlp_forward_%s: NULL argument", what); return dronesim_fail(DRONESIM_EINVAL, msg); }
    const int rc = check_mlp(what, m->N, m->d_in, m->h1, m->h2, m->nout, m->out_kind, m->sample_kind, E);
    if (rc) return rc;
    if (m->d_in > 16) { snprintf(msg, sizeof(msg), "%s path: d_in <= 16", what); return dronesim_fail(DRONESIM_EUNSUPPORTED, msg); }
    if (!m->w1p || !m->b1 || !m->b2 || !m->b3) {
        snprintf(msg, sizeof(msg), "dronesim_mlp_forward_%s: NULL weight array", what);
        return dronesim_fail(DRONESIM_EINVAL, msg);
    }
    if (E == 0) return DRONESIM_OK;
    MArgsX a{};
    a.E = E; a.N = m->N; a.d_in = m->d_in; a.h1 = m->h1; a.h2 = m->h2;
    a.nc1 = (m->h1 + 31) / 32; a.nc2 = (m->h2 + 31) / 32;
    a.stages = dronesim_mlp_bf16x3_stages(m->h1, m->h2);
    if (m->reserved != a.stages) {
        snprintf(msg, sizeof(msg), "dronesim_mlp_forward_%s: DroneMlpBf16.reserved must hold dronesim_mlp_bf16x3_stages(h1, h2), "
                                   "the stages per stream of w1p", what);
        return dronesim_fail(DRONESIM_EINVAL, msg);
    }
    a.x = x; a.b1 = m->b1; a.b2 = m->b2; a.b3 = m->b3;
    a.wscale = S::kScaled ? m->wscale : nullptr;
    a.ws = reinterpret_cast<const char *>(m->w1p);
    a.trace = kTrace ? g_policy_trace : nullptr;
    a.fin = make_finish(m->N, m->nout, m->out_kind, m->sample_kind, out, act, act_idx, seed, counter, env_base, t, episode);
    // TILES = 4 (128-row workgroups, one per CU, accumulators in the AGPR half of a 512-register wave) builds and is
    // correct, but hipcc 7.2 places the accumulators badly (1100 v_accvgpr copies, scratch spills whose waits drain the
    // DMA ring): 561 vs 301 us at C3 Gaussian f16x2.  Not instantiated; it needs hand-placed AGPR accumulators.
    return launch_split<S, 2>(a, m->N, static_cast<hipStream_t>(stream));
}

// blocks (4 KiB each) of one agent's row-tile weight stream, zero padding included (DroneMlp.w2_layout = 2; see mlp3_rt_kernel)
// (nout <= 4: layer 3 runs on the vector ALU from the plain w3 array and the stream holds no L3 blocks)
extern "C" int dronesim_mlp_rt_blocks(int h1, int h2, int nout)
{
    if (h1 < 1 || h2 < 1 || nout < 1) return 0;
    const int nc1 = (h1 + 31) / 32, nc2 = (h2 + 31) / 32;
    return rt_passes(nc2) * nc1 + nc1 * nc2 + (nout <= 4 ? 0 : nc2) + kRtPad;
}

extern "C" int dronesim_mlp_forward_bf16x3(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                                           uint64_t seed, uint64_t counter, int64_t env_base,
                                           const int32_t *t, const int32_t *episode, int E, void *stream)
{
    return mlp_forward_split<SchemeBf16x3>("bf16x3", m, x, out, act, act_idx, seed, counter, env_base, t, episode, E, stream);
}

extern "C" int dronesim_mlp_forward_f16x2(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                                          uint64_t seed, uint64_t counter, int64_t env_base,
                                          const int32_t *t, const int32_t *episode, int E, void *stream)
{
    return mlp_forward_split<SchemeF16x2>("f16x2", m, x, out, act, act_idx, seed, counter, env_base, t, episode, E, stream);
}

// blocks (4 KiB) of one agent's float16 row-tile stream (dronesim_mlp_forward_f16x2_rt): the real blocks rounded up to whole
// super-stages of four, plus three super-stages of padding for the run-ahead of the DMA requests
extern "C" int dronesim_mlp_rt16_blocks(int h1, int h2, int nout)
{
    if (h1 < 1 || h2 < 1 || nout < 1) return 0;
    const int nc1 = (h1 + 31) / 32, nc2 = (h2 + 31) / 32;
    const int real = rt_passes(nc2) * nc1 + nc1 * nc2 + (nout > 4 ? nc2 : 0);
    return ((real + 3) / 4 + 3) * 4;
}

extern "C" int dronesim_mlp_forward_f16x2_rt(const DroneMlpBf16 *m, const float *x, float *out, float *act, int32_t *act_idx,
                                             uint64_t seed, uint64_t counter, int64_t env_base,
                                             const int32_t *t, const int32_t *episode, int E, void *stream)
{
    if (!m || !x) return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward_f16x2_rt: NULL argument");
    const int rc = check_mlp("f16x2_rt", m->N, m->d_in, m->h1, m->h2, m->nout, m->out_kind, m->sample_kind, E);
    if (rc) return rc;
    if (m->d_in > 16) return dronesim_fail(DRONESIM_EUNSUPPORTED, "dronesim_mlp_forward_f16x2_rt: d_in <= 16");
    const bool vl3 = m->nout <= 4;                                // layer 3 on the vector ALU, from the plain float32 array
    if (!m->w1p || (vl3 && !m->w3p) || !m->b1 || !m->b2 || !m->b3)
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward_f16x2_rt: NULL weight array");
    if (m->reserved != dronesim_mlp_rt16_blocks(m->h1, m->h2, m->nout))
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward_f16x2_rt: DroneMlpBf16.reserved must hold dronesim_mlp_rt16_blocks(h1, h2, nout)");
    if ((reinterpret_cast<uintptr_t>(m->w1p) & 15u) != 0)
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward_f16x2_rt: the stream must be 16-byte aligned");
    if (E == 0) return DRONESIM_OK;
    MArgsR16 r{};
    r.E = E; r.N = m->N; r.d_in = m->d_in; r.h1 = m->h1; r.h2 = m->h2; r.nout = m->nout;
    r.nc1 = (m->h1 + 31) / 32; r.nc2 = (m->h2 + 31) / 32;
    r.blocks = m->reserved;
    r.x = x; r.b1 = m->b1; r.b2 = m->b2; r.b3 = m->b3; r.w3 = reinterpret_cast<const float *>(m->w3p); r.wscale = m->wscale;
    r.ws = reinterpret_cast<const char *>(m->w1p);
    r.fin = make_finish(m->N, m->nout, m->out_kind, m->sample_kind, out, act, act_idx, seed, counter, env_base, t, episode);
    r.trace = kTrace ? g_policy_trace : nullptr;
    const size_t lds = (size_t)(r.nc1 + r.nc2) * 32 * 4 + (vl3 ? (size_t)r.nc2 * 32 * 16 : 0) + (size_t)kR16Depth * 16384;
    const unsigned rb = (unsigned)((E + kRtRows - 1) / kRtRows);
    const dim3 grid(rb * m->N);
    r.rb_magic = div_magic(grid.x, rb);
    const int lrc = vl3 ? launch_rt16<true>(r, grid, lds, static_cast<hipStream_t>(stream))
                        : launch_rt16<false>(r, grid, lds, static_cast<hipStream_t>(stream));
    if (lrc) return lrc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dronesim_fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}

extern "C" int dronesim_mlp_forward(const DroneMlp *m, const float *x, float *out, float *act, int32_t *act_idx,
                                    uint64_t seed, uint64_t counter, int64_t env_base,
                                    const int32_t *t, const int32_t *episode, int E, void *stream)
{
    if (!m || !x) return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward: NULL argument");
    const int rc = check_mlp("f32", m->N, m->d_in, m->h1, m->h2, m->nout, m->out_kind, m->sample_kind, E);
    if (rc) return rc;
    if (m->w2_layout == 2) {
        // the row-tile stream (round 6): w2 = [N][dronesim_mlp_rt_blocks(h1, h2)][4][64][4] float32 holding W1, b1, W2 and W3 in the
        // kernel's consumption order; w1 / b1 / w3 are not read
        const bool vl3 = m->nout <= 4;                            // layer 3 on the vector ALU, from the plain w3 array
        if (!m->w2 || !m->b2 || !m->b3 || (vl3 && !m->w3)) return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward: NULL weight array");
        if (m->d_in > 14) return dronesim_fail(DRONESIM_EUNSUPPORTED, "dronesim_mlp_forward: w2_layout = 2 needs d_in <= 14");
        if ((reinterpret_cast<uintptr_t>(m->w2) & 15u) != 0)
            return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward: the row-tile stream (w2_layout = 2) must be 16-byte aligned");
        if (E == 0) return DRONESIM_OK;
        MArgsR r{};
        r.E = E; r.N = m->N; r.d_in = m->d_in; r.h1 = m->h1; r.h2 = m->h2; r.nout = m->nout;
        r.nc1 = (m->h1 + 31) / 32; r.nc2 = (m->h2 + 31) / 32;
        r.blocks = dronesim_mlp_rt_blocks(m->h1, m->h2, m->nout);
        r.x = x; r.ws = m->w2; r.b2 = m->b2; r.b3 = m->b3; r.w3 = m->w3;
        r.fin = make_finish(m->N, m->nout, m->out_kind, m->sample_kind, out, act, act_idx, seed, counter, env_base, t, episode);
        const size_t lds = (size_t)r.nc2 * 32 * 4 * (vl3 ? 5 : 1) + 4 * (size_t)kRtRing * 4096;
        {
            static std::mutex mu;
            static unsigned long long opted[2][4] = {};
            const int lrc = enable_big_lds(vl3 ? reinterpret_cast<const void *>(mlp3_rt_kernel<true>) : reinterpret_cast<const void *>(mlp3_rt_kernel<false>),
                                           opted[vl3 ? 1 : 0], mu, "mlp3_rt_kernel");
            if (lrc) return lrc;
        }
        const unsigned rb = (unsigned)((E + kRtRows - 1) / kRtRows);
        const dim3 grid(rb * m->N);
        r.rb_magic = div_magic(grid.x, rb);
        if (vl3) hipLaunchKernelGGL(mlp3_rt_kernel<true>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), r.x, r.E, r.N, r.d_in, r);
        else hipLaunchKernelGGL(mlp3_rt_kernel<false>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), r.x, r.E, r.N, r.d_in, r);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return dronesim_fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
        return DRONESIM_OK;
    }
    if (!m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3)
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward: NULL weight array");
    if (m->w2_layout != 0 && m->w2_layout != 1) return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward: w2_layout must be 0, 1 or 2");
    if (m->w2_layout == 1 && (reinterpret_cast<uintptr_t>(m->w2) & 15u) != 0)     // read with 16-byte vector loads
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward: fragment-packed w2 (w2_layout = 1) must be 16-byte aligned");
    if (E == 0) return DRONESIM_OK;
    MArgs a{};
    a.trace = kTrace ? g_policy_trace : nullptr;
    a.E = E; a.N = m->N; a.d_in = m->d_in; a.h1 = m->h1; a.h2 = m->h2; a.nout = m->nout;
    a.x = x; a.w1 = m->w1; a.b1 = m->b1; a.w2 = m->w2; a.b2 = m->b2; a.w3 = m->w3; a.b3 = m->b3;
    a.fin = make_finish(m->N, m->nout, m->out_kind, m->sample_kind, out, act, act_idx, seed, counter, env_base, t, episode);
    const bool packed = m->w2_layout == 1;
    const size_t ld1 = packed ? (size_t)packed_row_stride(m->h1) : (size_t)m->h1 + 1;
    // (x rows: d_in + 1 floats; the h1 tile follows on a 16-byte boundary)
    const bool narrow = m->nout <= 16;                           // layer 3 on the 16-column matrix instruction
    const size_t lds = sizeof(float) * ((size_t)kRows * (m->d_in + 1) + (size_t)kRows * ld1 + (kThreadsF / 64) * 32 * (narrow ? kStN : kStW));
    if (lds > 160 * 1024) return dronesim_fail(DRONESIM_EUNSUPPORTED, "hidden layer too wide for the LDS tile");
    typedef void (*Kernel)(const float *, int, int, int, const MArgs);
    static const Kernel kernels[5] = {mlp3_kernel<false, false>, mlp3_kernel<false, true>, mlp3_kernel<true, false>, mlp3_kernel<true, true>,
                                      mlp3_kernel<true, true, 3>};
    const int which = (packed && narrow && m->d_in <= 6) ? 4 : (packed ? 2 : 0) + (narrow ? 1 : 0);
    const Kernel kernel = kernels[which];
    {
        static std::mutex mu;
        static unsigned long long opted[5][4] = {};
        const int lrc = enable_big_lds(reinterpret_cast<const void *>(kernel), opted[which], mu, "mlp3_kernel");
        if (lrc) return lrc;
    }
    const dim3 grid(((E + kRows - 1) / kRows) * m->N);
    a.rb_magic = div_magic(grid.x, (unsigned)((E + kRows - 1) / kRows));
    hipLaunchKernelGGL(kernel, grid, dim3(kThreadsF), lds, static_cast<hipStream_t>(stream), a.x, a.E, a.N, a.d_in, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dronesim_fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}
