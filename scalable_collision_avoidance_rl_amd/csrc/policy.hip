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
// Decomposition: grid = (ceil(E/64), N): one workgroup = 64 env rows of ONE agent, 4 waves.
//   wave w owns every fourth 32-column chunk of the hidden layers for all 64 rows (two 32x32
//   accumulators that share every B fragment).
//   layer 1: A = x tile (LDS), B = W1 (global/L2)            -> h1 tile in LDS [64][h1+1]
//   layer 2 chunk (32 columns): A = h1 (LDS), B = W2          -> relu -> per-wave LDS staging [64][33]
//   layer 3 partial: A = staged chunk, B = W3 rows of the chunk -> accumulated in registers
//   the four waves' partials are summed through LDS, then activation + sampling.
// LDS row strides are odd (h1+1, 33) so the 32-row fragment reads are bank-conflict free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "dronesim.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kRows = 64;            // env rows per workgroup
constexpr int kMaxOut = 32;

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
};

// C/D layout of v_mfma_f32_32x32x2_f32: element reg r of lane l is (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31)
__device__ __forceinline__ int cd_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Output activation + sampling of ONE env row (y = pre-activation outputs of this row's agent network).

__device__ __forceinline__ void finish_row(const FinishArgs &a, float (&y)[kMaxOut], int e, int agent)
{
    const int nout = a.nout;
    if (a.out_kind == 1) {                                   // softmax (utils.py:286, dim = 0 of one sample)
        float m = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < kMaxOut; ++j) if (j < nout) m = fmaxf(m, y[j]);
        float ssum = 0.0f;
#pragma unroll
        for (int j = 0; j < kMaxOut; ++j) { y[j] = j < nout ? expf(y[j] - m) : 0.0f; ssum += y[j]; }
        const float inv = 1.0f / ssum;
#pragma unroll
        for (int j = 0; j < kMaxOut; ++j) y[j] *= inv;
    } else if (a.out_kind == 2) {                            // tanh means, sigmoid variances (utils.py:74-77)
        const int half = nout / 2;
#pragma unroll
        for (int j = 0; j < kMaxOut; ++j)
            if (j < nout) y[j] = j < half ? tanhf(y[j]) : 1.0f / (1.0f + expf(-y[j]));
    }
    const size_t row = (size_t)e * a.N + agent;
    if (a.out) {
#pragma unroll
        for (int j = 0; j < kMaxOut; ++j) if (j < nout) a.out[row * nout + j] = y[j];
    }
    if (a.sample_kind != 0) {
        uint32_t rnd[4];
        const uint32_t c2 = a.ctr2 + (a.t_dev ? (uint32_t)a.t_dev[e] : 0u);
        const uint32_t c3 = a.ctr3 + (a.episode_dev ? (uint32_t)a.episode_dev[e] : 0u);
        philox4x32_10((uint32_t)agent, (uint32_t)(a.env_base + e), c2, c3, a.key0, a.key1, rnd);
        if (a.sample_kind == 1) {                            // categorical -> unit vector (utils.py:262-269, 304-309)
            const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
            float cdf = 0.0f;
            int pick = nout - 1;
            bool found = false;
#pragma unroll
            for (int j = 0; j < kMaxOut; ++j) {
                if (j < nout) {
                    cdf += y[j];
                    if (!found && u < cdf) { pick = j; found = true; }
                }
            }
            if (a.act_idx) a.act_idx[row] = pick;
            if (a.act) {
                const float ang = (float)pick / (float)nout * 6.283185307179586f;
                a.act[row * 2 + 0] = cosf(ang);
                a.act[row * 2 + 1] = sinf(ang);
            }
        } else {                                             // Gaussian, Box-Muller (utils.py:110-117)
            const int half = nout / 2;
            for (int d = 0; d < half && d < 2; ++d) {
                const float u1 = ((float)(rnd[2 * d] >> 8) + 1.0f) * (1.0f / 16777216.0f);    // (0, 1]
                const float u2 = (float)(rnd[2 * d + 1] >> 8) * (1.0f / 16777216.0f);
                const float n01 = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
                if (a.act) a.act[row * half + d] = fmaf(sqrtf(y[half + d]), n01, y[d]);
            }
        }
    }
}

// acc += A[32 x K] * B[K x 32].  A row-major in LDS (lda floats per row, odd stride -> conflict-free), B row-major
// in global / L2 (ldb floats per row); B columns >= ncols_valid read a clamped column (never stored), k >= K reads
// as zero.  Operand loads of the next 8 k-steps are issued before the 8 MFMAs of the current ones.
constexpr int kU = 8;                  // k-steps (of 2) per pipeline stage

struct Frag { float b[kU], a[kU]; };

__device__ __forceinline__ void load_frag(Frag &f, const float *Arow, const float *__restrict__ Bcol, int ldb, int kbase)
{
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int k = kbase + 2 * u;
        f.b[u] = Bcol[(size_t)k * ldb];
        f.a[u] = Arow[k];
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
        load_frag(cur, Arow, Bcol, ldb, kk);
        for (int k0 = 0; k0 < Kmain; k0 += 2 * kU) {
            const bool more = k0 + 2 * kU < Kmain;         // wave-uniform
            if (more) load_frag(nxt, Arow, Bcol, ldb, k0 + 2 * kU + kk);
#pragma unroll
            for (int u = 0; u < kU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u], cur.b[u], acc, 0, 0, 0);
            if (more) cur = nxt;
        }
    }
    for (int k0 = Kmain; k0 < K; k0 += 2) {                // tail (K not a multiple of 16)
        const int k = k0 + kk;
        const bool kok = k < K;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kok ? Arow[k] : 0.0f, kok ? Bcol[(size_t)k * ldb] : 0.0f, acc, 0, 0, 0);
    }
}

// 64 env rows of one agent per workgroup, 8 waves = two per SIMD (one's operand loads hide behind the other's
// MFMAs): wave w owns feature chunks (w & 3), (w & 3) + 4, ... for the 32 rows of half (w >> 2).
__global__ void __launch_bounds__(512) mlp3_kernel(const MArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, rh = wave >> 2;
    const int agent = blockIdx.y;
    const int e0 = blockIdx.x * kRows;
    const int ldx = a.d_in + 1, ld1 = a.h1 + 1;
    float *sx = reinterpret_cast<float *>(smem);                 // [64][d_in+1]
    float *sh1 = sx + kRows * ldx;                               // [64][h1+1]
    float *sst = sh1 + kRows * ld1;                              // [8 waves][32][33] layer-2 chunk staging,
                                                                 // reused for the layer-3 partials
    const float *w1 = a.w1 + (size_t)agent * a.d_in * a.h1, *b1 = a.b1 + (size_t)agent * a.h1;
    const float *w2 = a.w2 + (size_t)agent * a.h1 * a.h2, *b2 = a.b2 + (size_t)agent * a.h2;
    const float *w3 = a.w3 + (size_t)agent * a.h2 * a.nout, *b3 = a.b3 + (size_t)agent * a.nout;

    // ---- x tile -> LDS (rows beyond E are zero)
    for (int idx = tid; idx < kRows * a.d_in; idx += 512) {
        const int r = idx / a.d_in, c = idx - r * a.d_in;
        const int e = e0 + r;
        sx[r * ldx + c] = e < a.E ? a.x[((size_t)e * a.N + agent) * a.d_in + c] : 0.0f;
    }
    __syncthreads();

    const int col = lane & 31;
    // ---- layer 1
    for (int c0 = cw * 32; c0 < a.h1; c0 += 128) {
        f32x16 acc = {0};
        tile_gemm(acc, sx + rh * 32 * ldx, ldx, w1 + c0, a.h1, a.d_in, a.h1 - c0, lane);
        const bool ok = c0 + col < a.h1;
        const float bias = ok ? b1[c0 + col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok) sh1[(rh * 32 + cd_row(r, lane)) * ld1 + c0 + col] = fmaxf(acc[r] + bias, 0.0f);
    }
    __syncthreads();

    // ---- layers 2 + 3 fused over this wave's column chunks
    f32x16 y = {0};
    float *st = sst + wave * 32 * 33;
    for (int c0 = cw * 32; c0 < a.h2; c0 += 128) {
        const bool ok = c0 + col < a.h2;
        const float bias = ok ? b2[c0 + col] : 0.0f;             // issued before the k-loop, needed after it
        f32x16 acc = {0};
        tile_gemm(acc, sh1 + rh * 32 * ld1, ld1, w2 + c0, a.h2, a.h1, a.h2 - c0, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[cd_row(r, lane) * 33 + col] = ok ? fmaxf(acc[r] + bias, 0.0f) : 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int kc = min(32, a.h2 - c0);
        tile_gemm(y, st, 33, w3 + (size_t)c0 * a.nout, a.nout, kc, a.nout, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) st[cd_row(r, lane) * 33 + col] = y[r];   // this wave's partial outputs
    __syncthreads();

    // ---- output activation + sampling: one thread per env row
    if (tid < kRows) {
        const int e = e0 + tid;
        if (e >= a.E) return;
        const int rhh = tid >> 5, rr = tid & 31;
        float yv[kMaxOut];
        const int nout = a.nout;
#pragma unroll
        for (int j = 0; j < kMaxOut; ++j) {
            float v = 0.0f;
            if (j < nout) {
                v = b3[j];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += sst[((rhh * 4 + w) * 32 + rr) * 33 + j];
            }
            yv[j] = v;
        }
        finish_row(a.fin, yv, e, agent);
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 variant (opt-in): weights and activations in bfloat16, float32 accumulation, on
// v_mfma_f32_32x32x16_bf16 (16x the float32 matrix rate).  Formulated transposed -- D[feature][env row] =
// W^T (A operand, pre-packed per fragment on the host, one 16-byte load per lane) x activations (B operand,
// LDS, [row][k] row-major, one ds_read_b128 per lane) -- so that a lane's 4 consecutive accumulator
// registers are 4 consecutive features of ONE env row and leave as one packed 8-byte LDS store.
// One workgroup = 64 env rows (2 row tiles sharing every weight fragment) of one agent, 4 waves; wave w owns
// feature chunks w, w+4, ...  63 KiB of LDS per workgroup keeps two workgroups per CU resident, so one's
// prologue / epilogue / barriers overlap the other's MFMAs.
// LDS row strides are odd multiples of 16 bytes (conflict-free b128 reads).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kRowsB = 64, kTiles = 2;        // 64 env rows (2 row tiles) per workgroup: 63 KiB of LDS, two workgroups per CU
constexpr int kLdx = 24, kLds = 40;      // bf16 per row of the x tile / of a staged 32-feature chunk

struct MArgsB {
    int E, N, d_in, h1, h2, nc1, nc2, ks1;
    const float *x, *b1, *b2, *b3;
    const bf16x8 *w1p, *w2p, *w3p;       // [agent][chunk][k-step][64 lanes] fragments
    FinishArgs fin;
};

// acc[t] += W^T chunk (k-steps [0, ks)) x activation rows of tile t.
// One wave per SIMD has nobody to hide latency behind, so operands are fetched one stage (2 k-steps:
// 2 weight fragments from L2, 8 activation fragments from LDS) ahead of the 8 MFMAs that consume them,
// ping-ponging between two register sets.
struct StageB { bf16x8 a[2]; bf16x8 b[2][kTiles]; };

__device__ __forceinline__ void load_stage(StageB &st, const bf16x8 *__restrict__ wfrag, const __bf16 *brow, int ld,
                                           int s, int lane)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        st.a[h] = wfrag[(size_t)(s + h) * 64 + lane];
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
            st.b[h][t] = *reinterpret_cast<const bf16x8 *>(brow + t * 32 * ld + (s + h) * 16);
    }
}

__device__ __forceinline__ void mma_stage(f32x16 (&acc)[kTiles], const StageB &st)
{
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(st.a[h], st.b[h][t], acc[t], 0, 0, 0);
}

__device__ __forceinline__ void chunk_gemm(f32x16 (&acc)[kTiles], const bf16x8 *__restrict__ wfrag, int ks,
                                           const __bf16 *act, int ld, int lane)
{
    const __bf16 *brow = act + (lane & 31) * ld + 8 * (lane >> 5);
    const int kse = ks & ~1;                               // whole stages
    if (kse > 0) {
        StageB p, q;
        load_stage(p, wfrag, brow, ld, 0, lane);
        int s = 0;
        while (true) {
            if (s + 2 < kse) load_stage(q, wfrag, brow, ld, s + 2, lane);
            mma_stage(acc, p);
            s += 2;
            if (s >= kse) break;
            if (s + 2 < kse) load_stage(p, wfrag, brow, ld, s + 2, lane);
            mma_stage(acc, q);
            s += 2;
            if (s >= kse) break;
        }
    }
    if (ks & 1) {                                          // odd tail (layer 1: a single k-step)
        const bf16x8 a0 = wfrag[(size_t)kse * 64 + lane];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(brow + t * 32 * ld + kse * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[t], 0, 0, 0);
        }
    }
}

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

__global__ void __launch_bounds__(256, 2) mlp3_bf16_kernel(const MArgsB a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int agent = blockIdx.y;
    const int e0 = blockIdx.x * kRowsB;
    const int ld1 = a.nc1 * 32 + 8;
    __bf16 *sx = reinterpret_cast<__bf16 *>(smem);                 // [64][24]
    __bf16 *sst = sx + kRowsB * kLdx;                              // [4 waves][64][40]
    float *sbias = reinterpret_cast<float *>(sst + 4 * kRowsB * kLds);   // [nc1*32 + nc2*32] zero-padded biases
    __bf16 *sh1 = reinterpret_cast<__bf16 *>(sbias + (a.nc1 + a.nc2) * 32);   // [64][ld1]; later f32 partials [4][64][33]
    float *spart = reinterpret_cast<float *>(sh1);

    for (int idx = tid; idx < kRowsB * 16; idx += 256) {           // x tile, zero padded to k = 16
        const int r = idx >> 4, c = idx & 15;
        const int e = e0 + r;
        const float v = (c < a.d_in && e < a.E) ? a.x[((size_t)e * a.N + agent) * a.d_in + c] : 0.0f;
        sx[r * kLdx + c] = (__bf16)v;
    }
    for (int idx = tid; idx < (a.nc1 + a.nc2) * 32; idx += 256) {  // biases once, so no epilogue waits on L2
        const int f = idx < a.nc1 * 32 ? idx : idx - a.nc1 * 32;
        sbias[idx] = idx < a.nc1 * 32 ? (f < a.h1 ? a.b1[(size_t)agent * a.h1 + f] : 0.0f)
                                      : (f < a.h2 ? a.b2[(size_t)agent * a.h2 + f] : 0.0f);
    }
    __syncthreads();

    // ---- layer 1 -> sh1 (bf16): wave w owns feature chunks w, w+4, ...
    for (int c = wave; c < a.nc1; c += 4) {
        f32x16 acc[kTiles] = {};
        chunk_gemm(acc, a.w1p + ((size_t)agent * a.nc1 + c) * a.ks1 * 64, a.ks1, sx, kLdx, lane);
        store_chunk(acc, sbias + c * 32, sh1 + c * 32, ld1, lane);
    }
    __syncthreads();

    // ---- layers 2 + 3 fused over this wave's feature chunks
    f32x16 y[kTiles] = {};
    __bf16 *st = sst + wave * kRowsB * kLds;
    const int ks2 = a.nc1 * 2;
    for (int c = wave; c < a.nc2; c += 4) {
        f32x16 acc[kTiles] = {};
        chunk_gemm(acc, a.w2p + ((size_t)agent * a.nc2 + c) * ks2 * 64, ks2, sh1, ld1, lane);
        store_chunk(acc, sbias + (a.nc1 + c) * 32, st, kLds, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        chunk_gemm(y, a.w3p + ((size_t)agent * a.nc2 * 2 + 2 * c) * 64, 2, st, kLds, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                               // everyone is done reading sh1
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            spart[((size_t)wave * kRowsB + t * 32 + (lane & 31)) * 33 + cd_row(r, lane)] = y[t][r];
    __syncthreads();

    if (tid < kRowsB) {
        const int e = e0 + tid;
        if (e >= a.E) return;
        const float *b3 = a.b3 + (size_t)agent * a.fin.nout;
        float yv[kMaxOut];
#pragma unroll
        for (int j = 0; j < kMaxOut; ++j) {
            float v = 0.0f;
            if (j < a.fin.nout) {
                v = b3[j];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += spart[((size_t)w * kRowsB + tid) * 33 + j];
            }
            yv[j] = v;
        }
        finish_row(a.fin, yv, e, agent);
    }
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
    a.E = E; a.N = m->N; a.d_in = m->d_in; a.h1 = m->h1; a.h2 = m->h2;
    a.nc1 = (m->h1 + 31) / 32; a.nc2 = (m->h2 + 31) / 32; a.ks1 = 1;
    a.x = x; a.b1 = m->b1; a.b2 = m->b2; a.b3 = m->b3;
    a.w1p = reinterpret_cast<const bf16x8 *>(m->w1p); a.w2p = reinterpret_cast<const bf16x8 *>(m->w2p);
    a.w3p = reinterpret_cast<const bf16x8 *>(m->w3p);
    a.fin = make_finish(m->N, m->nout, m->out_kind, m->sample_kind, out, act, act_idx, seed, counter, env_base, t, episode);
    const size_t sh1_bytes = sizeof(__bf16) * kRowsB * ((size_t)a.nc1 * 32 + 8);
    const size_t part_bytes = sizeof(float) * 4 * kRowsB * 33;
    const size_t lds = sizeof(__bf16) * (kRowsB * kLdx + 4 * kRowsB * kLds) + sizeof(float) * 32 * (a.nc1 + a.nc2) +
                       (sh1_bytes > part_bytes ? sh1_bytes : part_bytes);
    static bool big_lds_enabled = false;
    if (!big_lds_enabled) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp3_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return dronesim_fail(DRONESIM_ELAUNCH, "cannot enable 160 KiB of dynamic LDS for mlp3_bf16_kernel");
        big_lds_enabled = true;
    }
    if (lds > 160 * 1024) return dronesim_fail(DRONESIM_EUNSUPPORTED, "hidden layer too wide for the LDS tile");
    hipLaunchKernelGGL(mlp3_bf16_kernel, dim3((E + kRowsB - 1) / kRowsB, m->N), dim3(256), lds, static_cast<hipStream_t>(stream), a);
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
    if (!m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3)
        return dronesim_fail(DRONESIM_EINVAL, "dronesim_mlp_forward: NULL weight array");
    if (E == 0) return DRONESIM_OK;
    MArgs a{};
    a.E = E; a.N = m->N; a.d_in = m->d_in; a.h1 = m->h1; a.h2 = m->h2; a.nout = m->nout;
    a.x = x; a.w1 = m->w1; a.b1 = m->b1; a.w2 = m->w2; a.b2 = m->b2; a.w3 = m->w3; a.b3 = m->b3;
    a.fin = make_finish(m->N, m->nout, m->out_kind, m->sample_kind, out, act, act_idx, seed, counter, env_base, t, episode);
    const size_t lds = sizeof(float) * ((size_t)kRows * (m->d_in + 1) + (size_t)kRows * (m->h1 + 1) + 8 * 32 * 33);
    static bool big_lds_enabled = false;                 // > 64 KiB of dynamic LDS must be opted into once
    if (!big_lds_enabled) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return dronesim_fail(DRONESIM_ELAUNCH, "cannot enable 160 KiB of dynamic LDS for mlp3_kernel");
        big_lds_enabled = true;
    }
    if (lds > 160 * 1024) return dronesim_fail(DRONESIM_EUNSUPPORTED, "hidden layer too wide for the LDS tile");
    hipLaunchKernelGGL(mlp3_kernel, dim3((E + kRows - 1) / kRows, m->N), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dronesim_fail(DRONESIM_ELAUNCH, hipGetErrorString(e));
    return DRONESIM_OK;
}
