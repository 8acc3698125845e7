// Internal helpers shared by the translation units of libdronesim.so (not part of the C ABI).
#pragma once

// The only build switches of the product source: developer trace builds (per-wave phase stamps; tools/trace_*.py) and one A/B build,
//   make -C csrc -j8 EXTRA=-DDRONESIM_TRACE OUT=../../build/libdronesim_trace.so OBJDIR=../../build/obj_trace
#if defined(DRONESIM_TRACE)
constexpr bool kTrace = true;
#else
constexpr bool kTrace = false;
#endif
// -DDRONESIM_TRACE_SPAN: ONLY the entry / exit time of every wave on the chip-wide 100 MHz clock (s_memrealtime), with the product's
// code otherwise unchanged (hoisting on): first-wave-in -> last-wave-out span of a launch inside a graph replay (tools/trace_span.py)
#if defined(DRONESIM_TRACE_SPAN)
constexpr bool kTraceSpan = true;
#else
constexpr bool kTraceSpan = false;
#endif
#if defined(DRONESIM_TRACE_FINE)
constexpr bool kTraceFine = true;
#else
constexpr bool kTraceFine = false;
#endif
// -DDRONESIM_PAIR_PARALLEL_STEP: the single-step kernels of one-env-per-wave launches (kSym64 step / observe) ALSO go through the
// pair-parallel near-pair phase that the fused rollouts use (drone_kernel.hpp: measured slower there; the A/B build of round 6)
#if defined(DRONESIM_PAIR_PARALLEL_STEP)
constexpr bool kPairParallelStep = true;
#else
constexpr bool kPairParallelStep = false;
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

// records the thread-local error string returned by dronesim_last_error() and returns `code`
__attribute__((visibility("hidden"))) int dronesim_fail(int code, const char *msg);

// 32 x 32 -> 64-bit product (multiplier: a wave-uniform constant)
__device__ __forceinline__ uint64_t mul_wide_u32(uint32_t k, uint32_t x)
{
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "s"(k), "v"(x) : "vcc");
    return r;
}

// Philox4x32-10 (Salmon et al., SC'11), all four output words
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (the 64-bit products as ONE v_mad_u64_u32 each, by name: from __umulhi + a 32-bit multiply -- or from a 64-bit
        // multiply in the source -- hipcc issues two quarter-rate multiplies for most of them, 37 instead of 20 in the ten rounds)
        const uint64_t p0 = mul_wide_u32(0xD2511F53u, c0), p1 = mul_wide_u32(0xCD9E8D57u, c2);
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- the arithmetic of ONE ordered pair (i, j) of distance_data (drone_env.py:309-332), templated on the scalar type:
// the float32 instantiation is the hot kernels' pass 2, the float64 one is the verification kernel (drone_kernel_f64,
// test-only) that is compared with the float64 oracle without any float32-state allowance.
template <typename Real> struct RealOps;
template <> struct RealOps<float> {
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }     // v_log_f32 = log2
    static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return fmaf(a, b, c); }
    static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
};
template <> struct RealOps<double> {
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
    static __device__ __forceinline__ double min(double a, double b) { return ::fmin(a, b); }
};

template <typename Real> struct PairTerms { Real d, lg; bool coll, inm; };

// d2 = |x_i - x_j|^2 of the pair; l_i, l_j radii; dhat_i and log2(dhat_i) of the ROW; Delta_j of the COLUMN (Q1)
template <typename Real>
__device__ __forceinline__ PairTerms<Real> pair_terms(Real d2, Real li, Real lj, Real dhat, Real log2_dhat, Real delta_j)
{
    PairTerms<Real> r;
    const Real dist = RealOps<Real>::sqrt(d2);
    Real d = RealOps<Real>::min(dist - li - lj, dhat);                        // :318
    d = (d == Real(0)) ? Real(-1e-6) : d;                                     // :319-320
    r.coll = d < Real(0);                                                     // :327 (dhat > 0)
    // log(dhat/d) = ln2 * (log2 dhat - log2 d); collisions contribute 9990 (:330-332)
    r.lg = r.coll ? Real(9.99e3) : Real(0.693147180559945309417232121458) * (log2_dhat - RealOps<Real>::log2(d));
    r.inm = d <= delta_j;                                                     // :328 (Delta_j!)
    r.d = d;
    return r;
}

