// micro-benchmark: issue rate of plain vs packed fp32 VALU on gfx950 (developer tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f2 sv = {s, s};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 8 independent plain FMAs
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) {   // 4 packed FMAs = the same 8 results
            asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv));
        } else if (MODE == 2) {   // 8 v_cndmask + 8 or-like (mask building)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, 0, 1, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, 0, 1, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, 0, 1, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, 0, 1, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
        } else if (MODE == 3) {   // transcendental: 8 v_log_f32
            asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 4) {   // 8 plain adds
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 5) {   // 4 packed adds
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE> void run(const char *name, int wps)
{
    const int blocks = 256 * wps, iters = 20000;   // wps workgroups of 4 waves per CU -> wps waves per SIMD
    float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f); hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double results = (double)iters * 8 * wps;             // result-instructions (8 per iter) per SIMD
    printf("%-18s waves/SIMD %d: %.3f ms -> %.2f ns per result-instr per SIMD (= cycles at 1 GHz)\n", name, wps, ms, ms * 1e6 / results);
    hipFree(out);
}

__global__ void clk(long long *o, int iters) {
    long long c0 = __builtin_readcyclecounter(); long long w0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
    long long c1 = __builtin_readcyclecounter(); long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = c1 - c0; o[1] = w1 - w0; }
    if (a == 12345.f) o[2] = 1;
}
int main()
{
    { long long *o; hipMalloc(&o, 64); hipLaunchKernelGGL(clk, dim3(1024), dim3(256), 0, 0, o, 200000); long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
      printf("s_memtime ticks %lld, wall_clock64 ticks %lld (100 MHz) -> s_memtime runs at %.1f MHz; %d dependent FMAs -> %.2f memtime-ticks per FMA\n", h[0], h[1], (double)h[0] / h[1] * 100.0, 200000, (double)h[0] / 200000); }
    for (int w : {8}) { run<0>("v_fma_f32 x8", 8); run<4>("v_add_f32 x8", 8); run<1>("v_pk_fma_f32 x4", 8); }
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32 x8", 1); run<1>("v_pk_fma_f32 x4", 1); run<4>("v_add_f32 x8", 1); run<5>("v_pk_add_f32 x4", 1); run<2>("cmp+cndmask x4", 1); run<3>("v_log_f32 x8", 1); }
        if (w == 2) { run<0>("v_fma_f32 x8", 2); run<1>("v_pk_fma_f32 x4", 2); }
        if (w == 4) { run<0>("v_fma_f32 x8", 4); run<1>("v_pk_fma_f32 x4", 4); run<4>("v_add_f32 x8", 4); run<5>("v_pk_add_f32 x4", 4); run<2>("cmp+cndmask x4", 4); run<3>("v_log_f32 x8", 4); }
    }
    return 0;
}
