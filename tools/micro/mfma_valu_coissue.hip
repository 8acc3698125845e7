// Micro-benchmark: can the float32 matrix pipe and the packed-float32 vector pipe of a SIMD work at the same time?
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_valu_coissue tools/micro/mfma_valu_coissue.hip && ./build/mfma_valu_coissue
// Both pipes have the same float32 peak on gfx950 (v_mfma_f32_32x32x2_f32: 4096 flop per 64 cycles per SIMD;
// v_pk_fma_f32 on a wave64: 256 flop per 4 cycles per SIMD = 157 TFLOP/s each at 2.4 GHz x 1024 SIMDs).  The exact-float32
// policy kernel (csrc/policy.hip, mlp3_kernel) uses the matrix pipe only; if a second wave of the SIMD could run a
// k-ordered v_pk_fma_f32 GEMM on other output columns meanwhile, the float32-exact ceiling of a CU would double.
// Workgroups of 8 waves (2 per SIMD): mode 0 = all waves matrix, 1 = all waves vector, 2 = waves 0-3 matrix + waves 4-7
// vector (one of each per SIMD), 3 = every wave alternates 1 matrix instruction with 16 vector ones.
// Reports per mode: time, matrix and vector TFLOP/s, and the shader clock during the kernel (s_memtime ticks per
// s_memrealtime tick x 100 MHz) -- a co-issue that only lowers the clock is no gain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float *out, long long *clk, int iters)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    f32x16 a0 = {}, a1 = {};
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)lane, (float)i};
    float x = 1.0f + lane * 1e-3f, y = 0.5f;
    f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < iters; ++s) {
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                }
            }
            if (do_valu) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    float sacc = 0;
    for (int r = 0; r < 16; ++r) sacc += a0[r] + a1[r] + v[r].x + v[r].y;
    if (sacc == 12345.f) out[threadIdx.x] = sacc;
}

template <int MODE> void run(float *out, long long *clk)
{
    const int iters = 20000, wgs = 512;               // 512 x 8 waves = 4 waves per SIMD in two workgroups per CU
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(512), 0, 0, out, clk, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    std::vector<long long> h(2 * wgs); hipMemcpy(h.data(), clk, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
    double ticks = 0, rt = 0; for (int i = 0; i < wgs; ++i) { ticks += h[2 * i]; rt += h[2 * i + 1]; }
    const double waves = (double)wgs * 8;
    const double mw = MODE == 0 || MODE == 3 ? waves : MODE == 2 ? waves / 2 : 0;      // waves issuing matrix instructions
    const double vw = MODE == 1 || MODE == 3 ? waves : MODE == 2 ? waves / 2 : 0;
    const double mflop = mw * iters * 8 * 4096.0;                                      // 8 x 32x32x2 per iteration
    const double vflop = vw * iters * 128 * 256.0;                                     // 128 x pk_fma (wave64: 256 flop)
    const char *names[] = {"all waves matrix", "all waves vector", "4 waves matrix + 4 waves vector per CU-workgroup", "every wave interleaves"};
    printf("mode %d (%s): %8.1f us | matrix %6.1f TFLOP/s  vector %6.1f TFLOP/s  sum %6.1f | shader clock %.2f GHz\n", MODE, names[MODE],
           ms * 1e3, mflop / ms / 1e9, vflop / ms / 1e9, (mflop + vflop) / ms / 1e9, ticks / rt * 0.1);
}

int main()
{
    float *out; long long *clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, sizeof(long long) * 2 * 4096);
    run<0>(out, clk); run<1>(out, clk); run<2>(out, clk); run<3>(out, clk);
    return 0;
}
