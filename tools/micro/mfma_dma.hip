// Micro-benchmark: what does a weight stream by LDS DMA cost a wave that is issuing matrix instructions back to back?
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_dma tools/micro/mfma_dma.hip && ./build/mfma_dma
// 512 workgroups x 4 waves (2 waves per SIMD, like mlp3_bf16x3_kernel): per stage 12 independent-enough
// v_mfma_f32_32x32x16_bf16 and PARTS x 1 KiB global_load_lds into a per-wave ring (4 stages), from a per-wave stream in
// L2-resident memory (SAME = 1: always the same 3 KiB, i.e. L1 hits).  Reports time, matrix instructions per SIMD-cycle
// at the shader clock measured in the kernel (s_memtime ticks per s_memrealtime tick x 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int PARTS, int SAME>
__global__ void __launch_bounds__(256, 2) k(const bf16x8 *w, float *out, long long *clk, int stages, int region_vec)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * 4 * 3072;
    const bf16x8 *gp = w + (size_t)((blockIdx.x & 63) * 4 + wave) * region_vec + lane;
    const bf16x8 *gend = gp + region_vec - 192;
    f32x16 a0 = {}, a1 = {};
    bf16x8 f = {1, 2, 3, 4, 5, 6, 7, 8}, g = {8, 7, 6, 5, 4, 3, 2, (short)lane};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int slot = 0;
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, g, a0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g, f, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (m < PARTS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + m * 64),
                                                 (__attribute__((address_space(3))) void *)(ring + slot * 3072 + m * 1024), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!SAME) { gp += 192; if (gp > gend) gp -= region_vec - 192; }
        slot = (slot + 1) & 3;
        if (PARTS) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PARTS * 2) : "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    float sacc = 0; for (int r = 0; r < 16; ++r) sacc += a0[r] + a1[r];
    if (sacc == 12345.f) out[threadIdx.x] = sacc;
}

template <int PARTS, int SAME> void run(const bf16x8 *w, float *out, long long *clk, int region_vec)
{
    const int stages = 1000, wgs = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<PARTS, SAME>), dim3(wgs), dim3(256), 48 * 1024, 0, w, out, clk, stages, region_vec);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    long long h[2 * 4096]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double ticks = 0, rt = 0; for (int i = 0; i < wgs; ++i) { ticks += h[2 * i]; rt += h[2 * i + 1]; }
    const double mfma = (double)wgs * 4 * stages * 12;
    printf("parts %d same %d: %8.1f us  | s_memtime/s_memrealtime = %.2f (x100 MHz) | %.1f cycles per matrix instruction per SIMD at 2.4 GHz\n",
           PARTS, SAME, ms * 1e3, ticks / rt, ms * 1e-3 * 2.4e9 * 1024 / mfma);
}

int main()
{
    const int region_vec = 1200 * 192;             // 3.6 MB per (agent, wave) stream; 64 x 4 streams = 920 MB?  no: keep L2-sized below
    const int rv = 96 * 192;                       // 288 KiB per stream, 73 MB in all: L2/MALL resident
    (void)region_vec;
    bf16x8 *w; float *out; long long *clk;
    hipMalloc(&w, (size_t)64 * 4 * rv * 16); hipMalloc(&out, 4096); hipMalloc(&clk, 2 * 4096 * sizeof(long long));
    hipMemset(w, 0, (size_t)64 * 4 * rv * 16);
    run<0, 1>(w, out, clk, rv); run<1, 1>(w, out, clk, rv); run<3, 1>(w, out, clk, rv);
    run<1, 0>(w, out, clk, rv); run<2, 0>(w, out, clk, rv); run<3, 0>(w, out, clk, rv);
    return 0;
}
