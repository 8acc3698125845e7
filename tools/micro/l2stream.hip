// Micro-benchmark: how fast can every CU stream L2-resident data (the policy kernels' weight fragments)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2stream tools/micro/l2stream.hip && /tmp/l2stream
// Each workgroup (256 threads, 2 per CU) reads one of 64 "agents'" 768 KiB regions, 1 KiB per wave-instruction,
// like mlp3_bf16x3_kernel does; mode 0 = global_load_dwordx4 into registers (8 in flight per wave),
// mode 1 = global_load_lds_dwordx4 into a per-wave LDS ring (9 in flight).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ void __launch_bounds__(256, 2) stream_kernel(const u32x4 *w, unsigned *out, int region_vec, int reps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32x4 *base = w + (size_t)(blockIdx.x & 63) * region_vec + wave * (region_vec / 4) + lane;
    const int n = region_vec / 4 / 64;                 // 1 KiB pieces of this wave's quarter
    u32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < n; i += 8) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = base[(size_t)(i + j) * 64];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc ^= v[j];
            }
    } else {
        char *ring = smem + wave * 12 * 1024;
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < n; i += 3) {
                const int slot = (i / 3) & 3;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)min(i + j, n - 1) * 64),
                                                     (__attribute__((address_space(3))) void *)(ring + slot * 3072 + j * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                acc ^= *reinterpret_cast<const u32x4 *>(ring + ((slot + 1) & 3) * 3072 + lane * 16);
            }
    }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc.y ^ acc.z ^ acc.w;
}

int main()
{
    const int region_vec = 768 * 1024 / 16;
    u32x4 *w; unsigned *out;
    hipMalloc(&w, (size_t)64 * region_vec * 16); hipMalloc(&out, 4096);
    hipMemset(w, 1, (size_t)64 * region_vec * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {512, 4096}) {
            const int reps = wgs == 512 ? 8 : 1;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(wgs), dim3(256), 0, 0, w, out, region_vec, reps);
                else hipLaunchKernelGGL(stream_kernel<1>, dim3(wgs), dim3(256), 48 * 1024, 0, w, out, region_vec, reps);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double bytes = (double)wgs * reps * region_vec * 16;
                if (it == 2) printf("mode %d wgs %d: %.1f us, %.2f TB/s from L2, %.1f B/cycle/CU at 2.4 GHz\n", mode, wgs, ms * 1e3,
                                    bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
            }
        }
    return 0;
}
