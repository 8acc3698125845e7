// Does v_mov_b32_dpp wave_rol:1 / wave_ror:1 rotate over all 64 lanes on gfx950 (with wrap-around)?
//   hipcc --offload-arch=gfx950 -O2 -o build/dpp_wave_rot tools/micro/dpp_wave_rot.hip && build/dpp_wave_rot
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o)
{
    const unsigned v = threadIdx.x;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(0xdeadu, v, 0x134, 0xf, 0xf, false);        // wave_rol:1
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0xdeadu, v, 0x13C, 0xf, 0xf, false);   // wave_ror:1
}
int main()
{
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_rol:1 lane 0,1,2,31,32,62,63 <- %u %u %u %u %u %u %u\n", h[0], h[1], h[2], h[31], h[32], h[62], h[63]);
    printf("wave_ror:1 lane 0,1,2,31,32,62,63 <- %u %u %u %u %u %u %u\n", h[64], h[65], h[66], h[95], h[96], h[126], h[127]);
    int ok_l = 1, ok_r = 1;
    for (int i = 0; i < 64; ++i) { ok_l &= h[i] == (unsigned)((i + 1) & 63) || h[i] == (unsigned)((i + 63) & 63); }
    for (int i = 0; i < 64; ++i) { ok_r &= h[64 + i] == (unsigned)((i + 63) & 63) || h[64 + i] == (unsigned)((i + 1) & 63); }
    printf("full 64-lane rotation with wrap: rol %s, ror %s; opposite directions: %s\n", ok_l ? "yes" : "NO", ok_r ? "yes" : "NO",
           h[1] != h[65] ? "yes" : "NO");
    return 0;
}
