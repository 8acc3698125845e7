// Does v_mfma_f32_32x32x16_f16 honour float16 subnormal inputs (gfx950)?  hipcc --offload-arch=gfx950 -O2 ...
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float *out, float av, float bv)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
}
int main()
{
    float *d; hipMalloc(&d, 256);
    const float cases[][2] = {{1.0f, 1.0f}, {0x1p-20f, 1.0f}, {1.0f, 0x1p-24f}, {0x1p-20f, 0x1p-20f}, {0x1p-14f, 1.0f}, {0x1.8p-16f, 4.0f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %g, b = %g: mfma sum over k=16 -> %g (exact %g)\n", c[0], c[1], h, 16.0 * c[0] * c[1]);
    }
    return 0;
}
