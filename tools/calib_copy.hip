// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (developer tool, not product):
// a copy with the SAME access shape as the step kernel's streams (8 bytes per lane, lane-contiguous)
// over a known byte count, so the counters' unit/undercount on gfx950 can be measured, as
// MI355X_MICROARCH.md (HBM section) asks before trusting an absolute.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void calib_copy_b64(const float2 *__restrict__ src, float2 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void calib_copy_b128(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// 24-byte-per-lane strided stores like the z rows (3 x dwordx2 per lane), 12-byte like nbr_idx
__global__ void calib_write_z(float2 *__restrict__ z, int *__restrict__ nbr, size_t n_agents)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_agents) {
        z[3 * i + 0] = make_float2(1.f, 2.f);
        z[3 * i + 1] = make_float2(3.f, 4.f);
        z[3 * i + 2] = make_float2(5.f, 6.f);
        nbr[3 * i + 0] = (int)i; nbr[3 * i + 1] = -1; nbr[3 * i + 2] = -1;
    }
}

extern "C" {
int calib_copy(const void *src, void *dst, size_t bytes, int width, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (width == 8) {
        const size_t n = bytes / 8;
        hipLaunchKernelGGL(calib_copy_b64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           (const float2 *)src, (float2 *)dst, n);
    } else {
        const size_t n = bytes / 16;
        hipLaunchKernelGGL(calib_copy_b128, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           (const float4 *)src, (float4 *)dst, n);
    }
    return (int)hipGetLastError();
}
int calib_write(void *z, void *nbr, size_t n_agents, void *stream)
{
    hipLaunchKernelGGL(calib_write_z, dim3((unsigned)((n_agents + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), (float2 *)z, (int *)nbr, n_agents);
    return (int)hipGetLastError();
}
}
