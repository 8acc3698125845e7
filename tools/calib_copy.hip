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

// ---- launch-floor probes (developer): how long does a dependent launch of B workgroups x T threads take
// when each wave only (a) writes one dword, (b) streams `rd` bytes in and `wr` bytes out per thread?
__global__ void probe_empty(unsigned *out)
{
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = 1u;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ void probe_stream(const f32x2_t *__restrict__ a, const f32x2_t *__restrict__ b, f32x2_t *__restrict__ o, int nout)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const f32x2_t x = a[i], y = b[i];
    const size_t n = (size_t)gridDim.x * blockDim.x;
    for (int k = 0; k < nout; ++k) {
        f32x2_t v = x + y * (float)k;
        if (NT) __builtin_nontemporal_store(v, o + (size_t)k * n + i); else o[(size_t)k * n + i] = v;
    }
}
extern "C" {
int probe_launch_empty(void *out, int blocks, int threads, void *stream)
{
    hipLaunchKernelGGL(probe_empty, dim3(blocks), dim3(threads), 0, static_cast<hipStream_t>(stream), (unsigned *)out);
    return (int)hipGetLastError();
}
int probe_launch_stream(const void *a, const void *b, void *o, int blocks, int threads, int nout, int nt, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nt) hipLaunchKernelGGL(probe_stream<true>, dim3(blocks), dim3(threads), 0, s, (const f32x2_t *)a, (const f32x2_t *)b, (f32x2_t *)o, nout);
    else hipLaunchKernelGGL(probe_stream<false>, dim3(blocks), dim3(threads), 0, s, (const f32x2_t *)a, (const f32x2_t *)b, (f32x2_t *)o, nout);
    return (int)hipGetLastError();
}
}
