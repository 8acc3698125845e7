// Developer micro-benchmark (round 6): sustained HBM rates of pure READ, pure WRITE and COPY streams on gfx950, in the shapes of the
// env kernels -- 16 bytes per lane, lane-contiguous, non-temporal -- so that the step / rollout kernels' write-dominated traffic
// (60 of 76 B per agent-step written; 44 of 52 in the fused rollout) can be priced against what the chip sustains for such a mix,
// not only against the 8 TB/s spec.
//   hipcc -O3 --offload-arch=gfx950 -o tools/stream_bw tools/micro/stream_bw.hip && tools/stream_bw
// Two regimes per kind: ONE launch over `big` bytes (2 GiB: sustained) and a hipGraph of 4000 dependent launches (the benchmark's replay
// length: a replay costs 0.1-0.3 ms of idle time whatever it holds, 0.5-1.5 us per launch of a 200-launch graph) over `small` bytes each
// (15.7 MB = what one C3 step launch writes), rotating through a 1.9 GB buffer so that no launch rewrites lines still dirty in cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// kind 0: write only; 1: read only (sum kept alive through a never-true store); 2: copy.  n = number of 16-byte elements.
template <int KIND>
__global__ void __launch_bounds__(256) stream_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n, float seed)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f32x4 acc = {seed, seed, seed, seed};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (KIND == 0) __builtin_nontemporal_store(acc, dst + i);
        else if (KIND == 1) { const f32x4 v = __builtin_nontemporal_load(src + i); acc += v; }
        else __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    }
    if (KIND == 1 && acc.x == 12345.678f) dst[0] = acc;
}

template <int KIND> void launch(const f32x4 *s, f32x4 *d, size_t n, int blocks, hipStream_t st)
{
    hipLaunchKernelGGL(stream_kernel<KIND>, dim3(blocks), dim3(256), 0, st, s, d, n, 1.0f);
}

int main()
{
    const size_t big = (size_t)2 << 30, small = (size_t)64 * 4096 * 60;     // bytes
    f32x4 *a, *b;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big));
    CK(hipMemset(a, 0, big)); CK(hipMemset(b, 0, big));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[3] = {"write", "read ", "copy "};
    for (int kind = 0; kind < 3; ++kind) {
        // (a) one launch over 2 GiB, grid = 8 workgroups per CU
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, st));
            if (kind == 0) launch<0>(a, b, big / 16, 2048, st); else if (kind == 1) launch<1>(a, b, big / 16, 2048, st); else launch<2>(a, b, big / 16, 2048, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double gb = (kind == 2 ? 2.0 : 1.0) * big / 1e9;
        printf("%s one launch over 2 GiB            : %8.3f ms  -> %7.1f GB/s%s\n", names[kind], ts[3], gb / (ts[3] * 1e-3),
               kind == 2 ? "  (read + written)" : "");
        // (b) graph of 4000 dependent launches, 15.7 MB each, one element per thread (the step kernel's grid shape: 4096 waves)
        const int L = 4000;
        const size_t n = small / 16, slots = (big - small) / small;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int l = 0; l < L; ++l) {
            const size_t off = (size_t)(l % slots) * n;
            const int blocks = (int)((n + 255) / 256);
            if (kind == 0) launch<0>(a + off, b + off, n, blocks, st); else if (kind == 1) launch<1>(a + off, b + off, n, blocks, st); else launch<2>(a + off, b + off, n, blocks, st);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        ts.clear();
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / L);
        }
        std::sort(ts.begin(), ts.end());
        const double mb = (kind == 2 ? 2.0 : 1.0) * small / 1e6;
        printf("%s 4000 dependent launches x 15.7 MB: %8.3f us per launch -> %7.1f GB/s\n", names[kind], ts[2] * 1e3, mb / (ts[2] * 1e3) * 1e3);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
