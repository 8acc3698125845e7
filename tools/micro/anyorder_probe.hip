// Does a launch without the AQL barrier bit (hipExtAnyOrderLaunch) overlap its predecessor on gfx950,
// eagerly and inside a captured graph?  Two kernels: an empty 1024 x 256 launch (the dispatch floor) and a
// launch whose waves each sit for ~20 us on the 100 MHz clock (overlap shows as a per-launch time below that).
//   hipcc --offload-arch=gfx950 -O2 -o anyorder_probe anyorder_probe.hip && ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(int* sink) { if (sink == (int*)1) *sink = 0; }

__global__ void sit_kernel(int* sink, int ticks) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink == (int*)1) *sink = 0;
}

template <class F>
static float time_eager(hipStream_t s, int n, F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 50; ++i) launch(s);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < n; ++i) launch(s);
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / n;
}

template <class F>
static float time_graph(hipStream_t s, int n, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / (5 * n);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int* sink = nullptr;
    const int n = 1000;
    for (int flag = 0; flag < 2; ++flag) {
        unsigned f = flag ? hipExtAnyOrderLaunch : 0;
        auto e = [&](hipStream_t st) {
            hipExtLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, st, nullptr, nullptr, f, sink);
        };
        auto k = [&](hipStream_t st) {
            hipExtLaunchKernelGGL(sit_kernel, dim3(1024), dim3(256), 0, st, nullptr, nullptr, f, sink, 2000);
        };
        printf("flag=%s  empty: eager %.2f us  graph %.2f us   sit(20us): eager %.2f us  graph %.2f us\n",
               flag ? "anyorder" : "ordered ", time_eager(s, n, e), time_graph(s, n, e),
               time_eager(s, n, k), time_graph(s, n, k));
    }
    auto plain = [&](hipStream_t st) { empty_kernel<<<1024, 256, 0, st>>>(sink); };
    printf("plain <<<>>> empty: eager %.2f us graph %.2f us\n", time_eager(s, n, plain), time_graph(s, n, plain));
    return 0;
}
