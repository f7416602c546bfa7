// event_probe.hip -- what does timing every launch cost the launches?  A stream of identical copy kernels (~0.7 ms each)
//   (a) plain back to back, (b) a hipEventRecord pair around every launch, (c) hipExtLaunchKernelGGL with start / stop events
// attached to the dispatch itself.  Prints wall time per launch and the events' own average.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_copy4(const double4* __restrict__ a, double4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ul + threadIdx.x; i < n; i += gridDim.x * 256ul) b[i] = a[i];
}
int main()
{
    const size_t n = (size_t)2 << 26;   // 2 x 1 GiB per launch... 128 Mi double4 = 4 GiB read + 4 GiB written is too long: use 64 Mi
    const size_t m = n / 2;
    double4 *a, *b;
    hipMalloc(&a, m * sizeof(double4)); hipMalloc(&b, m * sizeof(double4));
    hipMemset(a, 0, m * sizeof(double4));
    hipStream_t s; hipStreamCreate(&s);
    const int K = 60;
    std::vector<hipEvent_t> ev(2 * K);
    for (auto& e : ev) hipEventCreate(&e);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int i = 0; i < 20; ++i) k_copy4<<<2048, 256, 0, s>>>(a, b, m);
    hipStreamSynchronize(s);
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            const double t0 = now();
            for (int i = 0; i < K; ++i) {
                if (mode == 0) k_copy4<<<2048, 256, 0, s>>>(a, b, m);
                else if (mode == 1) { hipEventRecord(ev[2 * i], s); k_copy4<<<2048, 256, 0, s>>>(a, b, m); hipEventRecord(ev[2 * i + 1], s); }
                else hipExtLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, a, b, m);
            }
            hipStreamSynchronize(s);
            const double wall = (now() - t0) / K * 1e6;
            double evavg = 0.0;
            if (mode) { for (int i = 0; i < K; ++i) { float ms = 0; hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); evavg += ms; } evavg = evavg / K * 1e3; }
            printf("%s: wall %.1f us per launch%s", mode == 0 ? "plain            " : mode == 1 ? "hipEventRecord x2" : "hipExtLaunch evts", wall, mode ? "" : "\n");
            if (mode) printf(", events' average %.1f us\n", evavg);
        }
    return 0;
}
