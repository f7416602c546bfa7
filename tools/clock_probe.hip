// clock_probe.hip -- in-kernel shader clock (s_memtime / s_memrealtime) for tiny dependent launches vs
// the same launches issued behind a heavy streaming kernel.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(unsigned long long* out, double* sink, int n)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x * 1e-9;
    for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-12;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (x == 123.456) sink[0] = x;
}
__global__ void k_heavy(double* a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] * 1.0001 + 1.0;
}
int main()
{
    unsigned long long* d; double* sink; double* big; size_t N = 1ull << 28;
    hipMalloc(&d, 16); hipMalloc(&sink, 8); hipMalloc(&big, N * 8);
    hipMemset(big, 0, N * 8);
    unsigned long long h[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: 2000 tiny launches back to back; mode 1: same after 200 ms of heavy kernels; mode 2: interleaved heavy+tiny
        if (mode == 1) for (int i = 0; i < 300; ++i) k_heavy<<<2048, 256>>>(big, N);
        hipEventRecord(e0);
        for (int i = 0; i < 2000; ++i) {
            if (mode == 2 && (i % 50) == 0) k_heavy<<<2048, 256>>>(big, N);
            k_probe<<<289, 256>>>(d, sink, 2000);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %.2f us per launch; in-kernel: %llu shader cycles over %llu x10ns -> %.0f MHz\n", mode, ms * 1e3 / 2000,
               h[0], h[1], (double)h[0] / ((double)h[1] * 10e-9) / 1e6);
    }
    return 0;
}
