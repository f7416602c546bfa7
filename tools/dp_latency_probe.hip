// dp_latency_probe.hip -- issue cadence of dependent vs independent FP64 VALU ops for ONE wave per SIMD
// (the register-heavy stencil kernels run at 1 wave/SIMD, so dependent-issue stalls are not hidden).
// Prints shader-clock cycles per v_add_f64 / v_mul_f64 for K interleaved dependency chains.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ void k_chain(double* out, long long* cyc, double c, int iters)
{
    double x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = threadIdx.x + k;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // alternate add / mul like the stencil; asm volatile keeps the order and defeats reassociation
                if (u & 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c));
                else asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K>
void run(double* out, long long* cyc, int threads)
{
    const int iters = 2000;
    k_chain<K><<<1, threads>>>(out, cyc, 1.0000001, iters);
    k_chain<K><<<1, threads>>>(out, cyc, 1.0000001, iters);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("chains=%d threads=%3d : %.2f cycles per DP instruction (per wave)\n", K, threads, (double)h / ((double)iters * 16 * K));
}

int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    for (int threads : {64, 256, 512}) {
        run<1>(out, cyc, threads); run<2>(out, cyc, threads); run<3>(out, cyc, threads); run<4>(out, cyc, threads); run<8>(out, cyc, threads);
    }
    return 0;
}
