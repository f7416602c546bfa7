// place_probe.hip -- does the speed of a streaming pass depend on WHICH allocation it runs on?  (The same k_diff3_march2 launch
// takes 0.775 or 0.855 ms depending on the allocation its five arrays came from, tools/diffusion_tune f2place; TLB counters are
// equal.)  K separate 1 GiB allocations; per array: read-only pass, write-only pass; then read i / write j for all pairs of the
// first few.  GB/s each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ a, size_t n2, double* __restrict__ out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const d2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(d2* __restrict__ a, size_t n2, double v)
{
    d2 t; t.x = v; t.y = v;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(t, &a[i]);
}
__global__ __launch_bounds__(256) void k_copy2(const d2* __restrict__ a, d2* __restrict__ b, size_t n2)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(a[i], &b[i]);
}
int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 10;
    const size_t B = (size_t)1 << 30, n2 = B / 16;
    std::vector<d2*> arr(K);
    double* out; CK(hipMalloc(&out, 8));
    // optional spacer allocations of argv[2] GiB between the arrays (reserved, never touched): candidates spread over the card's memory
    const size_t spacer = argc > 2 ? (size_t)atol(argv[2]) << 30 : 0;
    std::vector<void*> sp;
    for (int k = 0; k < K; ++k) {
        CK(hipMalloc(&arr[k], B)); CK(hipMemset(arr[k], 0, B));
        if (spacer && k + 1 < K) { void* q = nullptr; if (hipMalloc(&q, spacer) == hipSuccess) sp.push_back(q); else { printf("spacer %d failed\n", k); (void)hipGetLastError(); } }
    }
    printf("# %d arrays of 1 GiB, %zu spacers of %zu GiB\n", K, sp.size(), spacer >> 30);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / 10;
    };
    for (int rep = 0; rep < 1; ++rep)
        for (int k = 0; k < K; ++k) {
            const float r = time([&] { k_read<<<2048, 256>>>(arr[k], n2, out); });
            const float w = time([&] { k_write<<<2048, 256>>>(arr[k], n2, 1.0); });
            printf("array %2d at %p: read %7.1f GB/s  write %7.1f GB/s\n", k, (void*)arr[k], B / (r * 1e-3) / 1e9, B / (w * 1e-3) / 1e9);
        }
    const int P = K < 12 ? K : 12;
    for (int i = 0; i < P; ++i) {
        printf("copy from %d to:", i);
        for (int j = 0; j < P; ++j) {
            if (i == j) { printf("      -"); continue; }
            const float c = time([&] { k_copy2<<<2048, 256>>>(arr[i], arr[j], n2); });
            printf(" %6.0f", 2.0 * B / (c * 1e-3) / 1e9);
        }
        printf("  GB/s\n");
    }
    return 0;
}
