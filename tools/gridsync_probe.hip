// What does a grid-wide barrier cost on this chip?  hipLaunchCooperativeKernel + cooperative_groups::grid_group::sync()
// (the runtime checks that the whole grid is resident, so the barrier cannot deadlock), for several grid sizes, against
// the cost of a dependent kernel launch boundary.  Build: hipcc -O3 --offload-arch=gfx950 tools/gridsync_probe.hip -o gridsync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ void k_sync(int iters, double* out)
{
    cg::grid_group g = cg::this_grid();
    double v = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        v = v * 1.0000001 + 1.0;
        g.sync();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// the same barrier written by hand: one device-scope counter, sense by generation (all blocks are resident: cooperative launch)
__global__ void k_sync_hand(int iters, double* out, unsigned* ctr)
{
    double v = threadIdx.x;
    const unsigned nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        v = v * 1.0000001 + 1.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned target = (unsigned)(i + 1) * nb;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// barrier + all-reduce as in k_cg_persistent: every block stores a value into its own slot, lanes 0..nb-1 of wave 0 poll one
// slot each until it is no longer EMPTY; four slot sets rotate (see csrc/multigrid2d.hip, cgp_allsum)
__global__ void k_sync_slots(int iters, double* out, unsigned long long* slots)
{
    const unsigned long long EMPTY = 0x7ff8dead0badf00dull;
    __shared__ double g[256];
    double v = threadIdx.x;
    const int nb = gridDim.x;
    for (int i = 1; i <= iters; ++i) {
        v = v * 1.0000001 + 1.0;
        unsigned long long* set = slots + (i & 3) * 256;
        unsigned long long* nxt = slots + ((i + 2) & 3) * 256;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&nxt[blockIdx.x], EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&set[blockIdx.x], (unsigned long long)__double_as_longlong(v) & ~1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int w = threadIdx.x; w < nb; w += blockDim.x) {
            unsigned long long b;
            while ((b = __hip_atomic_load(&set[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == EMPTY) { }
            g[w] = __longlong_as_double((long long)b);
        }
        __syncthreads();
        double s = 0.0;
        for (int w = 0; w < nb; ++w) s += g[w];
        v += s * 1e-300;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// counter barrier, then the values by ordinary coherent loads (two round trips)
__global__ void k_sync_counter_values(int iters, double* out, unsigned* ctr, double* vals)
{
    __shared__ double g[256];
    double v = threadIdx.x;
    const unsigned nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        v = v * 1.0000001 + 1.0;
        double* set = vals + (i & 1) * 256;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&set[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * nb;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { }
        }
        __syncthreads();
        for (int w = threadIdx.x; w < (int)nb; w += blockDim.x) g[w] = __hip_atomic_load(&set[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        double s = 0.0;
        for (int w = 0; w < (int)nb; ++w) s += g[w];
        v += s * 1e-300;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

__global__ void k_empty(double* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1.0; }

// a kernel with a few microseconds of real work: every thread streams over `n` doubles of an L2-resident array
__global__ void k_work(double* a, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    for (int k = i; k < n; k += gridDim.x * blockDim.x) v += a[k];
    if (v == 12345.678) a[0] = v;
    if (i < n) a[i] = a[i] * 1.0000001 + 1e-9;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main()
{
    double* out; unsigned* ctr;
    CK(hipMalloc(&out, 4096 * sizeof(double)));
    CK(hipMalloc(&ctr, sizeof(unsigned)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int threads : {256, 1024}) {
        for (int blocks : {16, 32, 64, 128, 256}) {
            int it = iters;
            void* args[] = {&it, &out};
            // warm-up + timed
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, s));
                CK(hipLaunchCooperativeKernel((const void*)k_sync, dim3(blocks), dim3(threads), args, 0, s));
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemsetAsync(ctr, 0, sizeof(unsigned), s));
            void* args2[] = {&it, &out, &ctr};
            CK(hipEventRecord(e0, s));
            CK(hipLaunchCooperativeKernel((const void*)k_sync_hand, dim3(blocks), dim3(threads), args2, 0, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
            unsigned long long* slots;
            CK(hipMalloc(&slots, 4 * 256 * 8));
            std::vector<unsigned long long> em(4 * 256, 0x7ff8dead0badf00dull);
            CK(hipMemcpy(slots, em.data(), em.size() * 8, hipMemcpyHostToDevice));
            void* args3[] = {&it, &out, &slots};
            CK(hipEventRecord(e0, s));
            CK(hipLaunchCooperativeKernel((const void*)k_sync_slots, dim3(blocks), dim3(threads), args3, 0, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms3; CK(hipEventElapsedTime(&ms3, e0, e1));
            double* vals;
            CK(hipMalloc(&vals, 2 * 256 * 8));
            CK(hipMemsetAsync(ctr, 0, sizeof(unsigned), s));
            void* args4[] = {&it, &out, &ctr, &vals};
            CK(hipEventRecord(e0, s));
            CK(hipLaunchCooperativeKernel((const void*)k_sync_counter_values, dim3(blocks), dim3(threads), args4, 0, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms4; CK(hipEventElapsedTime(&ms4, e0, e1));
            CK(hipFree(slots)); CK(hipFree(vals));
            printf("grid %4d x %4d threads: grid.sync() %.2f us, counter barrier %.2f us, slot barrier + all-reduce %.2f us, counter barrier + "
                   "value loads %.2f us\n", blocks, threads, ms * 1e3 / iters, ms2 * 1e3 / iters, ms3 * 1e3 / iters, ms4 * 1e3 / iters);
        }
    }
    {   // dependent launch boundary
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) k_empty<<<64, 256, 0, s>>>(out);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dependent launches of an empty kernel: %.2f us per launch\n", ms * 1e3 / iters);
    }
    {   // the same chain captured into a hipGraph and replayed
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < iters; ++i) k_empty<<<64, 256, 0, s>>>(out);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("the same %d dependent launches replayed from a hipGraph: %.2f us per kernel node\n", iters, ms * 1e3 / iters);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    {   // the same comparison for a kernel that does some work (289 blocks over a 512 KB array, the size of a 257^2 level)
        double* arr; const int n = 257 * 257;
        CK(hipMalloc(&arr, n * sizeof(double)));
        CK(hipMemset(arr, 0, n * sizeof(double)));
        float ms_s = 0.f, ms_g = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) k_work<<<289, 256, 0, s>>>(arr, n);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms_s, e0, e1));
        }
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < iters; ++i) k_work<<<289, 256, 0, s>>>(arr, n);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms_g, e0, e1));
        }
        printf("a dependent chain of %d launches of a small working kernel (289 x 256 threads over 512 KB): %.2f us per launch from the "
               "stream, %.2f us per node from a hipGraph\n", iters, ms_s * 1e3 / iters, ms_g * 1e3 / iters);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipFree(arr));
    }
    return 0;
}
