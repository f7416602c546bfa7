// bw_probe.hip -- what does the access width cost a two-in / one-out streaming pass on MI355X?
// The 2D multigrid passes read two (4097 x 4097) arrays and write one with 8 bytes per lane, because rows of a
// (2^k + 1)-wide array are only 8-byte aligned.  This probe times out[i] = a[i] + b[i] over the same 134 MB arrays with
//   v8      : 8 B per lane (global_load_dwordx2)
//   v16     : 16 B per lane, 16-byte aligned
//   v16m    : 16 B per lane, every access 8 bytes off alignment (what an odd row looks like)
//   rows8 / rows16s : the access pattern of the march -- each wave walks down rows of 4097 doubles;
//            rows16s shifts the lane <-> column map by one element on odd rows so that every access is aligned
// GB/s = 3 * n * 8 / t.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_v8(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void k_v16(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o, size_t n2, size_t off)
{
    // element offset `off` (0 or 1) shifts every 16-byte access by 8 bytes
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 x = *reinterpret_cast<const double2*>(a + 2 * i + off), y = *reinterpret_cast<const double2*>(b + 2 * i + off);
        *reinterpret_cast<double2*>(o + 2 * i + off) = make_double2(x.x + y.x, x.y + y.y);
    }
}
// row march: wave w of the grid owns columns [64*strip, 64*strip+64) (8 B) or [128*strip, +128) (16 B) and rows [y0, y0+rpc)
__global__ __launch_bounds__(256) void k_rows8(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o, int nx, int ny, int rpc)
{
    const int lane = threadIdx.x & 63, strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = strip * 64 + lane;
    if (c >= nx) return;
    const int y0 = blockIdx.y * rpc, y1 = y0 + rpc < ny ? y0 + rpc : ny;
#pragma unroll 4
    for (int j = y0; j < y1; ++j) { const size_t id = (size_t)nx * j + c; o[id] = a[id] + b[id]; }
}
__global__ __launch_bounds__(256) void k_rows16s(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o, int nx, int ny, int rpc)
{
    const int lane = threadIdx.x & 63, strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int y0 = blockIdx.y * rpc, y1 = y0 + rpc < ny ? y0 + rpc : ny;
#pragma unroll 4
    for (int j = y0; j < y1; ++j) {
        // first element of the row image that is 16-byte aligned: shift by one on rows whose start is 8 bytes off
        const size_t row = (size_t)nx * j;
        const size_t e = ((row + (size_t)strip * 128) & ~(size_t)1) + 2 * lane;     // aligned pair
        if (e + 1 < row + nx && e >= row) {
            const double2 x = *reinterpret_cast<const double2*>(a + e), y = *reinterpret_cast<const double2*>(b + e);
            *reinterpret_cast<double2*>(o + e) = make_double2(x.x + y.x, x.y + y.y);
        }
    }
}
int main()
{
    const int nx = 4097, ny = 4097;
    const size_t n = (size_t)nx * ny;
    double *a, *b, *o;
    CK(hipMalloc(&a, (n + 8) * 8)); CK(hipMalloc(&b, (n + 8) * 8)); CK(hipMalloc(&o, (n + 8) * 8));
    CK(hipMemset(a, 0, (n + 8) * 8)); CK(hipMemset(b, 0, (n + 8) * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            if (ms < best) best = ms;
        }
        printf("%-10s %8.1f us  %7.0f GB/s\n", name, best * 1e3, 3.0 * n * 8 / (best * 1e-3) / 1e9);
    };
    for (int grid : {2048, 4096, 8192}) {
        printf("grid %d blocks\n", grid);
        time("v8", [&] { k_v8<<<grid, 256>>>(a, b, o, n); });
        time("v16", [&] { k_v16<<<grid, 256>>>(a, b, o, n / 2, 0); });
        time("v16m", [&] { k_v16<<<grid, 256>>>(a, b, o, n / 2 - 1, 1); });
    }
    for (int rpc : {32, 64, 128}) {
        printf("row march, %d rows per chunk\n", rpc);
        time("rows8", [&] { k_rows8<<<dim3((nx + 255) / 256, (ny + rpc - 1) / rpc), 256>>>(a, b, o, nx, ny, rpc); });
        time("rows16s", [&] { k_rows16s<<<dim3((nx + 511) / 512, (ny + rpc - 1) / rpc), 256>>>(a, b, o, nx, ny, rpc); });
    }
    return 0;
}
