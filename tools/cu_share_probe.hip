// Do the compute units a long launch leaves WITHOUT a workgroup really serve a second stream?  A "hog" kernel (G workgroups
// of 512 threads, one per compute unit by its LDS request, busy for a given time -- spinning, or streaming through a
// buffer to load the memory system) on one stream; a "guest" kernel (128 workgroups of 256 threads, 16 KiB LDS) on a
// second stream a little later.  Every guest workgroup records when it started and ended (s_memrealtime, 100 MHz) and where
// it ran (HW_ID, XCC_ID); the hog records where its workgroups ran.
// Build: hipcc -O3 --offload-arch=gfx950 tools/cu_share_probe.hip -o cu_share_probe ; usage: cu_share_probe [G=244] [us=600] [stream_mode=0|1] [hog_fills_its_unit=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20); }

template <bool HEAVY>   // HEAVY: the kernel declares 256 VGPRs, so its 8 waves fill every SIMD of their compute unit (as k_diff3_march2 does)
__global__ __launch_bounds__(512, 1) void k_hog(long ticks, const double* buf, size_t n, int mode, unsigned* where, long long* t0_out, double* sink, long long* starts)
{
    extern __shared__ double lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) starts[blockIdx.x] = t0;
    if constexpr (HEAVY) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (threadIdx.x == 0) { where[2 * blockIdx.x] = hw_id(); where[2 * blockIdx.x + 1] = xcc_id(); if (blockIdx.x == 0) *t0_out = t0; }
    double acc = 0.0;
    size_t i = ((size_t)blockIdx.x * 512 + threadIdx.x) * 2;
    const size_t stride = (size_t)gridDim.x * 512 * 2;
    while (wall_clock64() - t0 < ticks) {
        if (mode) {
            for (int r = 0; r < 8; ++r) {
                const double2 v = *reinterpret_cast<const double2*>(buf + (i % n));
                acc += v.x + v.y;
                i += stride;
            }
        } else {
            acc = acc * 1.0000001 + 1.0;
        }
    }
    lds[threadIdx.x] = acc;
    if (acc == 12345.678) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_guest(const double* src, double* dst, size_t per_wg, long long* rec, unsigned* where)
{
    __shared__ double tile[2048];
    const long long t0 = wall_clock64();
    // a little real work: copy per_wg doubles through LDS
    const double* s = src + (size_t)blockIdx.x * per_wg;
    double* d = dst + (size_t)blockIdx.x * per_wg;
    for (size_t o = 0; o < per_wg; o += 2048) {
        for (int k = threadIdx.x; k < 2048; k += 256) tile[k] = s[o + k];
        __syncthreads();
        for (int k = threadIdx.x; k < 2048; k += 256) d[o + k] = tile[k] + 1.0;
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { rec[2 * blockIdx.x] = t0; rec[2 * blockIdx.x + 1] = t1; where[2 * blockIdx.x] = hw_id(); where[2 * blockIdx.x + 1] = xcc_id(); }
}

int main(int argc, char** argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 244;
    const double us = argc > 2 ? atof(argv[2]) : 600.0;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int heavy = argc > 4 ? atoi(argv[4]) : 1;
    const int rbits = argc > 5 ? atoi(argv[5]) : 0;   // > 0: CU masks -- the guest's stream gets mask bits [rlo, rlo + rbits), the hog's stream all the others
    const int rlo = argc > 6 ? atoi(argv[6]) : 0;
    const int reps = argc > 7 ? atoi(argv[7]) : 1;
    const int NG = 128;
    const size_t per_wg = 16384;   // 128 KiB per guest workgroup
    const size_t nbuf = (size_t)1 << 27;   // 1 GiB
    double *buf, *gsrc, *gdst, *sink;
    unsigned *hwhere, *gwhere;
    long long *rec, *t0, *hstarts;
    CK(hipMalloc(&hstarts, 8 * 1024));
    CK(hipMalloc(&buf, nbuf * 8)); CK(hipMemset(buf, 0, nbuf * 8));
    CK(hipMalloc(&gsrc, NG * per_wg * 8)); CK(hipMalloc(&gdst, NG * per_wg * 8)); CK(hipMemset(gsrc, 0, NG * per_wg * 8));
    CK(hipMalloc(&sink, 8)); CK(hipMalloc(&hwhere, 2 * 4 * 1024)); CK(hipMalloc(&gwhere, 2 * 4 * NG));
    CK(hipMalloc(&rec, 2 * 8 * NG)); CK(hipMalloc(&t0, 8));
    hipStream_t sa, sb;
    if (rbits > 0) {
        uint32_t ma[8], mb[8];
        for (int i = 0; i < 8; ++i) { ma[i] = 0xffffffffu; mb[i] = 0; }
        for (int b = rlo; b < rlo + rbits; ++b) { ma[b / 32] &= ~(1u << (b % 32)); mb[b / 32] |= 1u << (b % 32); }
        CK(hipExtStreamCreateWithCUMask(&sa, 8, ma));
        CK(hipExtStreamCreateWithCUMask(&sb, 8, mb));
    } else {
        CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    }
    CK(hipFuncSetAttribute((const void*)k_hog<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024));
    CK(hipFuncSetAttribute((const void*)k_hog<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024));
    auto hog = [&](long tk) {
        if (heavy) k_hog<true><<<G, 512, 82 * 1024, sa>>>(tk, buf, nbuf, mode, hwhere, t0, sink, hstarts);
        else k_hog<false><<<G, 512, 82 * 1024, sa>>>(tk, buf, nbuf, mode, hwhere, t0, sink, hstarts);
    };
    const long ticks = (long)(us * 100.0);   // 100 MHz
    // warm both
    hog(100);
    k_guest<<<NG, 256, 0, sb>>>(gsrc, gdst, per_wg, rec, gwhere);
    CK(hipDeviceSynchronize());
    // guest alone
    k_guest<<<NG, 256, 0, sb>>>(gsrc, gdst, per_wg, rec, gwhere);
    CK(hipDeviceSynchronize());
    std::vector<long long> r(2 * NG);
    CK(hipMemcpy(r.data(), rec, 2 * 8 * NG, hipMemcpyDeviceToHost));
    long long mn = r[0], mx = r[1];
    for (int i = 0; i < NG; ++i) { mn = std::min(mn, r[2 * i]); mx = std::max(mx, r[2 * i + 1]); }
    printf("guest alone: %d workgroups, first start -> last end %.1f us\n", NG, (mx - mn) / 100.0);
    // hog, then the guest beside it
    for (int rep = 1; rep < reps; ++rep) { hog(ticks / 10); k_guest<<<NG, 256, 0, sb>>>(gsrc, gdst, per_wg, rec, gwhere); }
    hog(ticks);
    k_guest<<<NG, 256, 0, sb>>>(gsrc, gdst, per_wg, rec, gwhere);
    CK(hipDeviceSynchronize());
    long long T0;
    CK(hipMemcpy(&T0, t0, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), rec, 2 * 8 * NG, hipMemcpyDeviceToHost));
    std::vector<unsigned> hw(2 * G), gw(2 * NG);
    CK(hipMemcpy(hw.data(), hwhere, 2 * 4 * G, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gw.data(), gwhere, 2 * 4 * NG, hipMemcpyDeviceToHost));
    // compute unit key: XCC (0-7), SE (HW_ID bits 13..15), SH (12), CU (8..11)
    auto key = [](unsigned h, unsigned x) { return ((x & 0xf) << 8) | (((h >> 13) & 7) << 5) | (((h >> 12) & 1) << 4) | ((h >> 8) & 0xf); };
    std::vector<int> hogkeys;
    for (int i = 0; i < G; ++i) hogkeys.push_back(key(hw[2 * i], hw[2 * i + 1]));
    std::sort(hogkeys.begin(), hogkeys.end());
    const int distinct_hog = (int)(std::unique(hogkeys.begin(), hogkeys.end()) - hogkeys.begin());
    int on_hog_cu = 0;
    std::vector<int> gk;
    for (int i = 0; i < NG; ++i) {
        const int k = key(gw[2 * i], gw[2 * i + 1]);
        gk.push_back(k);
        if (std::binary_search(hogkeys.begin(), hogkeys.begin() + distinct_hog, k)) ++on_hog_cu;
    }
    std::sort(gk.begin(), gk.end());
    const int distinct_guest = (int)(std::unique(gk.begin(), gk.end()) - gk.begin());
    std::vector<double> st, en;
    for (int i = 0; i < NG; ++i) { st.push_back((r[2 * i] - T0) / 100.0); en.push_back((r[2 * i + 1] - T0) / 100.0); }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
    printf("hog: %d workgroups on %d distinct compute units, %.0f us, mode %s, %s\n", G, distinct_hog, us, mode ? "streaming" : "spin", heavy ? "256 VGPRs (fills its unit)" : "few VGPRs");
    {
        std::vector<long long> hs(G);
        CK(hipMemcpy(hs.data(), hstarts, 8 * G, hipMemcpyDeviceToHost));
        long long lo = hs[0]; int late = 0;
        for (int i = 0; i < G; ++i) lo = std::min(lo, hs[i]);
        long long hi = lo;
        for (int i = 0; i < G; ++i) { hi = std::max(hi, hs[i]); late += (hs[i] - lo) > 2000; }
        printf("hog workgroup starts: last %.1f us after the first; %d of %d more than 20 us late (a second round)\n", (hi - lo) / 100.0, late, G);
    }
    printf("guest beside it: on %d distinct compute units (%d of %d workgroups on a unit that also holds a hog workgroup)\n", distinct_guest, on_hog_cu, NG);
    if (rbits > 0) {
        printf("guest units (xcc.se.sh.cu):");
        for (int i = 0; i < distinct_guest; ++i) printf(" %d.%d.%d.%d", gk[i] >> 8, (gk[i] >> 5) & 7, (gk[i] >> 4) & 1, gk[i] & 15);
        printf("\n");
    }
    printf("guest workgroup starts (us after the hog's start): min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f ; last end %.1f\n",
           st[0], st[NG / 4], st[NG / 2], st[3 * NG / 4], st[NG - 1], en[NG - 1]);
    return 0;
}
