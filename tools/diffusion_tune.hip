// diffusion_tune.hip -- standalone tuning harness for the fused 3D diffusion kernels (gfx950).
// Runs every kernel variant / tiling of finalprojectrepo.jl_amd/csrc/diffusion3d_kernels.hpp on an
// n^3 grid, checks each against the naive variant bit for bit on the device, and prints the
// effective bandwidth (A_eff = 32 B per interior cell).  Usage: diffusion_tune [n] [iters] [filter]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../finalprojectrepo.jl_amd/csrc/diffusion3d_launch.hpp"
#include "../finalprojectrepo.jl_amd/csrc/diffusion3d_fused2.hpp"
#include <cmath>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void k_fill_rand(double* a, size_t n, unsigned long long seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        a[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}

__global__ void k_count_diff(const double* a, const double* b, size_t n, unsigned long long* cnt)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        c += (__double_as_longlong(a[i]) != __double_as_longlong(b[i]));
    if (c) atomicAdd(cnt, c);
}

__global__ void k_find_diff(const double* a, const double* b, size_t n, unsigned long long* cnt, unsigned long long* idx, int cap)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__double_as_longlong(a[i]) != __double_as_longlong(b[i])) {
            const unsigned long long k = atomicAdd(cnt, 1ull);
            if (k < (unsigned long long)cap) idx[k] = i;
        }
}

typedef double tune_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_copy_nt(const tune_d2* __restrict__ a, tune_d2* __restrict__ b, size_t n2)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(a[i], &b[i]);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int iters = argc > 2 ? atoi(argv[2]) : 10;
    const char* filter = argc > 3 ? argv[3] : "";
    const size_t N = (size_t)n * n * n;
    double *Ht, *Htau, *H2, *dH, *H2ref, *dHref, *parts;
    unsigned long long* cnt;
    CK(hipMalloc(&Ht, N * 8)); CK(hipMalloc(&Htau, N * 8)); CK(hipMalloc(&H2, N * 8)); CK(hipMalloc(&dH, N * 8));
    CK(hipMalloc(&H2ref, N * 8)); CK(hipMalloc(&dHref, N * 8)); CK(hipMalloc(&parts, (1 << 22) * 8));
    CK(hipMalloc(&cnt, 8));
    k_fill_rand<<<2048, 256>>>(Ht, N, 1);
    k_fill_rand<<<2048, 256>>>(Htau, N, 2);
    CK(hipMemset(H2ref, 0, N * 8)); CK(hipMemset(dHref, 0, N * 8));
    const double dx = 10.0 / n;
    Diff3Args a;
    a.Ht = Ht; a.Htau = Htau; a.nx = a.ny = a.nz = n;
    for (int d = 0; d < 3; ++d) { a.lo[d] = 1; a.hi[d] = n - 1; }
    a.dtau = dx * dx / 8.1; a._dt = 5.0; a._dx = a._dy = a._dz = 1 / dx; a.D_dx = a.D_dy = a.D_dz = 1 / dx;
    a.scale = 0.2; a.partials = parts;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int np;
    // reference: naive variant
    { Diff3Tuning t; t.variant = 1; a.Htau2 = H2ref; a.dHdtau = dHref; CK(diff3_launch(a, false, t, s, 1 << 22, &np)); CK(hipStreamSynchronize(s)); }
    const double bytes = 32.0 * (double)(n - 2) * (n - 2) * (n - 2);
    struct Cfg { int variant, ry, nt, xcd, zc, norm, vx; };
    std::vector<Cfg> cfgs;
    cfgs.push_back({1, 0, 0, 0, 0, 0, 0});
    cfgs.push_back({1, 0, 0, 0, 0, 1, 0});
    const int zcs[] = {0, 8, 16, 32, 64, 128, 1 << 20};
    for (int variant : {2, 3})
        for (int ry : {1, 2, 4})
            for (int nt : {0, 1})
                for (int xcd : {0, 1})
                    for (int zc : zcs) cfgs.push_back({variant, ry, nt, xcd, zc, 0, 2});
    for (int variant : {4, 5})
        for (int ry : {2, 4})
            for (int nt : {0, 1})
                for (int zc : {0, 8, 16, 32, 128})
                    for (int nrm : {0, 1}) cfgs.push_back({variant, ry, nt, 0, zc, nrm, 2});
    for (int xcd : {2, 3, 4})
        for (int zc : {0, 16, 32, 128}) cfgs.push_back({3, 4, 1, xcd, zc, 0, 2});
    for (int xcd : {1, 2, 3, 4})
        for (int zc : {8, 16, 32}) cfgs.push_back({5, 4, 1, xcd, zc, 0, 2});
    for (int variant : {2, 3})
        for (int ry : {2, 4}) {
            for (int nt : {0, 1})
                for (int zc : {0, 128}) cfgs.push_back({variant, ry, nt, 0, zc, 1, 2});   // with fused norm
            cfgs.push_back({variant, ry, 0, 1, 0, 0, 1});   // 8-byte path
        }
    printf("# n=%d iters=%d  A_eff bytes/iter=%.4e\n", n, iters, bytes);
    printf("%-8s %3s %3s %3s %3s %6s %5s %9s %9s %6s %s\n", "variant", "vx", "ry", "nt", "xcd", "zc", "norm", "ms", "GB/s", "%peak", "check");
    for (const Cfg& c : cfgs) {
        char name[64];
        snprintf(name, sizeof name, "v%d-vx%d-ry%d-nt%d-xcd%d-zc%d-n%d", c.variant, c.vx, c.ry, c.nt, c.xcd, c.zc, c.norm);
        if (filter[0] && !strstr(name, filter)) continue;
        Diff3Tuning t; t.variant = c.variant; t.ry = c.ry; t.nt = c.nt; t.xcd_remap = c.xcd; t.zc = c.zc; t.vx = c.vx;
        a.Htau2 = H2; a.dHdtau = dH;
        CK(hipMemsetAsync(H2, 0, N * 8, s)); CK(hipMemsetAsync(dH, 0, N * 8, s));
        hipError_t e = diff3_launch(a, c.norm != 0, t, s, 1 << 22, &np);
        if (e != hipSuccess) { printf("%-40s launch failed: %s\n", name, hipGetErrorString(e)); continue; }
        CK(hipMemsetAsync(cnt, 0, 8, s));
        k_count_diff<<<2048, 256, 0, s>>>(H2, H2ref, N, cnt);
        k_count_diff<<<2048, 256, 0, s>>>(dH, dHref, N, cnt);
        unsigned long long bad = 0;
        CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        // warm-up: at least 60 ms of back-to-back launches so the clocks have ramped, then 3 rounds, keep the median
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < 200; ++w) {
            CK(diff3_launch(a, c.norm != 0, t, s, 1 << 22, &np));
            if ((w & 15) == 15) {
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float wm; CK(hipEventElapsedTime(&wm, e0, e1));
                if (wm > 60.f) break;
            }
        }
        float r[3];
        for (int round = 0; round < 3; ++round) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) CK(diff3_launch(a, c.norm != 0, t, s, 1 << 22, &np));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&r[round], e0, e1));
            r[round] /= iters;
        }
        float ms = r[0] > r[1] ? (r[1] > r[2] ? r[1] : (r[0] > r[2] ? r[2] : r[0])) : (r[0] > r[2] ? r[0] : (r[1] > r[2] ? r[2] : r[1]));
        const double gbs = bytes / (ms * 1e-3) / 1e9;
        printf("%-8d %3d %3d %3d %3d %6d %5d %9.4f %9.1f %6.1f %s (blocks=%d)\n", c.variant, c.vx, c.ry, c.nt, c.xcd, c.zc, c.norm, ms, gbs,
               100.0 * gbs / 8000.0, bad ? "MISMATCH" : "ok", np);
        fflush(stdout);
    }

    // ---- fused two-step kernel (diffusion3d_fused2.hpp): checked against two naive steps ----
    if (!filter[0] || strstr("f2", filter) || strstr(filter, "f2")) {
        double *B, *Bref, *Cref, *C, *parts2;
        CK(hipMalloc(&B, N * 8)); CK(hipMalloc(&Bref, N * 8)); CK(hipMalloc(&Cref, N * 8)); CK(hipMalloc(&C, N * 8));
        CK(hipMalloc(&parts2, (1 << 22) * 8));
        k_fill_rand<<<2048, 256, 0, s>>>(B, N, 3);
        CK(hipMemcpyAsync(Bref, B, N * 8, hipMemcpyDeviceToDevice, s));
        CK(hipMemsetAsync(Cref, 0, N * 8, s));
        CK(hipMemsetAsync(dHref, 0, N * 8, s));
        double ref1 = 0, ref2 = 0;
        std::vector<double> hp(1 << 22);
        {   // reference: A -> Bref (keeps B's boundary), Bref -> Cref; norms from the naive kernel's partials
            Diff3Tuning t; t.variant = 1;
            a.Htau = Htau; a.Htau2 = Bref; a.dHdtau = dHref;
            CK(diff3_launch(a, true, t, s, 1 << 22, &np)); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(hp.data(), parts, (size_t)np * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < np; ++i) ref1 += hp[i];
            a.Htau = Bref; a.Htau2 = Cref;
            CK(diff3_launch(a, true, t, s, 1 << 22, &np)); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(hp.data(), parts, (size_t)np * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < np; ++i) ref2 += hp[i];
            a.Htau = Htau;
        }
        int ncu = 256;
        CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
        printf("# compute units: %d\n", ncu);
        Diff3Args2 f;
        f.skip = nullptr;
        f.lane_off = getenv("DIFF3_LANE_OFF") ? atoi(getenv("DIFF3_LANE_OFF")) : 1;
        f.Ht = Ht; f.A = Htau; f.B = B; f.C = C; f.dH = dH;
        f.nx = f.ny = f.nz = n;
        for (int d = 0; d < 3; ++d) { f.lo[d] = 1; f.hi[d] = n - 1; }
        f.dtau = a.dtau; f._dt = a._dt; f._dx = a._dx; f._dy = a._dy; f._dz = a._dz;
        f.D_dx = a.D_dx; f.D_dy = a.D_dy; f.D_dz = a.D_dz; f.scale = a.scale;
        f.partials1 = parts; f.partials2 = parts2;
        if (strstr(filter, "f2class")) {
            // Allocations fall into two classes: a copy between two arrays of the SAME class runs at ~4950 GB/s, between classes at ~5400
            // (tools/place_probe.hip).  Which assignment of classes to the fused kernel's four streams (Ht, A read; C, dH written) is the fast one?
            const size_t AB = N * 8;
            const int K = 12;
            double* arr[K];
            int cls[K];
            for (int k = 0; k < K; ++k) { CK(hipMalloc(&arr[k], AB)); CK(hipMemsetAsync(arr[k], 0, AB, s)); }
            auto copy_ms = [&](int i, int j) {
                for (int w = 0; w < 2; ++w) k_copy_nt<<<2048, 256, 0, s>>>((const tune_d2*)arr[i], (tune_d2*)arr[j], AB / 16);
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < 6; ++r) k_copy_nt<<<2048, 256, 0, s>>>((const tune_d2*)arr[i], (tune_d2*)arr[j], AB / 16);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                return ms / 6;
            };
            float t[K];
            t[0] = 0.f;
            float lo = 1e9f, hi = 0.f;
            for (int k = 1; k < K; ++k) { t[k] = copy_ms(0, k); lo = t[k] < lo ? t[k] : lo; hi = t[k] > hi ? t[k] : hi; }
            const float mid = 0.5f * (lo + hi);
            cls[0] = 0;
            printf("f2class copy 0 -> k [GB/s]:");
            for (int k = 1; k < K; ++k) { cls[k] = t[k] > mid ? 0 : 1; printf(" %.0f(%d)", 2.0 * AB / (t[k] * 1e-3) / 1e9, cls[k]); }
            printf("   (class 0 = like array 0: slow copy)\n");
            if (hi < 1.04f * lo) printf("f2class: all candidates look alike (spread %.1f %%)\n", 100.0 * (hi / lo - 1.0));
            // members of each class
            std::vector<int> m[2];
            for (int k = 0; k < K; ++k) m[cls[k]].push_back(k);
            printf("f2class: %zu arrays of class 0, %zu of class 1\n", m[0].size(), m[1].size());
            for (int pat = 0; pat < 16; ++pat) {
                const int want[4] = {pat & 1, (pat >> 1) & 1, (pat >> 2) & 1, (pat >> 3) & 1};   // Ht, A, C, dH
                int used[2] = {0, 0}, pick[4];
                bool ok = true;
                for (int q = 0; q < 4; ++q) {
                    if (used[want[q]] >= (int)m[want[q]].size()) { ok = false; break; }
                    pick[q] = m[want[q]][used[want[q]]++];
                }
                if (!ok) { printf("f2class pattern Ht%d A%d C%d dH%d: not enough arrays of a class\n", want[0], want[1], want[2], want[3]); continue; }
                CK(hipMemcpyAsync(arr[pick[0]], Ht, AB, hipMemcpyDeviceToDevice, s));
                CK(hipMemcpyAsync(arr[pick[1]], Htau, AB, hipMemcpyDeviceToDevice, s));
                Diff3Args2 g = f;
                g.Ht = arr[pick[0]]; g.A = arr[pick[1]]; g.C = arr[pick[2]]; g.dH = arr[pick[3]];
                for (int w = 0; w < 30; ++w) CK(diff3_launch2(g, true, 0, 0, s, 1 << 22, &np, 0, ncu));
                float best = 1e9f;
                for (int round = 0; round < 3; ++round) {
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 20; ++i) CK(diff3_launch2(g, true, 0, 0, s, 1 << 22, &np, 0, ncu));
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
                    best = ms < best ? ms : best;
                }
                Diff3Args a1 = a; Diff3Tuning t1;
                a1.Ht = g.Ht; a1.Htau = g.A; a1.Htau2 = g.C; a1.dHdtau = g.dH;
                for (int w = 0; w < 10; ++w) CK(diff3_launch(a1, true, t1, s, 1 << 22, &np));
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 20; ++i) CK(diff3_launch(a1, true, t1, s, 1 << 22, &np));
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms1; CK(hipEventElapsedTime(&ms1, e0, e1)); ms1 /= 20;
                printf("f2class pattern Ht%d A%d C%d dH%d: fused %.4f ms/launch, one-iteration kernel %.4f ms\n", want[0], want[1], want[2], want[3], best, ms1);
                fflush(stdout);
            }
            return 0;
        }
        if (strstr(filter, "f2place")) {
            // Does the time of a launch depend on WHERE its five arrays lie?  (Between two processes the same launch took 0.75-0.89 ms on
            // one box; inside a process it repeats to 0.1 %.)  Arrays carved out of ONE slab at base + k * (array size + delta) for several
            // delta, then out of separate allocations made in a fresh order; 3 x 20 launches each.
            const size_t AB = N * 8;
            const size_t deltas[] = {0, 4096, 64 << 10, 256 << 10, (1 << 20) + (64 << 10), (2 << 20), (3 << 20) + (192 << 10), (17 << 20) + (320 << 10)};
            for (int rep = 0; rep < 2; ++rep)
            for (size_t dlt : deltas) {
                char* slab;
                CK(hipMalloc(&slab, 5 * (AB + dlt) + (64 << 20)));
                double* arr[5];
                for (int k = 0; k < 5; ++k) arr[k] = (double*)(slab + (size_t)k * (AB + dlt));
                CK(hipMemcpyAsync(arr[0], Ht, AB, hipMemcpyDeviceToDevice, s));
                CK(hipMemcpyAsync(arr[1], Htau, AB, hipMemcpyDeviceToDevice, s));
                CK(hipMemcpyAsync(arr[2], B, AB, hipMemcpyDeviceToDevice, s));
                CK(hipMemsetAsync(arr[3], 0, AB, s)); CK(hipMemsetAsync(arr[4], 0, AB, s));
                Diff3Args2 g = f;
                g.Ht = arr[0]; g.A = arr[1]; g.B = arr[2]; g.C = arr[3]; g.dH = arr[4];
                for (int w = 0; w < 40; ++w) CK(diff3_launch2(g, true, 0, 0, s, 1 << 22, &np, 0, ncu));
                float best = 1e9f, worst = 0.f;
                for (int round = 0; round < 3; ++round) {
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 20; ++i) CK(diff3_launch2(g, true, 0, 0, s, 1 << 22, &np, 0, ncu));
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
                    best = ms < best ? ms : best; worst = ms > worst ? ms : worst;
                }
                CK(hipMemsetAsync(cnt, 0, 8, s));
                k_count_diff<<<2048, 256, 0, s>>>(arr[3], Cref, N, cnt);
                unsigned long long bad = 0;
                CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
                float ms1 = 0.f;
                {   // the one-iteration kernel on the same arrays (Ht, A -> C, dH) and a plain copy A -> C: slow placement or slow pattern?
                    Diff3Args a1 = a; Diff3Tuning t1;
                    a1.Ht = arr[0]; a1.Htau = arr[1]; a1.Htau2 = arr[3]; a1.dHdtau = arr[4];
                    for (int w = 0; w < 10; ++w) CK(diff3_launch(a1, true, t1, s, 1 << 22, &np));
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 20; ++i) CK(diff3_launch(a1, true, t1, s, 1 << 22, &np));
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms1, e0, e1)); ms1 /= 20;
                }
                float msc = 0.f;
                {
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 10; ++i) CK(hipMemcpyAsync(arr[3], arr[1], AB, hipMemcpyDeviceToDevice, s));
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&msc, e0, e1)); msc /= 10;
                }
                printf("f2place slab delta %9zu B: fused %.4f .. %.4f ms/launch  %s | one-iteration kernel %.4f ms | copy 1 GiB %.4f ms  (slab %p)\n", dlt, best, worst,
                       bad ? "MISMATCH" : "ok", ms1, msc, (void*)slab);
                fflush(stdout);
                CK(hipFree(slab));
            }
            for (int rep = 0; rep < 4; ++rep) {   // separate allocations, with a spacer of another size before each round
                void* spacer; CK(hipMalloc(&spacer, (size_t)(37 + 101 * rep) << 20));
                double* arr[5];
                for (int k = 0; k < 5; ++k) CK(hipMalloc(&arr[k], AB));
                CK(hipMemcpyAsync(arr[0], Ht, AB, hipMemcpyDeviceToDevice, s));
                CK(hipMemcpyAsync(arr[1], Htau, AB, hipMemcpyDeviceToDevice, s));
                CK(hipMemcpyAsync(arr[2], B, AB, hipMemcpyDeviceToDevice, s));
                CK(hipMemsetAsync(arr[3], 0, AB, s)); CK(hipMemsetAsync(arr[4], 0, AB, s));
                Diff3Args2 g = f;
                g.Ht = arr[0]; g.A = arr[1]; g.B = arr[2]; g.C = arr[3]; g.dH = arr[4];
                for (int w = 0; w < 40; ++w) CK(diff3_launch2(g, true, 0, 0, s, 1 << 22, &np, 0, ncu));
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 40; ++i) CK(diff3_launch2(g, true, 0, 0, s, 1 << 22, &np, 0, ncu));
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("f2place separate allocations, round %d: %.4f ms/launch (arrays at %p %p %p %p %p)\n", rep, ms / 40, (void*)arr[0], (void*)arr[1], (void*)arr[2], (void*)arr[3], (void*)arr[4]);
                fflush(stdout);
                for (int k = 0; k < 5; ++k) CK(hipFree(arr[k]));
                CK(hipFree(spacer));
            }
            return 0;
        }
        if (strstr(filter, "f2ab")) {
            // interleaved A/B of launch options (box-to-box and run-to-run noise is +-5 %): usage  f2ab:<nw>,<zc>,<xcd>,<lane_off>:<nw>,<zc>,<xcd>,<lane_off>
            int cfg[2][4] = {{0, 0, 1, 1}, {0, 0, 3, 1}};
            sscanf(strstr(filter, "f2ab") + 4, ":%d,%d,%d,%d:%d,%d,%d,%d", &cfg[0][0], &cfg[0][1], &cfg[0][2], &cfg[0][3], &cfg[1][0], &cfg[1][1], &cfg[1][2], &cfg[1][3]);
            double sum[2] = {0, 0}, sq[2] = {0, 0};
            const int reps = 24;
            for (int w = 0; w < 60; ++w) { f.lane_off = cfg[w & 1][3]; CK(diff3_launch2(f, true, cfg[w & 1][1], cfg[w & 1][2], s, 1 << 22, &np, cfg[w & 1][0], ncu)); }
            for (int rep = 0; rep < reps; ++rep)
                for (int c = 0; c < 2; ++c) {
                    f.lane_off = cfg[c][3];
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 20; ++i) CK(diff3_launch2(f, true, cfg[c][1], cfg[c][2], s, 1 << 22, &np, cfg[c][0], ncu));
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    ms /= 20; sum[c] += ms; sq[c] += (double)ms * ms;
                }
            for (int c = 0; c < 2; ++c) {
                const double m = sum[c] / reps, sd = sqrt(fmax(0.0, sq[c] / reps - m * m));
                printf("f2ab cfg %c (nw=%d zc=%d xcd=%d lane_off=%d): %.4f ms/launch +- %.4f  (%.0f GB/s A_eff)\n", 'A' + c, cfg[c][0], cfg[c][1], cfg[c][2], cfg[c][3], m, sd,
                       2.0 * bytes / (m * 1e-3) / 1e9);
            }
            return 0;
        }
        printf("# fused two-step kernel: ms per LAUNCH (= 2 iterations), GB/s in the per-iteration A_eff metric\n");
        for (int ring : {0, 4, 8})
          for (int xcd : {0, 1, 2, 3, 16, 32, 48})
            for (int zc : {0, 16, 22, 24, 32, 43, 47, 48, 57, 64, 85, 128, 170})
                for (int nrm : {0, 1}) {
                    char name[64];
                    snprintf(name, sizeof name, "f2-r%d-xcd%d-zc%d-n%d", ring, xcd, zc, nrm);
                    if (filter[0] && !strstr(name, filter)) continue;
                    CK(hipMemsetAsync(C, 0, N * 8, s)); CK(hipMemsetAsync(dH, 0, N * 8, s));
                    hipError_t e = diff3_launch2(f, nrm != 0, zc, xcd, s, 1 << 22, &np, ring, ncu);
                    if (e != hipSuccess) { printf("%-24s launch failed: %s\n", name, hipGetErrorString(e)); continue; }
                    CK(hipMemsetAsync(cnt, 0, 8, s));
                    k_count_diff<<<2048, 256, 0, s>>>(C, Cref, N, cnt);
                    k_count_diff<<<2048, 256, 0, s>>>(dH, dHref, N, cnt);
                    unsigned long long bad = 0;
                    CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    double n1 = 0, n2 = 0;
                    if (nrm) {
                        CK(hipMemcpy(hp.data(), parts, (size_t)np * 8, hipMemcpyDeviceToHost));
                        for (int i = 0; i < np; ++i) n1 += hp[i];
                        CK(hipMemcpy(hp.data(), parts2, (size_t)np * 8, hipMemcpyDeviceToHost));
                        for (int i = 0; i < np; ++i) n2 += hp[i];
                    }
                    const bool nbad = nrm && (fabs(n1 - ref1) > 1e-12 * ref1 || fabs(n2 - ref2) > 1e-12 * ref2);
                    CK(hipEventRecord(e0, s));
                    for (int w = 0; w < 200; ++w) {
                        CK(diff3_launch2(f, nrm != 0, zc, xcd, s, 1 << 22, &np, ring, ncu));
                        if ((w & 15) == 15) {
                            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                            float wm; CK(hipEventElapsedTime(&wm, e0, e1));
                            if (wm > 60.f) break;
                        }
                    }
                    float r[3];
                    for (int round = 0; round < 3; ++round) {
                        CK(hipEventRecord(e0, s));
                        for (int i = 0; i < iters; ++i) CK(diff3_launch2(f, nrm != 0, zc, xcd, s, 1 << 22, &np, ring, ncu));
                        CK(hipEventRecord(e1, s));
                        CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&r[round], e0, e1));
                        r[round] /= iters;
                    }
                    float ms = r[0] > r[1] ? (r[1] > r[2] ? r[1] : (r[0] > r[2] ? r[2] : r[0])) : (r[0] > r[2] ? r[0] : (r[1] > r[2] ? r[2] : r[1]));
                    const double gbs = 2.0 * bytes / (ms * 1e-3) / 1e9;
                    printf("%-24s %9.4f ms/launch %9.1f GB/s(A_eff) %6.1f%%  %s%s (blocks=%d)\n", name, ms, gbs, 100.0 * gbs / 8000.0,
                           bad ? "MISMATCH" : "ok", nbad ? " NORM-MISMATCH" : "", np);
                    if (nbad) printf("   norms: %.17g vs %.17g ; %.17g vs %.17g\n", n1, ref1, n2, ref2);
                    if (bad) {
                        printf("   mismatching values: %llu\n", bad);
                        unsigned long long* didx; CK(hipMalloc(&didx, 64 * 8));
                        for (int arr = 0; arr < 2; ++arr) {
                            CK(hipMemsetAsync(cnt, 0, 8, s));
                            k_find_diff<<<2048, 256, 0, s>>>(arr ? dH : C, arr ? dHref : Cref, N, cnt, didx, 64);
                            unsigned long long hidx[64], c2 = 0;
                            CK(hipMemcpyAsync(&c2, cnt, 8, hipMemcpyDeviceToHost, s));
                            CK(hipMemcpyAsync(hidx, didx, 64 * 8, hipMemcpyDeviceToHost, s));
                            CK(hipStreamSynchronize(s));
                            printf("   %s: %llu mismatches; some (x,y,z):", arr ? "dH" : "C", c2);
                            for (unsigned long long q = 0; q < (c2 < 24 ? c2 : 24); ++q)
                                printf(" (%llu,%llu,%llu)", hidx[q] % n, (hidx[q] / n) % n, hidx[q] / ((unsigned long long)n * n));
                            printf("\n");
                        }
                        CK(hipFree(didx));
                    }
                    fflush(stdout);
                }
    }
    return 0;
}
