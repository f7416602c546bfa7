// diffusion_tune3.hip -- harness for the three-iteration kernel k_diff3_march3 (csrc/diffusion3d_fused3.hpp): checks it bit for bit
// against three launches of the one-iteration kernel (fields with random contents, the reference's two ping-pong buffers with their own
// boundary values), compares the three fused norms, and times it beside k_diff3_march2 on the same arrays.
// usage: diffusion_tune3 [nx] [ny] [nz] [iters] [zc] [xcd|dbg<<4] [lane_off]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../finalprojectrepo.jl_amd/csrc/diffusion3d_launch.hpp"
#include "../finalprojectrepo.jl_amd/csrc/diffusion3d_fused3.hpp"

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
__global__ void k_find_diff(const double* a, const double* b, size_t n, unsigned long long* cnt, unsigned long long* idx, int cap)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__double_as_longlong(a[i]) != __double_as_longlong(b[i])) {
            const unsigned long long k = atomicAdd(cnt, 1ull);
            if (k < (unsigned long long)cap) idx[k] = i;
        }
}

int main(int argc, char** argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 512, ny = argc > 2 ? atoi(argv[2]) : nx, nz = argc > 3 ? atoi(argv[3]) : nx;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const int zc = argc > 5 ? atoi(argv[5]) : 0, xcd = argc > 6 ? atoi(argv[6]) : 0, lane_off = argc > 7 ? atoi(argv[7]) : 1;
    const size_t N = (size_t)nx * ny * nz;
    int ncu = 256;
    { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
    double *Ht, *A, *B, *refA, *refB, *dHref, *X, *Y, *dH, *C, *parts;
    unsigned long long *cnt, *didx;
    for (double** p : {&Ht, &A, &B, &refA, &refB, &dHref, &X, &Y, &dH, &C}) CK(hipMalloc(p, N * 8));
    CK(hipMalloc(&parts, 4 * (1 << 20) * 8)); CK(hipMalloc(&cnt, 8)); CK(hipMalloc(&didx, 64 * 8));
    double *p1 = parts, *p2 = parts + (1 << 20), *p3 = parts + 2 * (1 << 20), *pr = parts + 3 * (1 << 20);
    k_fill_rand<<<2048, 256>>>(Ht, N, 1);
    k_fill_rand<<<2048, 256>>>(A, N, 2);
    k_fill_rand<<<2048, 256>>>(B, N, 3);      // the other buffer: only its boundary values matter
    hipStream_t s; CK(hipStreamCreate(&s));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(refA, A, N * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(refB, B, N * 8, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(X, A, N * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(Y, B, N * 8, hipMemcpyDeviceToDevice));
    CK(hipMemset(dHref, 0, N * 8)); CK(hipMemset(dH, 0, N * 8));
    const double dx = 10.0 / nx, dy = 10.0 / ny, dz = 10.0 / nz;
    const double dmin = fmin(dx, fmin(dy, dz));
    Diff3Args a;
    a.Ht = Ht; a.nx = nx; a.ny = ny; a.nz = nz;
    a.lo[0] = a.lo[1] = a.lo[2] = 1; a.hi[0] = nx - 1; a.hi[1] = ny - 1; a.hi[2] = nz - 1;
    a.dtau = dmin * dmin / 8.1; a._dt = 5.0; a._dx = 1 / dx; a._dy = 1 / dy; a._dz = 1 / dz; a.D_dx = 1 / dx; a.D_dy = 1 / dy; a.D_dz = 1 / dz;
    a.scale = 0.2; a.partials = pr;
    int np = 0;
    std::vector<double> hp(1 << 20);
    double ref[3];
    Diff3Tuning t;
    for (int it = 0; it < 3; ++it) {          // refA -> refB -> refA -> refB
        a.Htau = (it & 1) ? refB : refA; a.Htau2 = (it & 1) ? refA : refB; a.dHdtau = dHref;
        CK(diff3_launch(a, true, t, s, 1 << 20, &np)); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hp.data(), pr, (size_t)np * 8, hipMemcpyDeviceToHost));
        ref[it] = 0; for (int i = 0; i < np; ++i) ref[it] += hp[i];
    }
    Diff3Args3 f;
    memset(&f, 0, sizeof f);
    f.Ht = Ht; f.X = X; f.Bnd = Y; f.Y = Y; f.dH = dH; f.nx = nx; f.ny = ny; f.nz = nz;
    for (int d = 0; d < 3; ++d) { f.lo[d] = a.lo[d]; f.hi[d] = a.hi[d]; }
    f.dtau = a.dtau; f._dt = a._dt; f._dx = a._dx; f._dy = a._dy; f._dz = a._dz; f.D_dx = a.D_dx; f.D_dy = a.D_dy; f.D_dz = a.D_dz;
    f.scale = a.scale; f.partials1 = p1; f.partials2 = p2; f.partials3 = p3; f.skip = nullptr; f.lane_off = lane_off;
    long bal = 0;
    hipError_t e = diff3_launch3(f, true, zc, xcd, s, 1 << 20, &np, ncu, &bal);
    if (e != hipSuccess) { printf("march3 launch failed: %s\n", hipGetErrorString(e)); return 1; }
    CK(hipStreamSynchronize(s));
    double got[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        CK(hipMemcpy(hp.data(), k == 0 ? p1 : (k == 1 ? p2 : p3), (size_t)np * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < np; ++i) got[k] += hp[i];
    }
    bool ok = true;
    for (int arr = 0; arr < 2; ++arr) {
        CK(hipMemsetAsync(cnt, 0, 8, s));
        k_find_diff<<<2048, 256, 0, s>>>(arr ? dH : Y, arr ? dHref : refB, N, cnt, didx, 64);
        unsigned long long hidx[64], c2 = 0;
        CK(hipMemcpyAsync(&c2, cnt, 8, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(hidx, didx, 64 * 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("%s: %llu mismatches of %zu", arr ? "residual" : "field   ", c2, N);
        if (c2) {
            ok = false;
            printf("; some (x,y,z):");
            for (unsigned long long q = 0; q < (c2 < 24 ? c2 : 24); ++q)
                printf(" (%llu,%llu,%llu)", hidx[q] % nx, (hidx[q] / nx) % ny, hidx[q] / ((unsigned long long)nx * ny));
        }
        printf("\n");
    }
    for (int k = 0; k < 3; ++k) {
        const bool nb = fabs(got[k] - ref[k]) > 1e-12 * ref[k];
        printf("norm %d: %.17g vs %.17g %s\n", k + 1, got[k], ref[k], nb ? "MISMATCH" : "ok");
        ok = ok && !nb;
    }
    printf("march3 %dx%dx%d: units %d, bal %ld, %s\n", nx, ny, nz, np, bal, ok ? "BIT-EXACT" : "WRONG");
    if (iters <= 0) return ok ? 0 : 2;
    // ---- timing: march3 (X <-> Y ping-pong) beside march2 on the same arrays ----
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 32.0 * (double)(nx - 2) * (ny - 2) * (nz - 2);
    auto run3 = [&](int k) {
        for (int i = 0; i < k; ++i) {
            Diff3Args3 g = f;
            if (i & 1) { g.X = Y; g.Bnd = X; g.Y = X; }
            CK(diff3_launch3(g, true, zc, xcd, s, 1 << 20, &np, ncu));
        }
    };
    Diff3Args2 g2;
    memset(&g2, 0, sizeof g2);
    g2.Ht = Ht; g2.A = X; g2.B = B; g2.C = C; g2.dH = dH; g2.nx = nx; g2.ny = ny; g2.nz = nz;
    for (int d = 0; d < 3; ++d) { g2.lo[d] = a.lo[d]; g2.hi[d] = a.hi[d]; }
    g2.dtau = a.dtau; g2._dt = a._dt; g2._dx = a._dx; g2._dy = a._dy; g2._dz = a._dz; g2.D_dx = a.D_dx; g2.D_dy = a.D_dy; g2.D_dz = a.D_dz;
    g2.scale = a.scale; g2.partials1 = p1; g2.partials2 = p2; g2.lane_off = 1;
    auto run2 = [&](int k) {
        for (int i = 0; i < k; ++i) {
            Diff3Args2 g = g2;
            if (i & 1) { g.A = C; g.C = X; }
            CK(diff3_launch2(g, true, 0, 0, s, 1 << 20, &np, 0, ncu));
        }
    };
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 2; ++which) {
            if (which) run2(60); else run3(60);
            float best = 1e9f;
            for (int round = 0; round < 3; ++round) {
                CK(hipEventRecord(e0, s));
                if (which) run2(iters); else run3(iters);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
                best = ms < best ? ms : best;
            }
            const int depth = which ? 2 : 3;
            printf("%s: %.4f ms/launch = %.4f ms/iteration, %.0f GB/s A_eff, physical %.0f GB/s = %.3f of 8 TB/s\n", which ? "march2" : "march3", best, best / depth,
                   depth * bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9 / 8000.0);
            fflush(stdout);
        }
    return ok ? 0 : 2;
}
