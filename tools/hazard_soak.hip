// hazard_soak.hip -- soak test of the fused two-iteration diffusion kernel's store-data hazard workaround.
// Builds with -DDIFF3_STORE_NOP=<n> (wait states after the 16-byte stores of a row; -1 = none).  Runs `launches`
// fused launches at nx x ny x nz and compares EVERY cell of both outputs, bit for bit, with two single-iteration
// launches computed once; a second stream keeps copying a 1 GiB buffer meanwhile (whatever of it the fused kernel's
// one-workgroup-per-CU grid lets in).  usage: hazard_soak nx ny nz launches [reserve_cus: the reserved form of the launch, 0 = plain grid]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../finalprojectrepo.jl_amd/csrc/diffusion3d_launch.hpp"
#include "../finalprojectrepo.jl_amd/csrc/diffusion3d_fused2.hpp"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                            \
        }                                                                                       \
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
__global__ void k_copy16(double2* dst, const double2* src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 512, ny = argc > 2 ? atoi(argv[2]) : nx, nz = argc > 3 ? atoi(argv[3]) : nx;
    const int launches = argc > 4 ? atoi(argv[4]) : 200;
    const size_t N = (size_t)nx * ny * nz;
    double *Ht, *A, *B, *C, *dH, *Bref, *Cref, *dHref, *parts, *bg0, *bg1;
    unsigned long long* cnt;
    for (double** p : {&Ht, &A, &B, &C, &dH, &Bref, &Cref, &dHref}) CK(hipMalloc(p, N * 8));
    const size_t NB = (size_t)1 << 27;   // 1 GiB of doubles
    CK(hipMalloc(&bg0, NB * 8)); CK(hipMalloc(&bg1, NB * 8));
    CK(hipMalloc(&parts, (1 << 22) * 8)); CK(hipMalloc(&cnt, 8));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    k_fill_rand<<<2048, 256, 0, s>>>(Ht, N, 1);
    k_fill_rand<<<2048, 256, 0, s>>>(A, N, 2);
    k_fill_rand<<<2048, 256, 0, s>>>(B, N, 3);
    k_fill_rand<<<2048, 256, 0, s>>>(bg0, NB, 4);
    CK(hipMemcpyAsync(Bref, B, N * 8, hipMemcpyDeviceToDevice, s));
    CK(hipMemcpyAsync(Cref, A, N * 8, hipMemcpyDeviceToDevice, s));   // the output carries A's boundary
    CK(hipMemcpyAsync(C, A, N * 8, hipMemcpyDeviceToDevice, s));
    CK(hipMemsetAsync(dHref, 0, N * 8, s)); CK(hipMemsetAsync(dH, 0, N * 8, s));
    const double dx = 10.0 / nx;
    int np = 0;
    {   // reference: two single-iteration launches, A -> Bref (keeps B's boundary), Bref -> Cref
        Diff3Args a;
        a.Ht = Ht; a.nx = nx; a.ny = ny; a.nz = nz;
        const int n3[3] = {nx, ny, nz};
        for (int d = 0; d < 3; ++d) { a.lo[d] = 1; a.hi[d] = n3[d] - 1; }
        a.dtau = dx * dx / 8.1; a._dt = 5.0; a._dx = a._dy = a._dz = 1 / dx; a.D_dx = a.D_dy = a.D_dz = 1 / dx;
        a.scale = 0.2; a.partials = parts;
        Diff3Tuning t;
        a.Htau = A; a.Htau2 = Bref; a.dHdtau = dHref;
        CK(diff3_launch(a, false, t, s, 1 << 22, &np));
        a.Htau = Bref; a.Htau2 = Cref;
        CK(diff3_launch(a, false, t, s, 1 << 22, &np));
        CK(hipStreamSynchronize(s));
    }
    const int reserve = argc > 5 ? atoi(argv[5]) : 0;
    int ncu = 256;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    Diff3Args2 f;
    f.skip = nullptr;
    f.lane_off = 1;
    f.Ht = Ht; f.A = A; f.B = B; f.C = C; f.dH = dH;
    f.nx = nx; f.ny = ny; f.nz = nz;
    {
        const int n3[3] = {nx, ny, nz};
        for (int d = 0; d < 3; ++d) { f.lo[d] = 1; f.hi[d] = n3[d] - 1; }
    }
    f.dtau = dx * dx / 8.1; f._dt = 5.0; f._dx = f._dy = f._dz = 1 / dx; f.D_dx = f.D_dy = f.D_dz = 1 / dx; f.scale = 0.2;
    f.partials1 = parts; f.partials2 = parts + (1 << 21);
    if (!diff3_can_fuse2(Ht, A, B, C, dH, nx, ny, nz)) { printf("size not supported by the fused kernel\n"); return 2; }
    CK(hipMemsetAsync(cnt, 0, 8, s));
    unsigned long long bad_launches = 0, prev = 0;
    for (int i = 0; i < launches; ++i) {
        for (int k = 0; k < 3; ++k) k_copy16<<<1024, 256, 0, s2>>>((double2*)(k & 1 ? bg0 : bg1), (const double2*)(k & 1 ? bg1 : bg0), NB / 2);
        CK(diff3_launch2(f, true, 0, 0, s, 1 << 21, &np, 0, ncu, 0, 0, reserve));
        k_count_diff<<<2048, 256, 0, s>>>(C, Cref, N, cnt);
        k_count_diff<<<2048, 256, 0, s>>>(dH, dHref, N, cnt);
        if ((i & 15) == 15 || i == launches - 1) {
            unsigned long long bad = 0;
            CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (bad != prev) ++bad_launches;
            prev = bad;
        }
    }
    CK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost));
    printf("DIFF3_STORE_NOP=%d  %dx%dx%d  %d fused launches (%d workgroups each): %llu mismatching values of %.3g compared  %s\n",
           (int)DIFF3_STORE_NOP, nx, ny, nz, launches, np, bad, 2.0 * (double)N * launches, bad ? "FAIL" : "ok");
    return bad ? 3 : 0;
}
