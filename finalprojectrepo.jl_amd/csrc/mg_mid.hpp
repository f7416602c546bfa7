// mg_mid.hpp -- the three levels above the LDS-resident sub-hierarchy in two launches
// Part of multigrid2d.hip (included there, in this order: mg_march.hpp, mg_cg.hpp, mg_small.hpp, mg_cg_persistent.hpp,
// mg_mid.hpp); kernels only, the host side that launches them is in multigrid2d.hip.
#pragma once

// ================================================================================================
// k_mid_down / k_mid_up: the three levels right above the LDS-resident sub-hierarchy in TWO launches
// ================================================================================================
// Below the finest levels a V-cycle is bound by launches, not by bytes: rocprofv3 shows ~4.5 us from dispatch to completion
// for a kernel that does nothing, and the marching passes of a 513^2, 257^2 or 129^2 level take 7-11 us each for 1-3 us
// of work.  The pre-smoothing passes of three consecutive levels A > B > C (each: two sweeps from the ZERO initial guess
// every level below the top starts from, multigrid.jl:132, + residual + injection) become ONE launch, and so do their three
// post-smoothing passes (prolongation + correction + two sweeps each): a workgroup owns a tile of the coarsest output and
// recomputes, in LDS, the halo it needs on the finer levels (pre: 53^2 points of A for 32^2 owned; post: 36^2 for 32^2),
// so workgroups never communicate.  Same point arithmetic and the same prolongation order as the per-level kernels: all
// arrays a later pass reads (tmp = the pre-smoothed field and res_c = the right-hand side of every level, the solution of
// level A) are bit-identical; the solutions of B and C only ever exist in LDS.
struct MidReg { int x0, x1, y0, y1; };   // inclusive index ranges at one level
__device__ __forceinline__ MidReg mid_clip(int x0, int x1, int y0, int y1, int nx, int ny)
{
    MidReg r;
    r.x0 = x0 < 0 ? 0 : x0; r.x1 = x1 > nx - 1 ? nx - 1 : x1;
    r.y0 = y0 < 0 ? 0 : y0; r.y1 = y1 > ny - 1 ? ny - 1 : y1;
    return r;
}
__device__ __forceinline__ MidReg mid_grow(MidReg r, int k, int nx, int ny) { return mid_clip(r.x0 - k, r.x1 + k, r.y0 - k, r.y1 + k, nx, ny); }
__device__ __forceinline__ int mid_w(const MidReg& r) { return r.x1 - r.x0 + 1; }
__device__ __forceinline__ int mid_h(const MidReg& r) { return r.y1 - r.y0 + 1; }
__device__ __forceinline__ int mid_n(const MidReg& r) { return mid_w(r) * mid_h(r); }
__device__ __forceinline__ int mid_at(const MidReg& r, int i, int j) { return (i - r.x0) + mid_w(r) * (j - r.y0); }

struct MidLevel {
    const double* f;     // right-hand side of the level (down: level A only is read, B and C are produced)
    double* tmp;         // pre-smoothed field (down writes, up reads)
    double* fout;        // down: right-hand side of the NEXT coarser level (res_c of this level)
    int nx, ny;
    double C, _h2, fac;
};
struct MidArgs {
    MidLevel L[3];       // A, B, C
    int zfuse;           // k_mid_down: the two sweeps from the zero guess as one pass (option mg_zero_fuse)
    FprFinishArgs fin;   // k_mid_down: partials != null = the finish of the cycle before, done by one more row of workgroups
    int nxD, nyD;        // the level below C (top of the LDS-resident sub-hierarchy)
    double* uD;          // down: its zero initial guess is written; up: its solution is read
    double* uA;          // up: solution of level A
    int apply_BCs;
    const int* skip;
};

constexpr int MID_TD = 4;     // k_mid_down: tile of the level-D right-hand side per workgroup
constexpr int MID_TA = 32;    // k_mid_up: tile of the level-A solution per workgroup
constexpr int MID_NT_DOWN = 1024, MID_NT_UP = 512;

// out(p) = in(p) + fac*res(in)(p) on region `ro`, boundary points copied; `in` lives on region `ri` (ro grown by one,
// clipped), f on region `rf`
__device__ __forceinline__ void mid_sweep(const double* in, const MidReg& ri, const double* f, const MidReg& rf, double* out,
                                          const MidReg& ro, int nx, int ny, double C, double _h2, double fac, int nt)
{
    const int w = mid_w(ro), n = mid_n(ro);
    const float rw = 1.0f / (float)w;
    const int wi = mid_w(ri);
    for (int idx = threadIdx.x; idx < n; idx += nt) {
        const int jj = mgs_row(idx, rw), ii = idx - jj * w;
        const int i = ro.x0 + ii, j = ro.y0 + jj;
        const int q = mid_at(ri, i, j);
        const double uc = in[q];
        double v = uc;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const double r = ((((in[q + 1] + in[q - 1]) + in[q + wi]) + in[q - wi]) - C * uc) * _h2 - f[mid_at(rf, i, j)];
            v = uc + fac * r;
        }
        out[idx] = v;
    }
}

// A level below the top starts from the ZERO guess (multigrid.jl:132): its first sweep (:124) is a pointwise function of the right-hand
// side -- the literal arithmetic of the sweep on zeros, kept bit for bit --
__device__ __forceinline__ double mid_z1(double f, double C, double _h2, double fac)
{
    const double r = ((((0.0 + 0.0) + 0.0) + 0.0) - C * 0.0) * _h2 - f;
    return 0.0 + fac * r;
}
// -- so the SECOND sweep (:125) is taken straight from the right-hand side (on region rf, which holds ro grown by one, clipped): no pass,
// no buffer and no barrier for the first.  out on region ro; same expressions in the same order as sweep after sweep.
__device__ __forceinline__ void mid_sweep_z2(const double* f, const MidReg& rf, double* out, const MidReg& ro, int nx, int ny, double C,
                                             double _h2, double fac, int nt)
{
    const int w = mid_w(ro), n = mid_n(ro);
    const float rw = 1.0f / (float)w;
    const int wf = mid_w(rf);
    auto u1 = [&](int i, int j, int q) { return (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) ? mid_z1(f[q], C, _h2, fac) : 0.0; };
    for (int idx = threadIdx.x; idx < n; idx += nt) {
        const int jj = mgs_row(idx, rw), ii = idx - jj * w;
        const int i = ro.x0 + ii, j = ro.y0 + jj;
        const int q = mid_at(rf, i, j);
        double v = 0.0;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const double uc = mid_z1(f[q], C, _h2, fac);
            const double r = ((((u1(i + 1, j, q + 1) + u1(i - 1, j, q - 1)) + u1(i, j + 1, q + wf)) + u1(i, j - 1, q - wf)) - C * uc) * _h2 - f[q];
            v = uc + fac * r;
        }
        out[idx] = v;
    }
}

// prolong_bf with the coarse field in an LDS region (indices clamped into it; a clamped value is never used)
__device__ __forceinline__ double mid_prolong(const double* cc, const MidReg& rc, int i, int j, int nx, int ny, int nxc, int nyc)
{
    const bool in = i >= 1 && j >= 1 && i <= nx - 2 && j <= ny - 2;
    const int io = i & 1, jo = j & 1;
    int icl = i >> 1, jcl = j >> 1;
    int ich = (icl + 1 < nxc) ? icl + 1 : nxc - 1, jch = (jcl + 1 < nyc) ? jcl + 1 : nyc - 1;
    const double w = (io | jo) ? ((io & jo) ? 0.25 : 0.5) : 1.0;
    const bool sx0 = icl >= 1 && icl <= nxc - 2, sx1 = io && (icl + 1 <= nxc - 2);
    const bool sy0 = jcl >= 1 && jcl <= nyc - 2, sy1 = jo && (jcl + 1 <= nyc - 2);
    icl = icl < rc.x0 ? rc.x0 : (icl > rc.x1 ? rc.x1 : icl); ich = ich < rc.x0 ? rc.x0 : (ich > rc.x1 ? rc.x1 : ich);
    jcl = jcl < rc.y0 ? rc.y0 : (jcl > rc.y1 ? rc.y1 : jcl); jch = jch < rc.y0 ? rc.y0 : (jch > rc.y1 ? rc.y1 : jch);
    const double c00 = cc[mid_at(rc, icl, jcl)], c10 = cc[mid_at(rc, ich, jcl)];
    const double c01 = cc[mid_at(rc, icl, jch)], c11 = cc[mid_at(rc, ich, jch)];
    double v = 0.0;
    v = v + ((in && sx0 && sy0) ? w * c00 : 0.0);
    v = v + ((in && sx1 && sy0) ? w * c10 : 0.0);
    v = v + ((in && sx0 && sy1) ? w * c01 : 0.0);
    v = v + ((in && sx1 && sy1) ? w * c11 : 0.0);
    return v;
}

__global__ __launch_bounds__(MID_NT_DOWN) void k_mid_down(MidArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (a.fin.partials && blockIdx.y == gridDim.y - 1) {   // the finish of the cycle before rides along (fprx_cycle_finish_defer)
        if (blockIdx.x == 0) fpr_cycle_finish_body<true>(a.fin, sm);
        return;
    }
    if (a.skip && *a.skip) return;
    constexpr int NT = MID_NT_DOWN;
    const int tid = threadIdx.x;
    const int ntx = (a.nxD - 1) / MID_TD > 0 ? (a.nxD - 1) / MID_TD : 1, nty = (a.nyD - 1) / MID_TD > 0 ? (a.nyD - 1) / MID_TD : 1;
    const int bx = blockIdx.x, by = blockIdx.y;
    // owned tile of the level-D right-hand side (the last tile of a dimension takes the remainder)
    MidReg own;
    own.x0 = bx * MID_TD; own.x1 = (bx == ntx - 1) ? a.nxD - 1 : own.x0 + MID_TD - 1;
    own.y0 = by * MID_TD; own.y1 = (by == nty - 1) ? a.nyD - 1 : own.y0 + MID_TD - 1;
    // regions, coarse to fine: rt[l] = where the pre-smoothed field of level l is needed, rf[l] = rt[l] grown by one
    // (first sweep / right-hand side); the residual at coarse point c reads the fine field at 2c-1 .. 2c+1
    MidReg rt[3], rf[3], rn = own;   // rn = region of the next coarser right-hand side
    for (int l = 2; l >= 0; --l) {
        rt[l] = mid_clip(2 * rn.x0 - 1, 2 * rn.x1 + 1, 2 * rn.y0 - 1, 2 * rn.y1 + 1, a.L[l].nx, a.L[l].ny);
        rf[l] = mid_grow(rt[l], 1, a.L[l].nx, a.L[l].ny);
        rn = rf[l];
    }
    // LDS: F | U1 (on rf) | U2 (on rt) of the current level, then the next level's F behind them
    double* F = sm;
    {   // right-hand side of level A from memory
        const MidLevel& L = a.L[0];
        const int w = mid_w(rf[0]), n = mid_n(rf[0]);
        const float rw = 1.0f / (float)w;
        for (int idx = tid; idx < n; idx += NT) {
            const int jj = mgs_row(idx, rw), ii = idx - jj * w;
            F[idx] = L.f[(size_t)(rf[0].x0 + ii) + (size_t)L.nx * (rf[0].y0 + jj)];
        }
    }
    __syncthreads();
    int scale = 8;   // level-l index = scale * level-D index
    for (int l = 0; l < 3; ++l, scale >>= 1) {
        const MidLevel& L = a.L[l];
        const int nf = mid_n(rf[l]), ntm = mid_n(rt[l]);
        double* U1 = F + nf;
        double* U2 = U1 + nf;
        double* Fn = U2 + ntm;   // next level's right-hand side
        if (a.zfuse) {   // both sweeps from the zero initial guess in one pass over rt (mid_sweep_z2)
            mid_sweep_z2(F, rf[l], U2, rt[l], L.nx, L.ny, L.C, L._h2, L.fac, NT);
        } else {
        {   // first sweep from the zero initial guess (:124 with u = 0; the literal arithmetic on zeros, kept bit for bit)
            const int w = mid_w(rf[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < nf; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = rf[l].x0 + ii, j = rf[l].y0 + jj;
                double v = 0.0;
                if (i >= 1 && j >= 1 && i < L.nx - 1 && j < L.ny - 1) {
                    const double r = ((((0.0 + 0.0) + 0.0) + 0.0) - L.C * 0.0) * L._h2 - F[idx];
                    v = 0.0 + L.fac * r;
                }
                U1[idx] = v;
            }
        }
        __syncthreads();
        mid_sweep(U1, rf[l], F, rf[l], U2, rt[l], L.nx, L.ny, L.C, L._h2, L.fac, NT);   // :125
        }
        __syncthreads();
        {   // the owned part of the pre-smoothed field goes to memory (the post-smoothing pass reads it)
            MidReg o;
            o.x0 = scale * own.x0; o.x1 = (bx == ntx - 1) ? L.nx - 1 : scale * (own.x1 + 1) - 1;
            o.y0 = scale * own.y0; o.y1 = (by == nty - 1) ? L.ny - 1 : scale * (own.y1 + 1) - 1;
            const int w = mid_w(o), n = mid_n(o);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = o.x0 + ii, j = o.y0 + jj;
                L.tmp[(size_t)i + (size_t)L.nx * j] = U2[mid_at(rt[l], i, j)];
            }
        }
        {   // residual at the injected points = right-hand side of the next level (:128-131; Neumann columns :355-357)
            const MidReg rc = (l < 2) ? rf[l + 1] : own;
            const int nxc = 1 + (L.nx - 1) / 2, nyc = 1 + (L.ny - 1) / 2;
            const int half = scale >> 1;
            MidReg o;   // owned part of that right-hand side
            o.x0 = half * own.x0; o.x1 = (bx == ntx - 1) ? nxc - 1 : half * (own.x1 + 1) - 1;
            o.y0 = half * own.y0; o.y1 = (by == nty - 1) ? nyc - 1 : half * (own.y1 + 1) - 1;
            const int w = mid_w(rc), n = mid_n(rc), wt = mid_w(rt[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int ic = rc.x0 + ii, jc = rc.y0 + jj;
                int is = ic;
                if (a.apply_BCs) is = (ic == 0) ? 1 : (ic == nxc - 1 ? nxc - 2 : ic);
                double v = 0.0;
                if (is >= 1 && is <= nxc - 2 && jc >= 1 && jc <= nyc - 2) {
                    const int q = mid_at(rt[l], 2 * is, 2 * jc);
                    v = ((((U2[q + 1] + U2[q - 1]) + U2[q + wt]) + U2[q - wt]) - L.C * U2[q]) * L._h2 - F[mid_at(rf[l], 2 * is, 2 * jc)];
                }
                if (l < 2) Fn[idx] = v;
                if (ic >= o.x0 && ic <= o.x1 && jc >= o.y0 && jc <= o.y1) {
                    L.fout[(size_t)ic + (size_t)nxc * jc] = v;
                    if (l == 2) a.uD[(size_t)ic + (size_t)nxc * jc] = 0.0;   // zero initial guess of level D (:132)
                }
            }
        }
        __syncthreads();
        F = Fn;
    }
}

__global__ __launch_bounds__(MID_NT_UP) void k_mid_up(MidArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (a.skip && *a.skip) return;
    constexpr int NT = MID_NT_UP;
    const int tid = threadIdx.x;
    const MidLevel& LA = a.L[0];
    const int ntx = (LA.nx - 1) / MID_TA > 0 ? (LA.nx - 1) / MID_TA : 1, nty = (LA.ny - 1) / MID_TA > 0 ? (LA.ny - 1) / MID_TA : 1;
    const int bx = blockIdx.x, by = blockIdx.y;
    MidReg own;   // owned tile of the level-A solution
    own.x0 = bx * MID_TA; own.x1 = (bx == ntx - 1) ? LA.nx - 1 : own.x0 + MID_TA - 1;
    own.y0 = by * MID_TA; own.y1 = (by == nty - 1) ? LA.ny - 1 : own.y0 + MID_TA - 1;
    // regions, fine to coarse: r2[l] = where the solution of level l is needed, r1 = r2 grown by one (after the first
    // sweep), rc = r2 grown by two (the corrected pre-smoothed field); the prolongation onto rc reads coarse points i>>1, (i>>1)+1
    MidReg r2[3], r1[3], rc[3], rD;
    r2[0] = own;
    for (int l = 0; l < 3; ++l) {
        r1[l] = mid_grow(r2[l], 1, a.L[l].nx, a.L[l].ny);
        rc[l] = mid_grow(r2[l], 2, a.L[l].nx, a.L[l].ny);
        const int nxc = 1 + (a.L[l].nx - 1) / 2, nyc = 1 + (a.L[l].ny - 1) / 2;
        const MidReg rn = mid_clip(rc[l].x0 >> 1, (rc[l].x1 + 1) >> 1, rc[l].y0 >> 1, (rc[l].y1 + 1) >> 1, nxc, nyc);
        if (l < 2) r2[l + 1] = rn; else rD = rn;
    }
    // LDS layout: per level X (on rc) | F (on r1) | U1 (on r1) | U2 (on r2; level A writes to memory instead), then D's field
    double* X[3]; double* Fv[3]; double* U1[3]; double* U2[3];
    double* p = sm;
    for (int l = 0; l < 3; ++l) {
        X[l] = p; p += mid_n(rc[l]);
        Fv[l] = p; p += mid_n(r1[l]);
        U1[l] = p; p += mid_n(r1[l]);
        U2[l] = p; p += (l == 0) ? 0 : mid_n(r2[l]);
    }
    double* UD = p;
    // every load the launch needs is issued up front: the pre-smoothed fields, the right-hand sides, the solution of level D
    for (int l = 0; l < 3; ++l) {
        const MidLevel& L = a.L[l];
        {
            const int w = mid_w(rc[l]), n = mid_n(rc[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                X[l][idx] = L.tmp[(size_t)(rc[l].x0 + ii) + (size_t)L.nx * (rc[l].y0 + jj)];
            }
        }
        {
            const int w = mid_w(r1[l]), n = mid_n(r1[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                Fv[l][idx] = L.f[(size_t)(r1[l].x0 + ii) + (size_t)L.nx * (r1[l].y0 + jj)];
            }
        }
    }
    {
        const int w = mid_w(rD), n = mid_n(rD);
        const float rw = 1.0f / (float)w;
        for (int idx = tid; idx < n; idx += NT) {
            const int jj = mgs_row(idx, rw), ii = idx - jj * w;
            UD[idx] = a.uD[(size_t)(rD.x0 + ii) + (size_t)a.nxD * (rD.y0 + jj)];
        }
    }
    __syncthreads();
    const double* cc = UD;
    MidReg rcc = rD;
    for (int l = 2; l >= 0; --l) {
        const MidLevel& L = a.L[l];
        const int nxc = 1 + (L.nx - 1) / 2, nyc = 1 + (L.ny - 1) / 2;
        {   // prolongation + correction (:136-139) of the pre-smoothed field, in place
            const int w = mid_w(rc[l]), n = mid_n(rc[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = rc[l].x0 + ii, j = rc[l].y0 + jj;
                int is = i;
                if (a.apply_BCs) is = (i == 0) ? 1 : (i == L.nx - 1 ? L.nx - 2 : i);
                X[l][idx] = X[l][idx] - mid_prolong(cc, rcc, is, j, L.nx, L.ny, nxc, nyc);
            }
        }
        __syncthreads();
        mid_sweep(X[l], rc[l], Fv[l], r1[l], U1[l], r1[l], L.nx, L.ny, L.C, L._h2, L.fac, NT);   // :142
        __syncthreads();
        if (l > 0) {
            mid_sweep(U1[l], r1[l], Fv[l], r1[l], U2[l], r2[l], L.nx, L.ny, L.C, L._h2, L.fac, NT);   // :143
            __syncthreads();
            cc = U2[l];
            rcc = r2[l];
        } else {   // level A: the second sweep writes the owned tile of the solution to memory
            const int w = mid_w(own), n = mid_n(own), wi = mid_w(r1[0]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = own.x0 + ii, j = own.y0 + jj;
                const int q = mid_at(r1[0], i, j);
                const double uc = U1[0][q];
                double v = uc;
                if (i >= 1 && j >= 1 && i < L.nx - 1 && j < L.ny - 1) {
                    const double r = ((((U1[0][q + 1] + U1[0][q - 1]) + U1[0][q + wi]) + U1[0][q - wi]) - L.C * uc) * L._h2 - Fv[0][q];
                    v = uc + L.fac * r;
                }
                a.uA[(size_t)i + (size_t)L.nx * j] = v;
            }
        }
    }
}

