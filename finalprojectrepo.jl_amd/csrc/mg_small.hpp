// mg_small.hpp -- a whole sub-hierarchy of the V-cycle in one workgroup, resident in LDS
// Part of multigrid2d.hip (included there, in this order: mg_march.hpp, mg_cg.hpp, mg_small.hpp, mg_cg_persistent.hpp,
// mg_mid.hpp); kernels only, the host side that launches them is in multigrid2d.hip.
#pragma once

// ================================================================================================
// k_mg_small: a whole sub-hierarchy of the V-cycle in ONE workgroup, resident in LDS.
//
// Coarse multigrid levels are launch-latency bound (a 65^2 level costs ~8 dependent launches of ~2 us
// each for ~0.1 us of work, and the 5x5 Jacobi solve another ~130).  Once a level and everything below
// it fit in the CU's 160 KiB LDS (3 arrays per level: u, rhs, ping-pong partner; sum over levels
// <= 20 000 doubles, i.e. up to 65x65 or 129x33), one 1024-thread workgroup executes
// Vcycle_2DPoisson! (multigrid.jl:91-170) for that level and all coarser ones with __syncthreads()
// between the passes: pre-smoothing, residual+injection, the Jacobi coarse solve with its early exit
// (:147-159), prolongation+correction, post-smoothing.  Same pointwise arithmetic as the per-level
// kernels (bit-identical fields); norms are block-tree sums.
// ================================================================================================
struct MgSmallArgs {
    double* u;          // top level of the sub-hierarchy, global memory, in/out
    const double* rhs;  // its right-hand side
    int nx, ny, nlev;   // nlev = levels including the coarsest one
    double h, c, tol;
    int css, apply_BCs, want_norm;
    double* out_sumsq;  // want_norm: sum(res.^2) of the last post-smoothing sweep of the top level
    FprSolveState* state;
    const int* skip;    // cycles enqueued ahead: return at once if *skip (null = unconditional)
    int zfuse;          // 1: the two pre-smoothing sweeps of a level that starts from the zero guess (every level but the top) are one pass; 2: the top level too
    int row_solve;      // 1: coarsest grids with <= 16 interior points are solved inside one DPP row (option mg_small_row)
    long long* prof;    // diagnostic (option mg_small_prof): wall_clock64 stamps (100 MHz) of thread 0 at the section borders
};

#ifndef MGS_NT_THREADS
#define MGS_NT_THREADS 1024
#endif
constexpr int MGS_NT = MGS_NT_THREADS;   // threads of the workgroup (A/B builds: -DMGS_NT_THREADS=512 / 256)
constexpr int MGS_RED = 32;  // doubles reserved for reductions / broadcasts
constexpr int MGS_WPL = 5;   // single-wave coarse solve: points per lane (up to 320 points: 17 x 17)

__device__ __forceinline__ double mgs_block_sum(double v, double* red)
{
    // all MGS_NT threads call; returns the total in every thread (wave sums by DPP, then every thread adds the 16 wave
    // totals in wave order from LDS: two barriers)
    v = fpr_wave_sum_all(v);
    const int tid = threadIdx.x;
    __syncthreads();  // protect red[] from the previous use
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < MGS_NT / 64; ++w) s += red[w];
    return s;
}

// idx -> row j = idx / nx without an integer division: floor((idx + 0.5) * (1/nx)) in float is exact while
// nx*ny*2.4e-7 < 0.5 (the rounding error of the product stays below the distance 0.5/nx of (idx + 0.5)/nx from an
// integer); the LDS arena holds 20000 doubles, so N < 2^15 here.
__device__ __forceinline__ int mgs_row(int idx, float rnx) { return (int)(((float)idx + 0.5f) * rnx); }

// uout = uin + fac*res(uin) on the interior, boundary copied; returns this thread's sum of res^2
__device__ __forceinline__ double mgs_sweep(const double* uin, const double* f, double* uout, int nx, int ny, double C,
                                            double _h2, double fac)
{
    double acc = 0.0;
    const int N = nx * ny;
    const float rnx = 1.0f / (float)nx;
    for (int idx = threadIdx.x; idx < N; idx += MGS_NT) {
        const int j = mgs_row(idx, rnx), i = idx - j * nx;
        const double uc = uin[idx];
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const double r = ((((uin[idx + 1] + uin[idx - 1]) + uin[idx + nx]) + uin[idx - nx]) - C * uc) * _h2 - f[idx];
            uout[idx] = uc + fac * r;
            acc += r * r;
        } else {
            uout[idx] = uc;
        }
    }
    return acc;
}

// the two sweeps from the ZERO guess (multigrid.jl:124-125 after :132) in one pass: the first is a pointwise function of f (the literal
// arithmetic of the sweep on zeros), so the second is taken straight from f -- same expressions in the same order
__device__ __forceinline__ double mgs_z1(double f, double C, double _h2, double fac)
{
    const double r = ((((0.0 + 0.0) + 0.0) + 0.0) - C * 0.0) * _h2 - f;
    return 0.0 + fac * r;
}
__device__ __forceinline__ void mgs_sweep_z2(const double* f, double* uout, int nx, int ny, double C, double _h2, double fac)
{
    const int N = nx * ny;
    const float rnx = 1.0f / (float)nx;
    auto u1 = [&](int i, int j, int q) { return (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) ? mgs_z1(f[q], C, _h2, fac) : 0.0; };
    for (int idx = threadIdx.x; idx < N; idx += MGS_NT) {
        const int j = mgs_row(idx, rnx), i = idx - j * nx;
        double v = 0.0;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const double uc = mgs_z1(f[idx], C, _h2, fac);
            const double r = ((((u1(i + 1, j, idx + 1) + u1(i - 1, j, idx - 1)) + u1(i, j + 1, idx + nx)) + u1(i, j - 1, idx - nx)) - C * uc) * _h2 - f[idx];
            v = uc + fac * r;
        }
        uout[idx] = v;
    }
}

__global__ __launch_bounds__(MGS_NT) void k_mg_small(MgSmallArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (a.skip && *a.skip) return;
    double* red = sm;
    double* arena = sm + MGS_RED;
    const int tid = threadIdx.x;
    int pslot = 0;
    auto stamp = [&]() { if (a.prof && tid == 0) a.prof[pslot++] = wall_clock64(); };
    stamp();   // 0: start

    // level d: dims ((nx-1)>>d)+1, arrays u|f|t at arena + off(d)
    auto lnx = [&](int d) { return ((a.nx - 1) >> d) + 1; };
    auto lny = [&](int d) { return ((a.ny - 1) >> d) + 1; };
    auto off = [&](int d) {
        int o = 0;
        for (int e = 0; e < d; ++e) o += 3 * lnx(e) * lny(e);
        return o;
    };
    auto hlev = [&](int d) {
        double h = a.h;
        for (int e = 0; e < d; ++e) h = h * 2;  // the recursion passes h*2 (multigrid.jl:133)
        return h;
    };

    {   // load the top level
        const int N = a.nx * a.ny;
        double* U = arena;
        double* F = arena + N;
        if (a.zfuse == 2 && a.nlev > 1) {   // u is the zero guess: only the right-hand side is loaded
            for (int idx = tid; idx < N; idx += MGS_NT) F[idx] = a.rhs[idx];
        } else
        for (int idx = tid; idx < N; idx += MGS_NT) {
            U[idx] = a.u[idx];
            F[idx] = a.rhs[idx];
        }
    }
    __syncthreads();
    stamp();   // 1: top level loaded

    const int c_lev = a.nlev - 1;
    // ---- down sweep ----
    for (int d = 0; d < c_lev; ++d) {
        const int nx = lnx(d), ny = lny(d), N = nx * ny;
        double* U = arena + off(d);
        double* F = U + N;
        double* T = F + N;
        const double h = hlev(d);
        const double C = 4.0 + a.c * (h * h), _h2 = 1 / (h * h);
        const double fac = (4.0 / 5.0) * ((h * h) / (4.0 + a.c * (h * h)));
        if (a.zfuse == 2 || (a.zfuse == 1 && d > 0)) {
            mgs_sweep_z2(F, U, nx, ny, C, _h2, fac);   // :124-125 from the zero guess
        } else {
            mgs_sweep(U, F, T, nx, ny, C, _h2, fac);  // :124
            __syncthreads();
            mgs_sweep(T, F, U, nx, ny, C, _h2, fac);  // :125
        }
        __syncthreads();
        // residual + injection + Neumann rows into the next level's rhs; next level's u = 0 (:128-132)
        const int nxc = lnx(d + 1), nyc = lny(d + 1), Nc = nxc * nyc;
        double* Uc = arena + off(d + 1);
        double* Fc = Uc + Nc;
        const float rnxc = 1.0f / (float)nxc;
        for (int idx = tid; idx < Nc; idx += MGS_NT) {
            const int jc = mgs_row(idx, rnxc), ic = idx - jc * nxc;
            int is = ic;
            if (a.apply_BCs) is = (ic == 0) ? 1 : (ic == nxc - 1 ? nxc - 2 : ic);
            double v = 0.0;
            if (is >= 1 && is <= nxc - 2 && jc >= 1 && jc <= nyc - 2) {
                const int id = 2 * is + nx * (2 * jc);
                v = ((((U[id + 1] + U[id - 1]) + U[id + nx]) + U[id - nx]) - C * U[id]) * _h2 - F[id];
            }
            Fc[idx] = v;
            Uc[idx] = 0.0;
        }
        __syncthreads();
        stamp();   // 2 .. 1+c_lev: level d went down
    }

    // ---- coarsest level: Jacobi with early exit (:147-159) ----
    double* ucur;  // where the coarse solution ends up
    {
        const int nx = lnx(c_lev), ny = lny(c_lev), N = nx * ny;
        double* U = arena + off(c_lev);
        double* F = U + N;
        double* T = F + N;
        const double h = hlev(c_lev);
        const double C = 4.0 + a.c * (h * h), _h2 = 1 / (h * h);
        const double fac = (4.0 / 5.0) * ((h * h) / (4.0 + a.c * (h * h)));
        double acc = 0.0;
        for (int idx = tid; idx < N; idx += MGS_NT) acc += F[idx] * F[idx];
        const double tol_rhs = a.tol * sqrt(mgs_block_sum(acc, red) / (double)N);  // :150
        const int iters = 20 * a.css;
        double res_rms = 0.0;
        int it = 0;
        double* pin = U;
        double* pout = T;
        const int nxi = nx - 2, nyi = ny - 2, ni = nxi * nyi;
        if (N <= 64 && ni >= 1 && ni <= 16 && a.row_solve) {
            // coarsest grid with at most 16 INTERIOR points (5x5 -> 3x3, the default coarse_solve_size): the interior lives in
            // the first lanes of wave 0, one point per lane (lane = (i-1) + nxi*(j-1)), all of it inside one 16-lane DPP row:
            // neighbours by row shifts (a boundary neighbour is a per-lane constant), the norm by four rotate-and-add steps
            // (lane 0's order of summation; uniform through readfirstlane).  sqrt and the division of :157 are only evaluated
            // when the exit test can possibly hold: sum > N * tol_rhs^2 * (1 + 1e-10) implies sqrt(sum/N) > tol_rhs.
            if (tid < 64) {
                const int lane = tid;
                const bool in = lane < ni;
                const int jj = in ? lane / nxi : 0, ii = in ? lane - jj * nxi : 0;   // interior coordinates, once
                const int g = (ii + 1) + nx * (jj + 1);
                const double fv = in ? F[g] : 0.0;
                double uu = in ? U[g] : 0.0;
                // boundary neighbours are constants (Dirichlet copy); interior ones come from the row shifts
                const bool iE = in && ii + 1 < nxi, iW = in && ii > 0, iN = in && jj + 1 < nyi, iS = in && jj > 0;
                const double cE = (in && !iE) ? U[g + 1] : 0.0, cW = (in && !iW) ? U[g - 1] : 0.0;
                const double cN = (in && !iN) ? U[g + nx] : 0.0, cS = (in && !iS) ? U[g - nx] : 0.0;
                const double hi_thr = ((double)N * (tol_rhs * tol_rhs)) * (1.0 + 1e-10);
                double sq_last = 0.0;
                bool have_rms = false;
                // the DPP control is an immediate: the loop is instantiated per interior width (dispatch ONCE, outside it)
                auto run = [&](auto NXIc) {
                    constexpr int NXI = decltype(NXIc)::value;
                    // The exit test of sweep k is evaluated while sweep k+1 is already in flight (its update is dropped if the
                    // test holds): the dependent chain of a sweep is then its stencil alone, not stencil + reduction + test.
                    double sq_vec = 0.0;     // per-lane total of sweep k-1 (every lane of row 0 holds a full sum)
                    bool done = false;
                    auto test = [&](int ksweep) {   // :157-158 for sweep `ksweep`, whose sum sits in sq_vec
                        const double sq = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sq_vec)),
                                                           __builtin_amdgcn_readfirstlane(__double2loint(sq_vec)));
                        it = ksweep;
                        sq_last = sq;
                        have_rms = false;
                        if (sq > hi_thr) return false;      // cannot have converged
                        res_rms = sqrt(sq / (double)N);     // :157
                        have_rms = true;
                        return res_rms < tol_rhs;
                    };
                    for (int k = 1; k <= iters; ++k) {
                        const double sE = fpr_dpp<0x101>(uu), sW = fpr_dpp<0x111>(uu);                 // lanes i+1, i-1
                        const double sN = fpr_dpp<0x100 + NXI>(uu), sS = fpr_dpp<0x110 + NXI>(uu);     // lanes i+nxi, i-nxi
                        const double E = iE ? sE : cE, W = iW ? sW : cW, Nn = iN ? sN : cN, Ss = iS ? sS : cS;
                        const double r = ((((E + W) + Nn) + Ss) - C * uu) * _h2 - fv;
                        const double uu_new = in ? uu + fac * r : uu;
                        double sq = in ? r * r : 0.0;
                        sq += fpr_row_ror<8>(sq);
                        sq += fpr_row_ror<4>(sq);
                        sq += fpr_row_ror<2>(sq);
                        sq += fpr_row_ror<1>(sq);
                        if (k > 1 && test(k - 1)) { done = true; break; }   // uu is still the field after sweep k-1
                        uu = uu_new;
                        sq_vec = sq;
                    }
                    if (!done) test(iters);
                };
#define FPR_ROW_CASE(n) case n: run(std::integral_constant<int, n>{}); break;
                switch (nxi) {
                    FPR_ROW_CASE(1) FPR_ROW_CASE(2) FPR_ROW_CASE(3) FPR_ROW_CASE(4) FPR_ROW_CASE(5) FPR_ROW_CASE(6) FPR_ROW_CASE(7)
                    FPR_ROW_CASE(8) FPR_ROW_CASE(9) FPR_ROW_CASE(10) FPR_ROW_CASE(11) FPR_ROW_CASE(12) FPR_ROW_CASE(13)
                    FPR_ROW_CASE(14) FPR_ROW_CASE(15)
                default: run(std::integral_constant<int, 15>{}); break;   // nxi = 16: one row of points, iN = iS = false everywhere
                }
#undef FPR_ROW_CASE
                if (!have_rms) res_rms = sqrt(sq_last / (double)N);
                if (in) U[g] = uu;
                if (tid == 0) { red[MGS_RED - 1] = res_rms; red[MGS_RED - 2] = (double)it; }
            }
            __syncthreads();
            res_rms = red[MGS_RED - 1];
            it = (int)red[MGS_RED - 2];
        } else
        if (N <= 64) {
            // tiny coarsest grid (5x5, 9x5, ...): one point per lane of wave 0, all in registers --
            // neighbours by wavefront shuffles, norm by a butterfly (every lane gets the same bits), no barrier
            if (tid < 64) {
                const int lane = tid;
                const bool in = lane < N;
                const int j = lane / nx, i = lane - j * nx;
                const bool inter = in && i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1;
                double uu = in ? U[lane] : 0.0;
                const double fv = in ? F[lane] : 0.0;
                for (int k = 1; k <= iters; ++k) {
                    const double E = __shfl(uu, lane + 1, 64), W = __shfl(uu, lane - 1, 64);
                    const double Nn = __shfl(uu, lane + nx, 64), Ss = __shfl(uu, lane - nx, 64);
                    const double r = ((((E + W) + Nn) + Ss) - C * uu) * _h2 - fv;
                    double sq = inter ? r * r : 0.0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
                    res_rms = sqrt(sq / (double)N);
                    if (inter) uu = uu + fac * r;
                    it = k;
                    if (res_rms < tol_rhs) break;
                }
                if (in) U[lane] = uu;
                if (tid == 0) { red[MGS_RED - 1] = res_rms; red[MGS_RED - 2] = (double)it; }
            }
            __syncthreads();
            res_rms = red[MGS_RED - 1];
            it = (int)red[MGS_RED - 2];
        } else if (N <= 64 * MGS_WPL && a.row_solve) {
            // small coarsest grid (9x9, 17x17, 17x9, ...): ONE wave sweeps it, up to MGS_WPL points per lane, the field
            // ping-ponging between U and T in LDS.  The LDS operations of a wave execute in order and the wave sum is in
            // registers, so a sweep needs no workgroup barrier (the block form: three per sweep, 16 waves).  sqrt and the
            // division of :157 only when the exit test can possibly hold (see the DPP-row form above).
            if (tid < 64) {
                const int lane = tid;
                int idx[MGS_WPL];
                bool inter[MGS_WPL];
                double fv[MGS_WPL];
                const float rnx = 1.0f / (float)nx;
#pragma unroll
                for (int m = 0; m < MGS_WPL; ++m) {
                    const int id = lane + 64 * m;
                    const bool in = id < N;
                    const int j = in ? mgs_row(id, rnx) : 0, i = id - j * nx;
                    idx[m] = in ? id : -1;
                    inter[m] = in && i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1;
                    fv[m] = in ? F[id] : 0.0;
                }
                const double hi_thr = ((double)N * (tol_rhs * tol_rhs)) * (1.0 + 1e-10);
                double sq_last = 0.0;
                bool have_rms = false;
                const double* pi = U;
                double* po = T;
                for (int k = 1; k <= iters; ++k) {
                    double sq = 0.0;
#pragma unroll
                    for (int m = 0; m < MGS_WPL; ++m) {
                        if (idx[m] >= 0) {
                            const int id = idx[m];
                            const double uc = pi[id];
                            double v = uc;
                            if (inter[m]) {
                                const double r = ((((pi[id + 1] + pi[id - 1]) + pi[id + nx]) + pi[id - nx]) - C * uc) * _h2 - fv[m];
                                v = uc + fac * r;
                                sq += r * r;
                            }
                            po[id] = v;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this sweep's LDS traffic is done before the next one starts
                    const double ssum = fpr_wave_sum_all(sq);
                    const double* t = pi; pi = po; po = const_cast<double*>(t);
                    it = k;
                    sq_last = ssum;
                    have_rms = false;
                    if (ssum > hi_thr) continue;        // cannot have converged
                    res_rms = sqrt(ssum / (double)N);   // :157
                    have_rms = true;
                    if (res_rms < tol_rhs) break;
                }
                if (!have_rms) res_rms = sqrt(sq_last / (double)N);
                if (tid == 0) { red[MGS_RED - 1] = res_rms; red[MGS_RED - 2] = (double)it; }
            }
            __syncthreads();
            res_rms = red[MGS_RED - 1];
            it = (int)red[MGS_RED - 2];
            pin = (it & 1) ? T : U;    // the solution sits in T after an odd number of sweeps
        } else
        for (int i = 1; i <= iters; ++i) {
            const double s = mgs_block_sum(mgs_sweep(pin, F, pout, nx, ny, C, _h2, fac), red);  // syncs inside
            res_rms = sqrt(s / (double)N);
            double* t = pin; pin = pout; pout = t;
            it = i;
            if (res_rms < tol_rhs) break;  // uniform: every thread holds the same value
        }
        ucur = pin;
        if (tid == 0) {
            a.state->acc_iters += it;
            a.state->last_rms = res_rms;
        }
        __syncthreads();
        stamp();   // coarsest level solved
    }

    // ---- up sweep ----
    for (int d = c_lev - 1; d >= 0; --d) {
        const int nx = lnx(d), ny = lny(d), N = nx * ny;
        double* U = arena + off(d);
        double* F = U + N;
        double* T = F + N;
        const double h = hlev(d);
        const double C = 4.0 + a.c * (h * h), _h2 = 1 / (h * h);
        const double fac = (4.0 / 5.0) * ((h * h) / (4.0 + a.c * (h * h)));
        const int nxc = lnx(d + 1);
        const double* Uc = (d + 1 == c_lev) ? ucur : arena + off(d + 1);
        const float rnx = 1.0f / (float)nx;
        for (int idx = tid; idx < N; idx += MGS_NT) {  // prolongation + correction (:136-139)
            const int j = mgs_row(idx, rnx), i = idx - j * nx;
            int is = i;
            if (a.apply_BCs) is = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i);
            U[idx] = U[idx] - prolong_bf(Uc, is, j, nx, ny, nxc, lny(d + 1));   // branch-free form of prolong_at: same value
        }
        __syncthreads();
        mgs_sweep(U, F, T, nx, ny, C, _h2, fac);  // :142
        __syncthreads();
        const double acc = mgs_sweep(T, F, U, nx, ny, C, _h2, fac);  // :143
        if (d == 0 && a.want_norm) {
            const double s = mgs_block_sum(acc, red);
            if (tid == 0) a.out_sumsq[0] = s;
        }
        __syncthreads();
        stamp();   // level d came up
    }

    {   // store the top level's solution
        const int N = a.nx * a.ny;
        const double* U = (c_lev == 0) ? ucur : arena;
        for (int idx = tid; idx < N; idx += MGS_NT) a.u[idx] = U[idx];
    }
    stamp();   // stored
}

