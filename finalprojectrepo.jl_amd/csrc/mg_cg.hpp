// mg_cg.hpp -- kernels of cg! (krylov.jl:55-91): one per operation, three and two dependent launches per iteration
// Part of multigrid2d.hip (included there, in this order: mg_march.hpp, mg_cg.hpp, mg_small.hpp, mg_cg_persistent.hpp,
// mg_mid.hpp); kernels only, the host side that launches them is in multigrid2d.hip.
#pragma once

// ---- CG kernels (krylov.jl:55-91) -------------------------------------------------------------------
// The three dot products of an iteration (:64, :69, :83) are Dot2 sums (fpr_internal.hpp: twofold precision, rounded once): block
// partials travel as (s, e) pairs, every consumer rounds s + e.  All launch forms and the oracle (orc_cg2d) then hold the same
// alpha / beta / exit test in every iteration, whatever their summation order.
__global__ __launch_bounds__(256) void k_cg_init(const double* __restrict__ b, double* __restrict__ r, double* __restrict__ p,
                                                  double* __restrict__ ph, double* __restrict__ x, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = b[i];
        r[i] = v; p[i] = v; ph[i] = v; x[i] = 0.0;
    }
}

// p_hat = A p on the interior (boundary of p_hat keeps b's values), partial sums of p .* p_hat over
// the WHOLE array (krylov.jl:68-69)
__global__ __launch_bounds__(256) void k_cg_matvec_dot(const double* __restrict__ p, double* __restrict__ ph, int nx, int ny,
                                                        double hx2, double hy2, double c, double* __restrict__ partials,
                                                        const FprSolveState* __restrict__ st)
{
    __shared__ double red[32];
    if (st->done) return;
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    double s = 0.0, e = 0.0;
    if (i < nx && j < ny) {
        const size_t id = (size_t)i + (size_t)nx * j;
        double q;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            q = lap_at(p, id, nx, hx2, hy2, c);
            ph[id] = q;
        } else {
            q = ph[id];
        }
        fpr_s2_add_prod(s, e, p[id], q);
    }
    fpr_block_sum_s2<256>(s, e, red);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        partials[2 * b] = s; partials[2 * b + 1] = e;
    }
}

__global__ __launch_bounds__(256) void k_cg_alpha(FprSolveState* st, const double* __restrict__ partials, int nparts)
{
    __shared__ double red[32];
    if (st->done) return;
    double ss, se;
    fpr_sum_partials_256_s2(partials, nparts, red, ss, se);
    if (threadIdx.x == 0) {
        const double s = ss + se;
        st->pq = s;
        st->alpha = st->rho / s;  // krylov.jl:69
    }
}

// x .+= alpha p ; r .-= alpha p_hat ; partial sums of r.^2   (krylov.jl:70-72)
__global__ __launch_bounds__(256) void k_cg_update(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                    const double* __restrict__ ph, size_t n, double* __restrict__ partials,
                                                    const FprSolveState* __restrict__ st)
{
    __shared__ double red[32];
    if (st->done) return;
    const double alpha = st->alpha;
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0.0, e = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        x[i] = x[i] + alpha * p[i];
        const double rn = r[i] - alpha * ph[i];
        r[i] = rn;
        fpr_s2_add_prod(s, e, rn, rn);
    }
    fpr_block_sum_s2<256>(s, e, red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s; partials[2 * blockIdx.x + 1] = e; }
}

__global__ __launch_bounds__(256) void k_cg_check(FprSolveState* st, const double* __restrict__ partials, int nparts, double N)
{
    __shared__ double red[32];
    if (st->done) return;
    double ss, se;
    fpr_sum_partials_256_s2(partials, nparts, red, ss, se);
    if (threadIdx.x == 0) {
        const double s = ss + se;
        const double normr = sqrt(s);
        st->iters += 1;
        st->last_rms = sqrt(s / N);  // krylov.jl:90
        if (normr < st->thresh) {
            st->done = 1;  // krylov.jl:76-81
        } else {
            st->rho_old = st->rho;
            st->rho = s;                      // krylov.jl:83
            st->beta = st->rho / st->rho_old; // krylov.jl:84
        }
    }
}

// ---- fused CG iteration: 3 dependent launches instead of 5 ---------------------------------------------
// k_cg_matvec_dot -> k_cg_update_f (every workgroup derives alpha from the dot partials) ->
// k_cg_p_f (every workgroup derives ||r||, the exit test and beta from the r.r partials).
// rho is double-buffered by iteration parity so that workgroup 0 can publish the new value while the
// others still read the old one.  `it` = 0-based iteration index.
__global__ __launch_bounds__(256) void k_cg_update_f(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                      const double* __restrict__ ph, size_t n, const double* __restrict__ pq_partials,
                                                      int npq, double* __restrict__ partials, FprSolveState* __restrict__ st, int it)
{
    __shared__ double red[32];
    __shared__ double s_alpha;
    if (st->done) return;
    // first element of this thread's grid-stride sequence: loaded before alpha is known (the loads overlap the reduction
    // of the dot-product partials; coarse grids have at most one element per thread)
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    double x0 = 0.0, p0 = 0.0, r0 = 0.0, q0 = 0.0;
    if (i0 < n) { x0 = x[i0]; p0 = p[i0]; r0 = r[i0]; q0 = ph[i0]; }
    double pqs, pqe;
    fpr_sum_partials_256_s2(pq_partials, npq, red, pqs, pqe);
    if (threadIdx.x == 0) {
        const double pq = pqs + pqe;
        const double alpha = st->rho2[it & 1] / pq;  // krylov.jl:69
        s_alpha = alpha;
        if (blockIdx.x == 0) { st->pq = pq; st->alpha = alpha; }
    }
    __syncthreads();
    const double alpha = s_alpha;
    double s = 0.0, e = 0.0;
    if (i0 < n) {
        x[i0] = x0 + alpha * p0;
        const double rn = r0 - alpha * q0;
        r[i0] = rn;
        fpr_s2_add_prod(s, e, rn, rn);
    }
    for (size_t i = i0 + stride; i < n; i += stride) {
        x[i] = x[i] + alpha * p[i];
        const double rn = r[i] - alpha * ph[i];
        r[i] = rn;
        fpr_s2_add_prod(s, e, rn, rn);
    }
    __syncthreads();
    fpr_block_sum_s2<256>(s, e, red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s; partials[2 * blockIdx.x + 1] = e; }
}

__global__ __launch_bounds__(256) void k_cg_p_f(double* __restrict__ p, const double* __restrict__ r, size_t n,
                                                 const double* __restrict__ rr_partials, int nrr, FprSolveState* __restrict__ st,
                                                 int it, double N)
{
    __shared__ double red[32];
    __shared__ double s_beta;
    __shared__ int s_conv;
    if (st->done) return;
    double rrs, rre;
    fpr_sum_partials_256_s2(rr_partials, nrr, red, rrs, rre);
    if (threadIdx.x == 0) {
        const double rr = rrs + rre;
        const double normr = sqrt(rr);
        const int conv = normr < st->thresh;          // krylov.jl:76
        const double rho_old = st->rho2[it & 1];
        const double beta = rr / rho_old;             // krylov.jl:83-84
        s_conv = conv;
        s_beta = beta;
        if (blockIdx.x == 0) {
            st->iters = it + 1;
            st->last_rms = sqrt(rr / N);              // krylov.jl:90
            if (conv) st->done = 1;
            else { st->rho2[(it + 1) & 1] = rr; st->rho_old = rho_old; st->rho = rr; st->beta = beta; }
        }
    }
    __syncthreads();
    if (s_conv) return;
    const double beta = s_beta;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = r[i] + beta * p[i];  // krylov.jl:85
}

// ---- CG iteration in TWO dependent launches ---------------------------------------------------------------
// k_cg_pmv_f (this kernel) -> k_cg_update_f.  The direction update p = r + beta p of iteration it-1 (krylov.jl:85)
// moves into the matvec of iteration it: every workgroup derives ||r||, the exit test and beta from the r.r partials
// of the previous iteration (as k_cg_p_f does), forms the new p on its 32 x 8 tile plus a one-point ring (the ring is
// recomputed, not communicated: p is double-buffered, nobody reads what a neighbour is writing), stores the tile,
// applies the operator from the LDS image and reduces p .* p_hat.  Same operations on the same operands as the
// three-launch form: x, r, p, the iteration count and the returned residual are bit-identical.
constexpr int CGX = BX, CGY = BY;   // the tiles of k_cg_matvec_dot: identical dot-product partials, identical alpha
__global__ __launch_bounds__(256) void k_cg_pmv_f(const double* __restrict__ p_old, double* __restrict__ p_new,
                                                   const double* __restrict__ r, double* __restrict__ ph, int nx, int ny,
                                                   double hx2, double hy2, double c, double* __restrict__ pq_partials,
                                                   const double* __restrict__ rr_partials, int nrr,
                                                   FprSolveState* __restrict__ st, int it, double N)
{
    __shared__ double red[32];
    __shared__ double s_beta;
    __shared__ int s_conv;
    __shared__ double tile[CGY + 2][CGX + 2];
    if (st->done) return;
    const int tid = threadIdx.x + CGX * threadIdx.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int i0 = blockIdx.x * CGX, j0 = blockIdx.y * CGY;
    // operands first: their loads do not depend on beta and overlap the reduction of the r.r partials below
    const int i = i0 + tx, j = j0 + ty;
    const bool own = i < nx && j < ny;
    const size_t id = (size_t)(own ? i : 0) + (size_t)nx * (own ? j : 0);
    const double p_own = p_old[id], r_own = it > 0 ? r[id] : 0.0;
    constexpr int NRING = 2 * (CGX + 2) + 2 * CGY;   // bottom row, top row, left column, right column
    int hx = 0, hy = 0;
    if (tid < CGX + 2) { hx = tid; hy = 0; }
    else if (tid < 2 * (CGX + 2)) { hx = tid - (CGX + 2); hy = CGY + 1; }
    else if (tid < 2 * (CGX + 2) + CGY) { hx = 0; hy = tid - 2 * (CGX + 2) + 1; }
    else if (tid < NRING) { hx = CGX + 1; hy = tid - 2 * (CGX + 2) - CGY + 1; }
    const int ri = i0 + hx - 1, rj = j0 + hy - 1;
    const bool ring = tid < NRING && ri >= 0 && rj >= 0 && ri < nx && rj < ny;
    const size_t rid = (size_t)(ring ? ri : 0) + (size_t)nx * (ring ? rj : 0);
    const double p_ring = p_old[rid], r_ring = it > 0 ? r[rid] : 0.0;
    double beta = 0.0;
    if (it > 0) {   // exit test and beta of iteration it-1 (krylov.jl:73-84)
        double rrs, rre;
        fpr_sum_partials_256_s2(rr_partials, nrr, red, rrs, rre);
        if (tid == 0) {
            const double rr = rrs + rre;
            const double normr = sqrt(rr);
            const int conv = normr < st->thresh;              // krylov.jl:76
            const double rho_old = st->rho2[(it - 1) & 1];
            const double b = rr / rho_old;                    // krylov.jl:83-84
            s_conv = conv;
            s_beta = b;
            if (blockIdx.x == 0 && blockIdx.y == 0) {
                st->iters = it;
                st->last_rms = sqrt(rr / N);                  // krylov.jl:90
                if (conv) st->done = 1;
                else { st->rho2[it & 1] = rr; st->rho_old = rho_old; st->rho = rr; st->beta = b; }
            }
        }
        __syncthreads();
        if (s_conv) return;
        beta = s_beta;
    }
    // krylov.jl:85 (p = r = b before the first iteration)
    tile[ty + 1][tx + 1] = own ? (it > 0 ? r_own + beta * p_own : p_own) : 0.0;
    if (tid < NRING) tile[hy][hx] = ring ? (it > 0 ? r_ring + beta * p_ring : p_ring) : 0.0;
    __syncthreads();
    double accs = 0.0, acce = 0.0;
    if (own) {
        const double t = tile[ty + 1][tx + 1];
        p_new[id] = t;
        double q;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            q = (((tile[ty + 1][tx + 2] - 2 * t) + tile[ty + 1][tx]) / hx2 + ((tile[ty + 2][tx + 1] - 2 * t) + tile[ty][tx + 1]) / hy2) - c * t;   // lap_at
            ph[id] = q;
        } else {
            q = ph[id];   // boundary of p_hat keeps b's values (krylov.jl:61, 68)
        }
        fpr_s2_add_prod(accs, acce, t, q);
    }
    __syncthreads();   // (red was read by thread 0 above)
    fpr_block_sum_s2<256>(accs, acce, red);
    if (tid == 0) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        pq_partials[2 * b] = accs; pq_partials[2 * b + 1] = acce;
    }
}

// exit test of the LAST enqueued iteration (its successor's k_cg_pmv_f would have made it): one workgroup
__global__ __launch_bounds__(256) void k_cg_tail_f(const double* __restrict__ rr_partials, int nrr, FprSolveState* __restrict__ st,
                                                    int it, double N)
{
    __shared__ double red[32];
    if (st->done) return;
    double rrs, rre;
    fpr_sum_partials_256_s2(rr_partials, nrr, red, rrs, rre);
    if (threadIdx.x == 0) {
        const double rr = rrs + rre;
        const double normr = sqrt(rr);
        const double rho_old = st->rho2[(it - 1) & 1];
        st->iters = it;
        st->last_rms = sqrt(rr / N);
        if (normr < st->thresh) st->done = 1;
        else { st->rho2[it & 1] = rr; st->rho_old = rho_old; st->rho = rr; st->beta = rr / rho_old; }
    }
}

// p .= r + beta p  (krylov.jl:85)
__global__ __launch_bounds__(256) void k_cg_p(double* __restrict__ p, const double* __restrict__ r, size_t n,
                                               const FprSolveState* __restrict__ st)
{
    if (st->done) return;
    const double beta = st->beta;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = r[i] + beta * p[i];
}

