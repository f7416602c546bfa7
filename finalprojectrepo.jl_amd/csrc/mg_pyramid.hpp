// mg_pyramid.hpp -- four levels of the way DOWN in one launch, halos exchanged instead of recomputed (experiment, option mg_pyr_down)
// Part of multigrid2d.hip (included there behind mg_mid.hpp); kernels only, the host side is mid_path() in multigrid2d.hip.
#pragma once

// ================================================================================================
// k_pyr_down: the pre-smoothing passes of FOUR consecutive levels P > A > B > C (each: two sweeps from the zero initial guess,
// multigrid.jl:124-125 after :132, + residual + injection, :128-131) in ONE launch -- what the level's own pass + k_mid_down do in two.
// k_mid_down recomputes in LDS the halo every workgroup needs on three levels (53^2 points of A for 32^2 owned) and is bound by the
// instruction issue of that redundancy; a fourth level on top costs more than its pass (option mg_mid4, EXPERIMENTS 13.14).  Here a
// workgroup owns one tile per level (64^2 / 32^2 / 16^2 / 8^2 for the 4 x 4 tile of level D it produces) and works on the tile grown by
// two: the two sweeps and the residual need the level's right-hand side there.  The top level reads that from memory; every level below
// receives the halo from its neighbours as data-tagged granules {value, tag} (the hand-off of k_jacobi_persist_tag: ONE sc1 store, ONE
// sc1 load, re-polled until the tag is the launch's; no flags, no drains) -- one hand-off per level.  Same point arithmetic as the
// per-level kernels: every array a later pass reads (the pre-smoothed field and the right-hand side of every level) is bit-identical.
// All workgroups must be resident together (<= one per compute unit); a poll that times out poisons level D's right-hand side with NaN
// (this form has no replay yet: an experiment).
// ================================================================================================
struct PyrArgs {
    MidLevel L[4];                // P, A, B, C: f (P only: read from memory), tmp, fout = the next level's right-hand side, nx, ny, C, _h2, fac
    void* ftag[3];                // right-hand sides of A, B, C as granules (16 bytes per point)
    int nxD, nyD;
    double* uD;                   // level D's zero initial guess is written (:132)
    int apply_BCs;
    const int* skip;
    unsigned long long tag_base;  // the right-hand side of level l (1..3) carries tag_base + l
    FprFinishArgs fin;            // partials != null: the finish of the cycle before, done by one more row of workgroups
    long long* prof;              // diagnostic (option mg_pyr_prof): wall_clock64 stamps (100 MHz) of thread 0 of workgroup (gx/2, gy/2), 16 slots
};

__global__ __launch_bounds__(MID_NT_DOWN) void k_pyr_down(PyrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (a.fin.partials && blockIdx.y == gridDim.y - 1) {
        if (blockIdx.x == 0) fpr_cycle_finish_body<true>(a.fin, sm);
        return;
    }
    if (a.skip && *a.skip) return;
    constexpr int NT = MID_NT_DOWN;
    const int tid = threadIdx.x;
    const int ntx = (a.nxD - 1) / MID_TD > 0 ? (a.nxD - 1) / MID_TD : 1, nty = (a.nyD - 1) / MID_TD > 0 ? (a.nyD - 1) / MID_TD : 1;
    const int bx = blockIdx.x, by = blockIdx.y;
    MidReg ownD;
    ownD.x0 = bx * MID_TD; ownD.x1 = (bx == ntx - 1) ? a.nxD - 1 : ownD.x0 + MID_TD - 1;
    ownD.y0 = by * MID_TD; ownD.y1 = (by == nty - 1) ? a.nyD - 1 : ownD.y0 + MID_TD - 1;
    auto own_at = [&](int scale, int nx, int ny) {   // the tile of a level whose index is scale * the level-D index
        MidReg o;
        o.x0 = scale * ownD.x0; o.x1 = (bx == ntx - 1) ? nx - 1 : scale * (ownD.x1 + 1) - 1;
        o.y0 = scale * ownD.y0; o.y1 = (by == nty - 1) ? ny - 1 : scale * (ownD.y1 + 1) - 1;
        return o;
    };
    bool failed = false;
    int pslot = 0;
    const bool profme = a.prof && tid == 0 && bx == ntx / 2 && by == nty / 2;
    auto stamp = [&]() { if (profme) a.prof[pslot++] = wall_clock64(); };
    stamp();   // 0: start
    double* F = sm;
    MidReg own = own_at(16, a.L[0].nx, a.L[0].ny);
    MidReg R = mid_grow(own, 2, a.L[0].nx, a.L[0].ny);
    {   // right-hand side of the top level from memory
        const MidLevel& L = a.L[0];
        const int w = mid_w(R), n = mid_n(R);
        const float rw = 1.0f / (float)w;
        for (int idx = tid; idx < n; idx += NT) {
            const int jj = mgs_row(idx, rw), ii = idx - jj * w;
            F[idx] = L.f[(size_t)(R.x0 + ii) + (size_t)L.nx * (R.y0 + jj)];
        }
    }
    __syncthreads();
    stamp();   // 1: top level's right-hand side loaded
    int scale = 16;
    for (int l = 0; l < 4; ++l, scale >>= 1) {
        const MidLevel& L = a.L[l];
        const MidReg Q = mid_grow(own, 1, L.nx, L.ny);
        const int nr = mid_n(R), nq = mid_n(Q);
        double* U1 = F + nr;
        double* U2 = U1 + nr;
        double* Fn = U2 + nq;    // the next level's right-hand side on ITS tile grown by two
        mid_sweep_z2(F, R, U2, Q, L.nx, L.ny, L.C, L._h2, L.fac, NT);   // :124-125 from the zero initial guess, one pass
        __syncthreads();
        stamp();   // 2 + 3 l: the level's two sweeps
        {   // the owned part of the pre-smoothed field goes to memory (the post-smoothing pass reads it)
            const int w = mid_w(own), n = mid_n(own);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = own.x0 + ii, j = own.y0 + jj;
                L.tmp[(size_t)i + (size_t)L.nx * j] = U2[mid_at(Q, i, j)];
            }
        }
        // residual at the injected points of the owned tile = right-hand side of the next level (:128-131; Neumann columns :355-357)
        const int nxc = 1 + (L.nx - 1) / 2, nyc = 1 + (L.ny - 1) / 2;
        const MidReg ownc = (l < 3) ? own_at(scale >> 1, nxc, nyc) : ownD;
        const MidReg Rc = (l < 3) ? mid_grow(ownc, 2, nxc, nyc) : ownD;
        const unsigned long long tg = a.tag_base + (unsigned long long)(l + 1);
        {
            const int w = mid_w(ownc), n = mid_n(ownc), wq = mid_w(Q);
            const float rw = 1.0f / (float)w;
            const __amdgpu_buffer_rsrc_t rOut = fpr_rsrc(l < 3 ? a.ftag[l] : (void*)a.uD);
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int ic = ownc.x0 + ii, jc = ownc.y0 + jj;
                int is = ic;
                if (a.apply_BCs) is = (ic == 0) ? 1 : (ic == nxc - 1 ? nxc - 2 : ic);
                double v = 0.0;
                if (is >= 1 && is <= nxc - 2 && jc >= 1 && jc <= nyc - 2) {
                    const int q = mid_at(Q, 2 * is, 2 * jc);
                    v = ((((U2[q + 1] + U2[q - 1]) + U2[q + wq]) + U2[q - wq]) - L.C * U2[q]) * L._h2 - F[mid_at(R, 2 * is, 2 * jc)];
                }
                const size_t g = (size_t)ic + (size_t)nxc * jc;
                L.fout[g] = v;
                if (l < 3) {
                    Fn[mid_at(Rc, ic, jc)] = v;
                    // ... and as a granule {value, tag}, write-through: the neighbours are already polling for it
                    const unsigned long long vb = __builtin_bit_cast(unsigned long long, v);
                    fpr_u4v qv;
                    qv.x = (unsigned)vb; qv.y = (unsigned)(vb >> 32); qv.z = (unsigned)tg; qv.w = (unsigned)(tg >> 32);
                    __builtin_amdgcn_raw_buffer_store_b128(qv, rOut, (unsigned)(g * 16u), 0, JACP_SC1);
                } else {
                    a.uD[g] = 0.0;   // zero initial guess of level D (:132)
                }
            }
        }
        stamp();   // 3 + 3 l: field stored, residual stored and published
        if (l == 3) break;
        {   // the halo of the next level's right-hand side from the neighbours (granules; re-polled until the tag is this launch's)
            const int w = mid_w(Rc), n = mid_n(Rc);
            const float rw = 1.0f / (float)w;
            const __amdgpu_buffer_rsrc_t rIn = fpr_rsrc(a.ftag[l]);
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int ic = Rc.x0 + ii, jc = Rc.y0 + jj;
                if (ic >= ownc.x0 && ic <= ownc.x1 && jc >= ownc.y0 && jc <= ownc.y1) continue;
                const unsigned off = (unsigned)(((size_t)ic + (size_t)nxc * jc) * 16u);
                unsigned spins = 0;
                double v = 0.0;
                while (true) {
                    const fpr_u4v qv = __builtin_amdgcn_raw_buffer_load_b128(rIn, off, 0, JACP_SC1);
                    const unsigned long long t = ((unsigned long long)qv.w << 32) | qv.z;
                    if (t == tg) { v = __builtin_bit_cast(double, ((unsigned long long)qv.y << 32) | qv.x); break; }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 18)) { failed = true; break; }   // a neighbour that never became resident
                }
                Fn[idx] = v;
            }
        }
        __syncthreads();
        // next level: its right-hand side moves to the front of the arena
        {
            const int n = mid_n(Rc);
            double tmpv[3];   // (up to 37^2 = 1369 values for 1024 threads)
            int cnt = 0;
            for (int idx = tid; idx < n; idx += NT) tmpv[cnt++] = Fn[idx];
            __syncthreads();
            cnt = 0;
            for (int idx = tid; idx < n; idx += NT) sm[idx] = tmpv[cnt++];
        }
        __syncthreads();
        stamp();   // 4 + 3 l: the next level's halo received and moved
        F = sm;
        own = ownc;
        R = Rc;
    }
    // a poll that gave up: poison what the rest of the cycle reads -- the solve's norms turn NaN instead of silently wrong
    __syncthreads();
    int* flag = reinterpret_cast<int*>(sm);
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (failed) *flag = 1;
    __syncthreads();
    if (*flag && tid == 0) a.L[3].fout[(size_t)ownD.x0 + (size_t)a.nxD * ownD.y0] = __builtin_nan("");
}
