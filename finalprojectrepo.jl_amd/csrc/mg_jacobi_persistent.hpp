// mg_jacobi_persistent.hpp -- the damped-Jacobi coarse solve on a large coarse grid as FEW launches of MANY groups of sweeps
// Part of multigrid2d.hip (included there behind mg_cg_persistent.hpp); kernels only, the host side is in multigrid2d.hip.
#pragma once

// k_jacobi_patch applies 8 sweeps per launch to 32 x 32 regions whose inner tile is exact, and a launch costs ~10.7 us of which the
// sweeps are 3-4 (tools/exp_patch_sweeps.py: 8.5 / 8.75 / 9.5 / 11.4 / 12.1 us for 1 / 2 / 4 / 6 / 8 sweeps per launch under the event
// timer): the rest is the launch boundary, the replay of the previous group's exit test and the per-launch prologue / epilogue.  The
// five-level V-cycle of BASELINE config 3 spends 5140 sweeps = 643 such launches per cycle on its 257^2 coarse grid (multigrid.jl:147-159).
//
// k_jacobi_persist keeps the workgroups resident for G groups of S sweeps.  What a kernel boundary did between two groups -- make
// every tile of group g visible to the regions of group g + 1 -- is done between NEIGHBOURS only: a workgroup stores its tile with
// sc1 (write-through) stores, drains them, and publishes "group g done" in a flag word; before it loads the region of the next group
// (sc1 loads: the vector L1 is not coherent) it waits for the flags of its up to eight neighbours.  No workgroup waits for the whole
// grid.  Three work buffers rotate (group g reads W[g % 3] and writes W[(g + 1) % 3]; the buffer a workgroup overwrites in group g was
// last read by its neighbours in group g - 2, and their flags have said g since), the launch's input X is never written, so the
// exit test multigrid.jl:152-155 can be evaluated AFTER the launch from the per-sweep partial sums (k_jacobi_check_groups: same sums in
// the same order as k_jacobi_check_multi) and, when it fires inside the launch, the exact number of sweeps is replayed from X by the
// ordinary launches.  Same point arithmetic as k_jacobi_patch / k_sweep2d: bit-identical fields, identical norms.
// All workgroups have to be resident at once (the host asks the runtime); every wait is bounded and raises an abort flag.
struct JacPersistArgs {
    const double* X;        // input of the launch (never written)
    double* W[3];           // rotating work buffers; the result of the launch is W[(ngroups - 1) % 3]
    const double* rhs;
    int nx, ny;
    double C, _h2, fac;
    int ngroups, nsw_last;  // groups of this launch; sweeps of the last one (<= S), all others S
    double* partials;       // [group][sweep][block] sums of res^2 over the own tile
    int* flags;             // one word per workgroup: groups finished since the start of the SOLVE (= g0 before the launch)
    int g0;                 // groups of the solve that earlier launches have done
    int* abort_flag;
    const FprSolveState* state;
};

constexpr int JACP_SC1 = 16;   // cache-policy bit sc1 of the raw buffer intrinsics (gfx94x / gfx950)

template <int S, int P>
__global__ __launch_bounds__((P / 2) * (P / 2)) void k_jacobi_persist(JacPersistArgs a)
{
    constexpr int T = P - 2 * S;
    constexpr int HT = P / 2;
    constexpr int NT = HT * HT, NWV = (NT + 63) / 64;
    static_assert(T > 0 && P % 2 == 0 && HT == 16, "32 x 32 regions: a row of threads is one 16-lane DPP row");
    __shared__ __attribute__((aligned(16))) double img[2][P * P];
    __shared__ double red[NWV][S];
    __shared__ int s_abort;
    if (a.state->done) return;    // a launch enqueued behind the group that met the criterion
    const int tid = threadIdx.x;
    const int nx = a.nx, ny = a.ny;
    const int ty = tid / HT, tx = tid - ty * HT;
    const int lx = 2 * tx, ly = 2 * ty;
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T;
    const int gx = x0 - S + lx, gy = y0 - S + ly;
    const int blk = blockIdx.x + gridDim.x * blockIdx.y, nblk = gridDim.x * gridDim.y;
    double ff[2][2];
    bool inter[2][2], own[2][2];
    unsigned voff[2][2], vst[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int gi = gx + c, gj = gy + b;
            const bool in = gi >= 0 && gj >= 0 && gi < nx && gj < ny;
            const size_t g = in ? (size_t)gi + (size_t)nx * gj : 0;
            ff[b][c] = in ? a.rhs[g] : 0.0;
            inter[b][c] = gi >= 1 && gj >= 1 && gi < nx - 1 && gj < ny - 1;
            own[b][c] = in && gi >= x0 && gi < x0 + T && gj >= y0 && gj < y0 + T;
            voff[b][c] = in ? (unsigned)g * 8u : FPR_OOR;        // outside the grid: the load is dropped and returns 0
            vst[b][c] = own[b][c] ? (unsigned)g * 8u : FPR_OOR;
        }
    // the (up to) eight neighbours whose tiles this region reads: lanes 0..7 of wave 0 watch one flag each
    int nb = -1;
    if (tid < 8) {
        const int k = tid < 4 ? tid : tid + 1;                    // 0..8 without the centre
        const int bx = (int)blockIdx.x + k % 3 - 1, by = (int)blockIdx.y + k / 3 - 1;
        if (bx >= 0 && by >= 0 && bx < (int)gridDim.x && by < (int)gridDim.y) nb = bx + (int)gridDim.x * by;
    }
    const int xl = lx > 0 ? lx - 1 : lx, xr = lx + 2 < P ? lx + 2 : lx + 1;
    const int yd = ly > 0 ? ly - 1 : ly, yu = ly + 2 < P ? ly + 2 : ly + 1;
    (void)xl; (void)xr;
    if (tid == 0) s_abort = 0;
    bool alive = true;
    for (int g = 0; g < a.ngroups && alive; ++g) {
        const double* in = g == 0 ? a.X : a.W[g % 3];
        double* out = a.W[(g + 1) % 3];
        if (g > 0) {
            if (tid < 64) {
                int ab = 0;
                if (nb >= 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(a.flags + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.g0 + g) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 0x3ff) == 0) {
                            if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
                            if (spins > (1u << 21)) {   // seconds: a neighbour that never became resident
                                __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ab = 1;
                                break;
                            }
                        }
                    }
                }
                ab = __any(ab);
                if (tid == 0 && ab) s_abort = 1;
            }
            __syncthreads();
            if (s_abort) { alive = false; break; }
        }
        const __amdgpu_buffer_rsrc_t rIn = fpr_rsrc(in), rOut = fpr_rsrc(out);
        const int nsw = (g == a.ngroups - 1) ? a.nsw_last : S;
        double u[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                u[b][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rIn, voff[b][c], 0, JACP_SC1));
        double acc[S];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = 0.0;
        {
            double* w = img[0];
            *reinterpret_cast<double2*>(&w[lx + P * ly]) = make_double2(u[0][0], u[0][1]);
            *reinterpret_cast<double2*>(&w[lx + P * (ly + 1)]) = make_double2(u[1][0], u[1][1]);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s < nsw) {
                const double* cur = img[s & 1];
                double* nxt = img[(s + 1) & 1];
                // x-neighbours of a patch from the adjacent lanes' registers (zero beyond the region's edge: such garbage stays more
                // than S cells away from the own tile), y-neighbours from the LDS image
                const double wl0 = fpr_dpp<0x111>(u[0][1]), wl1 = fpr_dpp<0x111>(u[1][1]);
                const double er0 = fpr_dpp<0x101>(u[0][0]), er1 = fpr_dpp<0x101>(u[1][0]);
                const double2 dn = *reinterpret_cast<const double2*>(&cur[lx + P * yd]);
                const double2 up = *reinterpret_cast<const double2*>(&cur[lx + P * yu]);
                const double E[2][2] = {{u[0][1], er0}, {u[1][1], er1}};
                const double Wv[2][2] = {{wl0, u[0][0]}, {wl1, u[1][0]}};
                const double Nn[2][2] = {{u[1][0], u[1][1]}, {up.x, up.y}};
                const double Sx[2][2] = {{dn.x, dn.y}, {u[0][0], u[0][1]}};
                double un[2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double rr = ((((E[b][c] + Wv[b][c]) + Nn[b][c]) + Sx[b][c]) - a.C * u[b][c]) * a._h2 - ff[b][c];
                        un[b][c] = inter[b][c] ? u[b][c] + a.fac * rr : u[b][c];
                        if (inter[b][c] && own[b][c]) acc[s] += rr * rr;
                    }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) u[b][c] = un[b][c];
                *reinterpret_cast<double2*>(&nxt[lx + P * ly]) = make_double2(u[0][0], u[0][1]);
                *reinterpret_cast<double2*>(&nxt[lx + P * (ly + 1)]) = make_double2(u[1][0], u[1][1]);
            }
            __syncthreads();
        }
        // the own tile, write-through; its stores have left every wave before the flag is published behind the workgroup barrier
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fpr_u2v, u[b][c]), rOut, vst[b][c], 0, JACP_SC1);
        {
            const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const double v = fpr_wave_sum_all(acc[s]);
                if (lane == 0) red[wv][s] = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __syncthreads();
        if (tid < S) {
            double v = red[0][tid];
#pragma unroll
            for (int w = 1; w < NWV; ++w) v += red[w][tid];
            a.partials[((size_t)g * S + tid) * nblk + blk] = v;     // read by k_jacobi_check_groups behind the launch
        }
        if (tid == 0) __hip_atomic_store(a.flags + blk, a.g0 + g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (red / img are rewritten only behind the next group's barriers)
    }
}

// Behind a launch of k_jacobi_persist: the per-sweep exit test of its groups, in order (multigrid.jl:152-155).  One workgroup per GROUP
// sums that group's partial lists exactly as k_jacobi_check_multi does (wave w: sweeps w, w + 4; 64-lane strided sums, then the DPP wave
// sum) and leaves the S sums in global memory (agent-scope atomic stores); the workgroup whose arrival at the counter is the last one
// then evaluates the tests in order, 64 at a time.  (One workgroup summing all 256 lists took 96 us behind a 195 us launch.)
// group0 = index of the launch's first group in the whole solve (FprSolveState::group is global).  *counter is zero between launches.
__global__ __launch_bounds__(256) void k_jacobi_check_groups(FprSolveState* st, const double* __restrict__ partials, int nblk, int ngroups,
                                                             int S, int nsw_last, double N, int group0, const int* __restrict__ abort_flag,
                                                             double* __restrict__ gsums, int* __restrict__ counter)
{
    __shared__ int s_last;
    if (st->done) return;      // (uniform over the launch: st is written only by the last workgroup, behind everybody's arrival)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = blockIdx.x;
    const int nsw = g == ngroups - 1 ? nsw_last : S;
    for (int sw = wv; sw < nsw; sw += 4) {
        double t = 0.0;
        for (int i = lane; i < nblk; i += 64) t += partials[((size_t)g * S + sw) * nblk + i];
        t = fpr_wave_sum_all(t);
        if (lane == 0) __hip_atomic_store(gsums + g * S + sw, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the sums have left every wave
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    if (wv == 0) {
        const int total = (ngroups - 1) * S + nsw_last;      // sweeps of the launch, in order: index = group * S + sweep
        int first = -1;
        double rms_last = 0.0;
        for (int base = 0; base < total && first < 0; base += 64) {
            const int q = base + lane;
            const double v = q < total ? __hip_atomic_load(gsums + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            const double rms = q < total ? sqrt(v / N) : 0.0;
            const unsigned long long hit = __ballot(q < total && rms < st->thresh);
            const int n_here = total - base < 64 ? total - base : 64;
            const int last = hit ? (int)__builtin_ctzll(hit) : n_here - 1;
            rms_last = __shfl(rms, last, 64);
            if (hit) first = base + last;
        }
        if (lane == 0) {
            *counter = 0;
            if (abort_flag && *abort_flag) {
                st->done = -1;            // a wait timed out: the host gives up on this form and resumes from this launch's input
                st->group = group0;       // (st->iters counts the sweeps of the launches before it)
            } else if (first >= 0) {
                st->iters += first + 1;
                st->last_rms = rms_last;
                st->done = 1;
                st->group = group0 + first / S;
                st->redo = first % S + 1;
            } else {
                st->iters += total;
                st->last_rms = rms_last;
            }
        }
    }
}
