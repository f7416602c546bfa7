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
    long long* prof;        // diagnostic (option mg_jacp_prof): per workgroup 8 words -- wall_clock64 ticks (100 MHz) thread 0 spent waiting for
                            // its neighbours / loading the region / sweeping / storing and draining / publishing, groups, XCC id
    int fences;             // option handoff_fences = 1: agent-scope release in front of the flag store and acquire behind the poll as well (see CgpArgs)
};

constexpr int JACP_SC1 = 16;   // cache-policy bit sc1 of the raw buffer intrinsics (gfx94x / gfx950)

// ================================================================================================================================
// k_jacobi_persist_tag: the same groups of sweeps, but the hand-off between neighbours is the DATA itself.
//
// In k_jacobi_persist a hand-off is three dependent memory round trips on top of the sweeps (tools/exp_jacp_prof.py, 257^2, per group of 8:
// sweeps 2.1 us; stores of the tile drained + sums 0.8; the neighbours' flags seen 2.1; region loaded 0.8 -- 5.9 us): tile stores ->
// s_waitcnt -> flag store -> (neighbour) flag poll -> region load.  Here every cell travels as a 16-byte granule {value, tag} written by
// ONE sc1 (write-through) store and read by ONE sc1 load; tag = solve epoch and group that produced the value.  A consumer loads the
// cells of its region straight away and re-loads those whose tag is not the one it needs yet: one store -> load round trip, no flag, no
// drain.  (16-byte sc1 accesses of one lane are observed untorn on gfx950 -- MI355X_MICROARCH, hand-off forms; not an architectural
// guarantee: the flag form stays available, option mg_jacp_tagged = 0, and every solve of the tests is compared bit for bit with it.)
// Ordering needs no flags either: a workgroup that stores group g into W[(g+1)%3] has loaded its neighbours' results of group g-1, so
// they are past reading that buffer (their input of group g-2) -- and a reader never sees a tag NEWER than the one it waits for.
// Own cells stay in registers from group to group; only the halo of the region is loaded.
// ================================================================================================================================
struct JacTagArgs {
    const double* X;            // plain input of the solve's FIRST launch (x_tagged = 0)
    const void* Xg;             // granule input of a later launch (the last group's buffer of the launch before)
    void* W[3];                 // rotating granule buffers (16 bytes per cell)
    const double* rhs;
    int nx, ny, x_tagged;
    double C, _h2, fac;
    int ngroups, nsw_last;
    double* partials;
    int g0;
    long long tag_base;         // the group with index q of the solve writes tag tag_base + q + 1 (so its input carries tag_base + q)
    int* abort_flag;
    const FprSolveState* state;
    long long* prof;
};

template <int S, int P, int PY = 2>
__global__ __launch_bounds__((P / 2) * (P / PY)) void k_jacobi_persist_tag(JacTagArgs a)
{
    constexpr int T = P - 2 * S;
    constexpr int HT = P / 2;
    constexpr int NT = HT * (P / PY), NWV = (NT + 63) / 64;
    static_assert(T > 0 && P % 2 == 0 && HT == 16 && (PY == 1 || PY == 2), "32 x 32 regions: a row of threads is one 16-lane DPP row");
    __shared__ __attribute__((aligned(16))) double img[2][P * P];
    __shared__ double red[NWV][S];
    __shared__ int s_abort;
    if (a.state->done) return;
    const int tid = threadIdx.x;
    const int nx = a.nx, ny = a.ny;
    const int ty = tid / HT, tx = tid - ty * HT;
    const int lx = 2 * tx, ly = PY * ty;
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T;
    const int gx = x0 - S + lx, gy = y0 - S + ly;
    const int blk = blockIdx.x + gridDim.x * blockIdx.y, nblk = gridDim.x * gridDim.y;
    double ff[PY][2];
    bool inter[PY][2], own[PY][2], ing[PY][2];
    unsigned cell[PY][2];
#pragma unroll
    for (int b = 0; b < PY; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int gi = gx + c, gj = gy + b;
            const bool in = gi >= 0 && gj >= 0 && gi < nx && gj < ny;
            const size_t g = in ? (size_t)gi + (size_t)nx * gj : 0;
            ff[b][c] = in ? a.rhs[g] : 0.0;
            ing[b][c] = in;
            inter[b][c] = gi >= 1 && gj >= 1 && gi < nx - 1 && gj < ny - 1;
            own[b][c] = in && gi >= x0 && gi < x0 + T && gj >= y0 && gj < y0 + T;
            cell[b][c] = (unsigned)g;
        }
    const int yd = ly > 0 ? ly - 1 : ly, yu = ly + PY < P ? ly + PY : ly + PY - 1;
    if (tid == 0) s_abort = 0;
    __syncthreads();
    bool alive = true;
    long long pd0 = 0, pd1 = 0, pd2 = 0, pd3 = 0, pd4 = 0, pt = 0;
    const bool prof = a.prof != nullptr;
    double u[PY][2];
#pragma unroll
    for (int b = 0; b < PY; ++b) { u[b][0] = 0.0; u[b][1] = 0.0; }
    for (int g = 0; g < a.ngroups && alive; ++g) {
        if (prof) pt = wall_clock64();
        const long long want = a.tag_base + (long long)(a.g0 + g);
        if (g == 0 && !a.x_tagged) {
            const __amdgpu_buffer_rsrc_t rX = fpr_rsrc(a.X);
#pragma unroll
            for (int b = 0; b < PY; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    u[b][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rX, ing[b][c] ? cell[b][c] * 8u : FPR_OOR, 0, 0));
        } else {
            const __amdgpu_buffer_rsrc_t rIn = fpr_rsrc(g == 0 ? a.Xg : a.W[g % 3]);
            // cells still to be fetched: everything inside the grid that this thread does not hold already (own cells carry over)
            bool need[PY][2];
            bool any = false;
#pragma unroll
            for (int b = 0; b < PY; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) { need[b][c] = ing[b][c] && !(own[b][c] && g > 0); any |= need[b][c]; }
            unsigned spins = 0;
            int ab = 0;
            while (any) {
                fpr_u4v q[PY][2];
#pragma unroll
                for (int b = 0; b < PY; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        q[b][c] = __builtin_amdgcn_raw_buffer_load_b128(rIn, need[b][c] ? cell[b][c] * 16u : FPR_OOR, 0, JACP_SC1);
                any = false;
#pragma unroll
                for (int b = 0; b < PY; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const long long tg = (long long)(((unsigned long long)q[b][c].w << 32) | q[b][c].z);
                        if (need[b][c] && tg == want) {
                            u[b][c] = __builtin_bit_cast(double, ((unsigned long long)q[b][c].y << 32) | q[b][c].x);
                            need[b][c] = false;
                        }
                        any |= need[b][c];
                    }
                if (any) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 0xff) == 0) {
                        if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
                        if (spins > (1u << 20)) {   // seconds: a neighbour that never became resident
                            __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ab = 1;
                            break;
                        }
                    }
                }
            }
            if (ab) s_abort = 1;
        }
        if (prof) { const long long t = wall_clock64(); pd0 += t - pt; pt = t; }
        const int nsw = (g == a.ngroups - 1) ? a.nsw_last : S;
        double acc[S];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = 0.0;
        {
            double* w = img[0];
#pragma unroll
            for (int b = 0; b < PY; ++b) *reinterpret_cast<double2*>(&w[lx + P * (ly + b)]) = make_double2(u[b][0], u[b][1]);
        }
        __syncthreads();
        if (s_abort) { alive = false; break; }      // (uniform: written before the barrier, by whoever gave up)
        if (prof) { const long long t = wall_clock64(); pd1 += t - pt; pt = t; }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s < nsw) {
                const double* cur = img[s & 1];
                double* nxt = img[(s + 1) & 1];
                double E[PY][2], Wv[PY][2], Nn[PY][2], Sx[PY][2];
#pragma unroll
                for (int b = 0; b < PY; ++b) {
                    E[b][0] = u[b][1]; E[b][1] = fpr_dpp<0x101>(u[b][0]);
                    Wv[b][0] = fpr_dpp<0x111>(u[b][1]); Wv[b][1] = u[b][0];
                }
                const double2 dn = *reinterpret_cast<const double2*>(&cur[lx + P * yd]);
                const double2 up = *reinterpret_cast<const double2*>(&cur[lx + P * yu]);
#pragma unroll
                for (int b = 0; b < PY; ++b) {
                    Nn[b][0] = b == PY - 1 ? up.x : u[b == PY - 1 ? b : b + 1][0];
                    Nn[b][1] = b == PY - 1 ? up.y : u[b == PY - 1 ? b : b + 1][1];
                    Sx[b][0] = b == 0 ? dn.x : u[b == 0 ? 0 : b - 1][0];
                    Sx[b][1] = b == 0 ? dn.y : u[b == 0 ? 0 : b - 1][1];
                }
                double un[PY][2];
#pragma unroll
                for (int b = 0; b < PY; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double rr = ((((E[b][c] + Wv[b][c]) + Nn[b][c]) + Sx[b][c]) - a.C * u[b][c]) * a._h2 - ff[b][c];
                        un[b][c] = inter[b][c] ? u[b][c] + a.fac * rr : u[b][c];
                        if (inter[b][c] && own[b][c]) acc[s] += rr * rr;
                    }
#pragma unroll
                for (int b = 0; b < PY; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) u[b][c] = un[b][c];
#pragma unroll
                for (int b = 0; b < PY; ++b) *reinterpret_cast<double2*>(&nxt[lx + P * (ly + b)]) = make_double2(u[b][0], u[b][1]);
            }
            __syncthreads();
        }
        if (prof) { const long long t = wall_clock64(); pd2 += t - pt; pt = t; }
        // the own tile as granules {value, tag of this group}, write-through: the neighbours are already polling for them
        {
            const __amdgpu_buffer_rsrc_t rOut = fpr_rsrc(a.W[(g + 1) % 3]);
            const unsigned long long tg = (unsigned long long)(want + 1);
#pragma unroll
            for (int b = 0; b < PY; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned long long v = __builtin_bit_cast(unsigned long long, u[b][c]);
                    fpr_u4v q;
                    q.x = (unsigned)v; q.y = (unsigned)(v >> 32); q.z = (unsigned)tg; q.w = (unsigned)(tg >> 32);
                    __builtin_amdgcn_raw_buffer_store_b128(q, rOut, own[b][c] ? cell[b][c] * 16u : FPR_OOR, 0, JACP_SC1);
                }
        }
        {
            const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const double v = fpr_wave_sum_all(acc[s]);
                if (lane == 0) red[wv][s] = v;
            }
        }
        __syncthreads();
        if (prof) { const long long t = wall_clock64(); pd3 += t - pt; pt = t; }
        if (tid < S) {
            double v = red[0][tid];
#pragma unroll
            for (int w = 1; w < NWV; ++w) v += red[w][tid];
            a.partials[((size_t)g * S + tid) * nblk + blk] = v;
        }
        // (red is rewritten behind the next group's barriers; img[0] by this thread's own rows only, and its readers are past the sweeps)
        if (prof) { const long long t = wall_clock64(); pd4 += t - pt; pt = t; }
    }
    if (prof && tid == 0) {
        long long* q = a.prof + (size_t)blk * 8;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        q[0] += pd0; q[1] += pd1; q[2] += pd2; q[3] += pd3; q[4] += pd4; q[5] += a.ngroups; q[6] = (long long)(xcc & 7u);
    }
}

// granules -> plain doubles (the result of a solve; the input of the launch an exit or a time-out fell into)
__global__ __launch_bounds__(256) void k_jacp_untag(const void* __restrict__ g, double* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = reinterpret_cast<const double*>(g)[2 * i];
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
