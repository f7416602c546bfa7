// mg_cg_persistent.hpp -- cg! (krylov.jl:55-91) as one persistent launch of 16 workgroups
// Part of multigrid2d.hip (included there, in this order: mg_march.hpp, mg_cg.hpp, mg_small.hpp, mg_cg_persistent.hpp,
// mg_mid.hpp); kernels only, the host side that launches them is in multigrid2d.hip.
#pragma once

// ---- cg! as ONE launch: a persistent 16-workgroup kernel with two grid barriers per iteration -----------------------------
// The two-launch form spends ~5 us per launch boundary, 10 us per iteration, on a 257^2 problem whose arithmetic takes
// well under 1 us.  tools/gridsync_probe.hip: a dependent launch costs 2.8 us (4.5 under rocprofv3), cooperative_groups'
// grid.sync() 2.9 us for 16 workgroups and 32 us for 256 -- but a hand-written counter barrier between 16 workgroups costs
// 1.1 us.  So: 4 x 4 workgroups of 1024 threads, every vector of the iteration (x, r, p, p_hat) in REGISTERS (<= 5 points per
// thread), the direction p of the tile plus a one-point ring in LDS for the operator; the only data that travel between
// workgroups are the tile-edge values of r (through the r array, coherent accesses) and one partial sum per workgroup and
// dot product.  The ring of p is recomputed from the neighbour's r and the ring's own previous p (same operations as the
// owner), so two barriers per iteration suffice: after the p.p_hat partials and after the r.r partials + r edges.
// 16 workgroups are resident together on every device this runs on; every spin is bounded and raises an abort flag all
// workgroups honour, so a workgroup that does not arrive ends the solve with an error instead of hanging the device.
// Same operations per point as k_cg_pmv_f / k_cg_update_f; the dot products are Dot2 sums (fpr_internal.hpp: every addition's
// and product's rounding error is carried along and the total rounded once), summed per workgroup and then over the 16
// workgroups -- another order than the 64x4-tile partials of the other forms, the same value: all forms and the oracle agree
// bit for bit.
// Two geometries (template parameters NB workgroups as NBX x NBX tiles, NT threads each; <= 5 points per thread either way):
//   16 x 1024 threads (rounds 2-3): one workgroup per CU on 16 CUs -- four waves per SIMD, so every per-thread instruction costs four
//             issue slots: 2.4 us of an iteration were arithmetic (7.5 us per iteration with the Dot2 sums of round 4);
//   64 x 256 threads (round 4 default): the same points on 64 CUs with ONE wave per SIMD.  A barrier is bounded by one store -> poll
//             hand-off whatever the number of slots a wave polls (64 lanes, one or two words each), so the arithmetic between the
//             barriers shrinks fourfold at the price of 64 instead of 16 arrivals per barrier.
constexpr int CGP_PPT = 5, CGP_RPT = 1;   // tile / ring points per thread
// (16 x 256 threads x 17 points: 9.0 us per iteration against 6.5 -- the two divisions per point of lap_at then weigh 2.8 us)
struct CgpArgs {
    const double* b;
    double* x_out;         // solution (whole array written)
    double* r_glob;        // N doubles: tile-edge values of r are exchanged through it
    double* part;          // 4 x 2 CGP_NB slots (8 bytes every 128: a workgroup's (s, e) pair, each word its own arrival flag), all CGP_EMPTY before the launch
    unsigned* ctr;         // [1] abort flag
    FprSolveState* st;
    int nx, ny, Nmax;
    double hx2, hy2, c, tol, N;
    double ihx2, ihy2;     // pow2: 1 / hx2, 1 / hy2 (exact)
    int pow2;              // hx2 and hy2 are powers of two (every grid of the multigrid hierarchy: h = 2^-k): x / hx2 == x * (1 / hx2) bit for
                           // bit -- an exact scaling either way -- and the operator needs no division (two per point: ~0.4 us per iteration)
                           // iteration} through r_glob (2 N doubles): one sc1 store, one sc1 load issued BEFORE barrier 2 and checked behind
                           // it -- no drain in front of the barrier words, no load round trip behind them (k_jacobi_persist_tag's hand-off)
    int fences;            // option handoff_fences = 1: the hand-off of the r edges ALSO inside the HIP memory model -- an agent-scope release
                           // fence (buffer_wbl2 sc1) in front of the barrier words and an acquire fence (buffer_inv sc1) behind the poll, on top
                           // of the sc1 stores / drains / sc1 loads that carry it by themselves on gfx950 (the conservative switch)
};

// The tile-edge values of r travel between workgroups as `sc1` (write-through / L1-bypassing) stores and loads through a raw buffer
// descriptor: every byte handed over is stored sc1 and loaded sc1, every storing wave drains its stores (s_waitcnt vmcnt(0)) before
// the workgroup barrier that precedes the publication of the barrier word, and the readers load after the barrier that follows
// their poll -- the hand-off form MI355X_MICROARCH.md lists as valid WITHOUT an agent-scope release (buffer_wbl2: 1.7-6.5 us) and
// acquire (buffer_inv); with both in barrier 2 that barrier cost 3.2 us on 64 workgroups against 2.0 us for barrier 1.
constexpr int CGP_SC1 = 16;   // cache-policy bit of the raw buffer intrinsics on gfx94x / gfx950: sc1
__device__ __forceinline__ double cgp_ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, CGP_SC1));
}
__device__ __forceinline__ void cgp_st_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fpr_u2v, v), r, voff, 0, CGP_SC1);
}

// Grid barrier and all-reduce in ONE round trip: workgroup b stores its partial sum -- a Dot2 pair (s, e), two words -- into its own
// two slots of the set that belongs to this barrier; lanes 0..31 of wave 0 of every workgroup each watch one slot until it no longer
// holds the EMPTY pattern (a NaN payload no sum produces) -- every word is its own arrival flag, so the two need no order.  FOUR sets rotate: when a workgroup publishes for
// barrier g it first empties its slot of set (g+2) mod 4, last used at barrier g-2 (everybody has read that one: they all
// published g-1 since).  That slot is polled next at barrier g+2; between the emptying and that poll lie the owner's
// publications g and g+1 (stores of one lane to the same cache-line-sized slots, issued in order behind an s_waitcnt).  DRAIN:
// the barrier that hands over the tile-edge values of r -- every wave waits for its (sc1) stores before the workgroup barrier.  Returns the 16 pairs summed in workgroup order and rounded once (s + e), in every thread;
// *ok = false if the wait timed out (abort raised for everybody).
constexpr unsigned long long CGP_EMPTY = 0x7ff8dead0badf00dull;
constexpr int CGP_SLOT_STRIDE = 16;   // 8-byte words between two slots
// Grid barrier + all-reduce.  Inside the workgroup: NT = 1024: the 1024 pairs are folded 4 -> 1 through LDS by the first four waves
// (one per SIMD) before the DPP wave sums (all 16 waves summing their own lanes, four per SIMD, cost 0.5 us per barrier more);
// NT = 256: every wave has its SIMD to itself and sums its lanes at once.  The <= 4 wave totals are merged by thread 0 / 1, which
// publish s / e.  Across workgroups: lane w < NB of wave 0 polls the two words of workgroup w; the NB pairs are summed by one DPP
// wave sum (lanes >= NB hold zeros): a fixed tree, a handful of registers (summing 32 words one after the other out of LDS in
// every thread cost 67 spilled VGPRs and 4 us per iteration).
template <bool DRAIN, int NB, int NT>
__device__ __forceinline__ double cgp_allsum(unsigned long long* slots, unsigned* abort_flag, double vs, double ve, unsigned& gen, double* red,
                                             double* fold, double* gsum, int* s_abort, bool* ok, int fences = 0, bool drain = true)
{
    static_assert(NB <= 64 && (NT == 256 || NT == 1024), "one polling lane per workgroup; four summing waves per workgroup");
    ++gen;
    // (a slot per 128-byte line: the pollers of different slots do not queue at one memory channel)
    unsigned long long* set = slots + (gen & 3u) * (2 * NB * CGP_SLOT_STRIDE);
    unsigned long long* nxt = slots + ((gen + 2u) & 3u) * (2 * NB * CGP_SLOT_STRIDE);
    if constexpr (NT == 1024) {
        fold[threadIdx.x] = vs;
        fold[NT + threadIdx.x] = ve;
        __syncthreads();
        if (threadIdx.x < NT / 4) {
#pragma unroll
            for (int k = 1; k < 4; ++k) fpr_s2_merge(vs, ve, fold[threadIdx.x + k * (NT / 4)], fold[NT + threadIdx.x + k * (NT / 4)]);
        }
    }
    if (threadIdx.x < 256) {
        fpr_wave_sum_all_s2(vs, ve);
        if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = vs; red[2 * (threadIdx.x >> 6) + 1] = ve; }
    }
    // wave totals in LDS; the workgroup's earlier sc1 stores (tile-edge values of r) have left every wave before the barrier, the
    // barrier words are published behind it
    if constexpr (DRAIN) { if (drain) __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0)
    __syncthreads();
    if (threadIdx.x < 64) {                // wave 0 sums the workgroup, publishes, polls and sums the grid
        const int w = threadIdx.x;
        if (w < 2) {                       // thread 0 publishes s, thread 1 e
            double bs = red[0], be = red[1];
#pragma unroll
            for (int k = 1; k < 4; ++k) fpr_s2_merge(bs, be, red[2 * k], red[2 * k + 1]);
            const int slot = (2 * blockIdx.x + w) * CGP_SLOT_STRIDE;
            if (DRAIN && fences) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (inline asm: the compiler may drop the builtin's wait behind a drained scoreboard)
            }
            asm volatile("" ::: "memory");   // nothing of the hand-off moves across the publication
            __hip_atomic_store(&nxt[slot], CGP_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long bits = (unsigned long long)__double_as_longlong(w == 0 ? bs : be);
            if (bits == CGP_EMPTY) bits ^= 1ull;   // (a sum that happens to be this very NaN stays a NaN)
            __hip_atomic_store(&set[slot], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int ab = 0;
        double gs = 0.0, ge = 0.0;
        if (w < NB) {                      // lane w waits for the two words of workgroup w (each its own arrival flag)
            unsigned spins = 0;
            unsigned long long b0 = CGP_EMPTY, b1 = CGP_EMPTY;
            while (true) {
                if (b0 == CGP_EMPTY) b0 = __hip_atomic_load(&set[(2 * w) * CGP_SLOT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b1 == CGP_EMPTY) b1 = __hip_atomic_load(&set[(2 * w + 1) * CGP_SLOT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b0 != CGP_EMPTY && b1 != CGP_EMPTY) break;
                if ((++spins & 0x3ff) == 0) {
                    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
                    if (spins > (1u << 22)) {   // seconds, not minutes
                        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ab = 1;
                        break;
                    }
                }
            }
            gs = __longlong_as_double((long long)b0);
            ge = __longlong_as_double((long long)b1);
        }
        ab = __any(ab);
        asm volatile("" ::: "memory");       // the readers' sc1 loads stay behind the poll
        if (DRAIN && fences) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        fpr_wave_sum_all_s2(gs, ge);
        if (threadIdx.x == 0) { *s_abort = ab; *gsum = gs + ge; }
    }
    __syncthreads();
    *ok = *s_abort == 0;
    return *gsum;
}

__global__ void k_cgp_slots_init(unsigned long long* slots, int nb)
{
    for (int t = threadIdx.x; t < 4 * 2 * nb; t += blockDim.x) slots[t * CGP_SLOT_STRIDE] = CGP_EMPTY;
}

template <int CGP_NB, int CGP_NBX, int CGP_NT>
__global__ __launch_bounds__(CGP_NT) void k_cg_persistent(CgpArgs a)
{
    static_assert(CGP_NBX * CGP_NBX == CGP_NB, "square arrangement of tiles");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[8];
    __shared__ double fold[CGP_NT == 1024 ? 2 * CGP_NT : 2];
    __shared__ double gsum;
    __shared__ int s_abort;
    const int tid = threadIdx.x;
    const int bx = blockIdx.x % CGP_NBX, by = blockIdx.x / CGP_NBX;
    const int nx = a.nx, ny = a.ny;
    const int twm = (nx + CGP_NBX - 1) / CGP_NBX, thm = (ny + CGP_NBX - 1) / CGP_NBX;   // nominal tile
    const int i0 = bx * twm, j0 = by * thm;
    int tw = nx - i0 < twm ? nx - i0 : twm, th = ny - j0 < thm ? ny - j0 : thm;
    if (tw < 0) tw = 0;
    if (th < 0) th = 0;
    const int npt = tw * th;                       // points of this tile (0: the workgroup only takes part in the barriers)
    const int lw = twm + 2;                        // LDS row length: tile + ring
    double* P = sm;                                // (thm + 2) x lw image of p: tile cell (ti, tj) at (ti + 1) + lw * (tj + 1)
    const __amdgpu_buffer_rsrc_t rR = fpr_rsrc(a.r_glob);   // tile-edge values of r (sc1 stores / loads)
    unsigned gen = 0;
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(a.part);   // 4 sets x 16 slots, all EMPTY at the start
    bool ok = true;
    if (tid == 0) s_abort = 0;
    // this thread's points
    int li[CGP_PPT], gi[CGP_PPT];                  // LDS index, global index (-1: none)
    bool inter[CGP_PPT], edge[CGP_PPT];
    double x[CGP_PPT], r[CGP_PPT], p[CGP_PPT], q[CGP_PPT];
    const float rtw = tw > 0 ? 1.0f / (float)tw : 0.0f;
#pragma unroll
    for (int k = 0; k < CGP_PPT; ++k) {
        const int idx = tid + k * CGP_NT;
        gi[k] = -1; li[k] = 0; inter[k] = false; edge[k] = false;
        x[k] = 0.0; r[k] = 0.0; p[k] = 0.0; q[k] = 0.0;
        if (idx < npt) {
            const int tj = mgs_row(idx, rtw), ti = idx - tj * tw;
            const int i = i0 + ti, j = j0 + tj;
            gi[k] = i + nx * j;
            li[k] = (ti + 1) + lw * (tj + 1);
            inter[k] = i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1;
            edge[k] = ti == 0 || tj == 0 || ti == tw - 1 || tj == th - 1;
            const double v = a.b[gi[k]];
            r[k] = v; p[k] = v; q[k] = v;          // krylov.jl:61-66: r = p = b, p_hat starts as b (its boundary keeps b), x = 0
        }
    }
    // ring cells of this thread (at most CGP_RPT): bottom row, top row, left column, right column of the (tw+2) x (th+2) frame
    int rl[CGP_RPT], rg[CGP_RPT];
    double p_ring[CGP_RPT];
    {
        const int nring = npt > 0 ? 2 * (tw + 2) + 2 * th : 0;
#pragma unroll
        for (int m = 0; m < CGP_RPT; ++m) {
            const int t = tid + m * CGP_NT;
            rl[m] = -1; rg[m] = -1;
            if (t < nring) {
                int hx, hy;
                if (t < tw + 2) { hx = t; hy = 0; }
                else if (t < 2 * (tw + 2)) { hx = t - (tw + 2); hy = th + 1; }
                else if (t < 2 * (tw + 2) + th) { hx = 0; hy = t - 2 * (tw + 2) + 1; }
                else { hx = tw + 1; hy = t - 2 * (tw + 2) - th + 1; }
                const int ri = i0 + hx - 1, rj = j0 + hy - 1;
                rl[m] = hx + lw * hy;
                if (ri >= 0 && rj >= 0 && ri < nx && rj < ny) rg[m] = ri + nx * rj;
            }
            p_ring[m] = rg[m] >= 0 ? a.b[rg[m]] : 0.0;   // p = b before the first iteration
        }
    }
    // rho = sum(r .* r) with r = b (krylov.jl:64), threshold tol * ||b|| (:57-58)
    double rho = 0.0, rho_old = 0.0, rr = 0.0;
    int it = 0;
    bool conv = false, alive = true;
    {
        double acc0 = 0.0, err0 = 0.0;
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k) fpr_s2_add_prod(acc0, err0, r[k], r[k]);
        rho = cgp_allsum<false, CGP_NB, CGP_NT>(slots, &a.ctr[1], acc0, err0, gen, red, fold, &gsum, &s_abort, &ok);
        alive = ok;
        rr = rho;
    }
    const double thresh = a.tol * sqrt(rho);
    for (; alive && it < a.Nmax; ++it) {
        // ---- exit test and beta of iteration it-1 (krylov.jl:73-84), new direction (:85), operator and p.p_hat (:68-69) ----
        double beta = 0.0;
        if (it > 0) {
            if (sqrt(rr) < thresh) { conv = true; break; }      // :76 (every thread of every workgroup holds the same sum)
            rho_old = rho;
            beta = rr / rho_old;                                 // :83-84
            rho = rr;
#pragma unroll
            for (int k = 0; k < CGP_PPT; ++k) p[k] = r[k] + beta * p[k];
#pragma unroll
            for (int m = 0; m < CGP_RPT; ++m)   // sc1 loads behind barrier 2 (its poll, then the workgroup barrier)
                if (rg[m] >= 0) p_ring[m] = cgp_ld_sc1(rR, (unsigned)rg[m] * 8u) + beta * p_ring[m];
        }
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k)
            if (gi[k] >= 0) P[li[k]] = p[k];
#pragma unroll
        for (int m = 0; m < CGP_RPT; ++m)
            if (rl[m] >= 0) P[rl[m]] = p_ring[m];
        __syncthreads();
        double acc = 0.0, acce = 0.0;
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k) {
            if (inter[k]) {
                const double t = p[k];
                const int l = li[k];
                const double dxx = (P[l + 1] - 2 * t) + P[l - 1], dyy = (P[l + lw] - 2 * t) + P[l - lw];
                q[k] = a.pow2 ? (dxx * a.ihx2 + dyy * a.ihy2) - a.c * t : (dxx / a.hx2 + dyy / a.hy2) - a.c * t;   // lap_at
            }
            if (gi[k] >= 0) fpr_s2_add_prod(acc, acce, p[k], q[k]);
        }
        const double pq = cgp_allsum<false, CGP_NB, CGP_NT>(slots, &a.ctr[1], acc, acce, gen, red, fold, &gsum, &s_abort, &ok);   // barrier 1 of the iteration
        if (!ok) { alive = false; break; }
        // ---- alpha, x and r (krylov.jl:69-72), r.r; the tile-edge values of r go to the neighbours ----
        const double alpha = rho / pq;
        double acc2 = 0.0, acc2e = 0.0;
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k) {
            if (gi[k] >= 0) {
                x[k] = x[k] + alpha * p[k];
                const double rn = r[k] - alpha * q[k];
                r[k] = rn;
                fpr_s2_add_prod(acc2, acc2e, rn, rn);
                if (edge[k]) cgp_st_sc1(rR, (unsigned)gi[k] * 8u, rn);   // write-through; drained before barrier 2 publishes (an atomic
                                                                          // store each would be issued behind an s_waitcnt of its own)
            }
        }
        rr = cgp_allsum<true, CGP_NB, CGP_NT>(slots, &a.ctr[1], acc2, acc2e, gen, red, fold, &gsum, &s_abort, &ok, a.fences, true);   // barrier 2 (publishes the r edges)
        if (!ok) { alive = false; break; }
    }
    if (alive && !conv && it == a.Nmax && a.Nmax > 0) conv = sqrt(rr) < thresh;   // the loop ran out: the last norm (k_cg_tail_f)
#pragma unroll
    for (int k = 0; k < CGP_PPT; ++k)
        if (gi[k] >= 0) a.x_out[gi[k]] = x[k];              // krylov.jl:88
    if (blockIdx.x == 0 && tid == 0) {
        a.st->iters = it;
        a.st->last_rms = sqrt(rr / a.N);                   // :90 (r = b if the loop body never ran)
        a.st->thresh = thresh;
        a.st->done = alive ? (conv ? 1 : 0) : -1;          // -1: a barrier timed out
        a.st->rho = rr;
    }
}

