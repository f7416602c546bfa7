// mg_march.hpp -- register-rolling marching passes of the fine multigrid levels (2 sweeps, 4 sweeps between two cycles)
// Part of multigrid2d.hip (included there, in this order: mg_march.hpp, mg_cg.hpp, mg_small.hpp, mg_cg_persistent.hpp,
// mg_mid.hpp); kernels only, the host side that launches them is in multigrid2d.hip.
#pragma once

// ---- buffer addressing helpers (descriptor base + per-lane byte offset + scalar byte offset) ---------------------
// An offset of FPR_OOR is beyond every descriptor's num_records: the hardware drops the access (loads return 0).  The
// marching kernels use it instead of branches around loads / stores: with conditional memory instructions hipcc cannot
// count the operations younger than a prefetch and waits for more than it has to (finding 1).
constexpr unsigned FPR_OOR = 0x7fffffffu;
typedef unsigned fpr_u2v __attribute__((ext_vector_type(2)));
typedef unsigned fpr_u4v __attribute__((ext_vector_type(4)));   // a 16-byte granule {value, tag} of the data-tagged hand-offs
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fpr_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)FPR_OOR, 0x00020000);
}
__device__ __forceinline__ double fpr_bld(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void fpr_bst(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, double x)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fpr_u2v, x), r, voff, soff, 0);
}

// ---- two fused Jacobi sweeps, register-rolling march in y (fine levels) ------------------------------
// A wave owns a strip of 64 columns (one column per lane) and marches down a chunk of rows keeping a
// 3-row window of u (sweep 0) and of the once-smoothed field (sweep 1) in registers; x-neighbours come
// from wavefront shuffles, y-neighbours from the window.  Strips overlap by 4 columns (a wave loads 64
// columns and owns the 60 in the middle: the two outer lanes on each side only feed the stencils), so
// waves never communicate: no LDS, no barrier.  One pass reads u and f once and writes u once for TWO
// sweeps (multigrid.jl:124-125 / :142-143).  Point arithmetic is that of k_sweep2d: bit-identical.
// PROLONG: the input is corrected on the fly, u = uin - P(corr_c) (multigrid.jl:136-139 fused into the
// post-smoothing pass: the prolongation/correction pass over the fine grid disappears).
// RESTRICT: a third stage evaluates the residual of the twice-smoothed field at the injected points
// (even row, even column) from a 3-row window of the output and writes the coarse right-hand side and
// the zero initial coarse correction (multigrid.jl:128-132): the residual/restriction pass disappears
// too.  Strips then overlap by 6 columns and chunks by 3+2 rows.  Coarse boundary points get 0; the
// Neumann rows of apply_BCs are copied afterwards by k_bc_neumann on the (small) coarse array.
//
// ---- the seam between two V-cycles on the finest level: FOUR sweeps in one pass ------------------------------
// MGsolve_2DPoisson! (multigrid.jl:57-71) runs V-cycle after V-cycle on the same arrays: the post-smoothing pair of
// cycle k (:142-143, on the field corrected by the prolongated coarse solution, :136-139) is followed -- if the exit
// test :70 does not end the loop -- by the pre-smoothing pair of cycle k+1 (:124-125) and its residual + injection
// (:128-132).  The seam kernels (k_seam_march_v2 / _v3) do all of that in ONE pass over the finest grid: the same register-rolling march as
// k_smooth2_march with five 3-row windows (the corrected input, the fields after sweeps 1..4) and a residual stage,
//     read  uin (+ P(corr_c) on the fly), f                          (8 + 8 + 2 bytes per point)
//     write the twice pre-smoothed field of cycle k+1, its restricted residual, the zero coarse guess   (8 + 2 + 2)
// -- 30 bytes per point and cycle where the two separate passes move 26 + 28.  The field after sweep 2 is u at the end
// of cycle k: it is not stored (its residual norm, the r_rms of :252, is summed exactly like k_smooth2_march<NORM>
// does); if that norm ends the loop the host replays the plain post-smoothing pass from the untouched inputs
// (fpr_mgsolve2d).  Same point arithmetic as k_sweep2d / prolong_bf: all fields bit-identical to the separate passes.
// Strips overlap by 10 columns (54 owned of 64), chunks by 5 + 4 rows.  BCS: the boundary conditions the loop re-applies
// between two cycles (:60-62) act on the field between sweep 2 and sweep 3 (Neumann columns; see below), the correction
// is prolongated with its Neumann rows, and the host copies the Neumann columns of the coarse residual afterwards
// (k_bc_neumann, as behind the separate pre-smoothing pass).
//
// ---- k_seam_march_v2: the pass with ~half the vector instructions per row of the first version (round 3; that version is history) ----
// Counters of the first version (profiles/r2_mg_seam_pmc.txt): 49.8 M vector instructions per launch at 4097^2 = 138 per
// wave-row = a VALU floor of 81-91 us beside a memory floor of 91 us for a pass that takes 130 us -- as much VALU-bound as
// memory-bound.  What changed, results unchanged (same operations on the same operands in the same order per point):
//   * chunks start on an EVEN row (one more row of overlap), so the parity of every row of the unrolled loop is a
//     compile-time constant: the prolongation of an even row has two terms, of an odd row four -- no per-row branch on
//     "did the coarse row change", no per-term selects;
//   * the coarse correction is fetched through a buffer descriptor with the per-lane column masks (interior coarse column,
//     odd fine column) folded into the lane's byte offset and the per-row masks (interior coarse row) into the scalar
//     offset: an excluded term is a load that returns 0 (w * 0 added is what the first version's `cond ? w*c : 0.0` adds),
//     and the coarse rows sit in a ring of three register pairs with compile-time slots, loaded three rows before use
//     (the first version loaded them in the row that used them: every other row waited for its newest load);
//   * boundary COLUMNS keep their value by a per-lane factor (fac or 0) instead of a select per sweep, boundary ROWS (first
//     / last row of the grid: first / last chunk only) by a uniform branch;
//   * the residual stage and its stores run in the rows that are injected (every other one), the norm is accumulated
//     unmasked under a uniform row test and masked per lane once at the end.
template <bool BCS, int PF = 4>   // PF rows of u and f in flight per lane (4 or 6: the loop is unrolled by 12)
__global__ __launch_bounds__(256) void k_seam_march_v2(const double* __restrict__ uin, const double* __restrict__ f,
                                                        double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                        double fac, int rows_per_chunk, int nstrips,
                                                        double* __restrict__ partials, const double* __restrict__ corr_c,
                                                        double* __restrict__ res_c_out, double* __restrict__ corr_c_out,
                                                        const int* __restrict__ skip)
{
    if (skip && *skip) return;
    __shared__ double red[16];
    constexpr int HX = 5;                                    // feeder lanes on each side of a strip
    constexpr int SW = 64 - 2 * HX;                          // columns owned by a strip
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + w;
    const bool active = strip < nstrips;
    const int gi = strip * SW - HX + lane;                   // global column of this lane
    const bool col_ok = active && gi >= 0 && gi < nx;
    const int gic = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);  // clamped for loads
    const bool col_bnd = gi <= 0 || gi >= nx - 1;            // domain boundary column (or outside)
    const bool owner = col_ok && lane >= HX && lane < 64 - HX;
    const int y0 = blockIdx.y * rows_per_chunk;              // host: rows_per_chunk is even
    const int y1 = (y0 + rows_per_chunk < ny) ? y0 + rows_per_chunk : ny;  // output rows [y0, y1)
    const int rs = y0 - 6 < 0 ? 0 : y0 - 6;                  // EVEN: row rs + T has the parity of T
    double acc = 0.0;
    if (active) {
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        int gis = gic;  // Neumann columns of the prolongated correction (part2_utils.jl:35-39)
        if (BCS) gis = (gic == 0) ? 1 : (gic == nx - 1 ? nx - 2 : gic);
        const int p_io = gis & 1, p_icl = gis >> 1, p_ich = (p_icl + 1 < nxc) ? p_icl + 1 : nxc - 1;
        const bool p_sx0 = p_icl >= 1 && p_icl <= nxc - 2, p_sx1 = p_io && (p_icl + 1 <= nxc - 2);
        const bool p_inx = gis >= 1 && gis <= nx - 2;
        const double wE = p_io ? 0.5 : 1.0, wO = p_io ? 0.25 : 0.5;   // prolong_bf's weight in an even / odd fine row
        const unsigned vcl = (p_inx && p_sx0) ? (unsigned)p_icl * 8u : FPR_OOR;   // excluded terms read 0
        const unsigned vch = (p_inx && p_sx1) ? (unsigned)p_ich * 8u : FPR_OOR;
        const __amdgpu_buffer_rsrc_t rCor = fpr_rsrc(corr_c);
        const int crow = nxc * 8;
        double cs[3][2];                                     // coarse rows j, j+1, j+2 (slot = (j - rs/2) mod 3): columns icl, ich
        auto ldc = [&](int slot, int j) {
            const int so = (j >= 1 && j <= nyc - 2) ? j * crow : (int)FPR_OOR;   // a boundary coarse row contributes nothing
            cs[slot][0] = fpr_bld(rCor, vcl, so);
            cs[slot][1] = fpr_bld(rCor, vch, so);
        };
        const int rowB = nx * 8;
        const __amdgpu_buffer_rsrc_t rUin = fpr_rsrc(uin + (size_t)nx * rs), rF = fpr_rsrc(f + (size_t)nx * rs);
        const __amdgpu_buffer_rsrc_t rUout = fpr_rsrc(uout + (size_t)nx * rs);
        const unsigned vld = (unsigned)gic * 8u;
        const unsigned vst = owner ? (unsigned)gi * 8u : FPR_OOR;   // lanes that own nothing store out of range
        // corrected input of row `row` = rs + LR (mod 12): u - P(corr), terms and order of prolong_bf
        auto ldu = [&](auto LRc, int row) {
            constexpr int LR = decltype(LRc)::value;
            constexpr int JR = LR >> 1;
            constexpr int SA = JR % 3, SB = (JR + 1) % 3, SN = (JR + 2) % 3;
            const int rc = row > ny - 1 ? ny - 1 : row;
            const double v = fpr_bld(rUin, vld, (rc - rs) * rowB);
            double pv;
            if constexpr ((LR & 1) == 0) {
                ldc(SN, (row >> 1) + 2);           // used three rows from now; its slot held the row last used one row ago
                pv = wE * cs[SA][0];
                pv = pv + wE * cs[SA][1];
            } else {
                pv = wO * cs[SA][0];
                pv = pv + wO * cs[SA][1];
                pv = pv + wO * cs[SB][0];
                pv = pv + wO * cs[SB][1];
            }
            return v - pv;
        };
        auto ldf = [&](int r) { const int rc = r > ny - 1 ? ny - 1 : r; return fpr_bld(rF, vld, (rc - rs) * rowB); };
        double wv[5][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        double fw[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        using I0 = std::integral_constant<int, 0>;
        ldc(0, rs >> 1);
        ldc(1, (rs >> 1) + 1);
        wv[0][0] = ldu(I0{}, rs);
        fw[0] = ldf(rs);
        static_assert(PF == 4 || PF == 6, "ring slots are compile-time constants of a loop unrolled by 12");
        double pu[PF], pfv[PF];
        pu[0] = ldu(std::integral_constant<int, 1>{}, rs + 1); pfv[0] = ldf(rs + 1);
        pu[1] = ldu(std::integral_constant<int, 2>{}, rs + 2); pfv[1] = ldf(rs + 2);
        pu[2] = ldu(std::integral_constant<int, 3>{}, rs + 3); pfv[2] = ldf(rs + 3);
        pu[3] = ldu(std::integral_constant<int, 4>{}, rs + 4); pfv[3] = ldf(rs + 4);
        if constexpr (PF == 6) {
            pu[4] = ldu(std::integral_constant<int, 5>{}, rs + 5); pfv[4] = ldf(rs + 5);
            pu[5] = ldu(std::integral_constant<int, 6>{}, rs + 6); pfv[5] = ldf(rs + 6);
        }
        const __amdgpu_buffer_rsrc_t rResC = fpr_rsrc(res_c_out), rCorC = fpr_rsrc(corr_c_out);
        const unsigned vstc = (owner && !(gi & 1)) ? (unsigned)(gi >> 1) * 8u : FPR_OOR;
        // BCS: Neumann columns of the coarse right-hand side (:355-357), see k_smooth2_march
        const unsigned vstr = (BCS && (gi == 0 || gi == nx - 1)) ? FPR_OOR : vstc;
        const unsigned vstn = (BCS && owner && (gi == 2 || gi == nx - 3)) ? (gi == 2 ? 0u : (unsigned)(nxc - 1) * 8u) : FPR_OOR;
        const double facL = col_bnd ? 0.0 : fac;             // a boundary column keeps its value: mid + 0 * rr
        const bool cint_col = (gi >> 1) >= 1 && (gi >> 1) <= nxc - 2;
        // one Jacobi sweep at row j of a field whose rows j-1, j, j+1 are (lo, mid, hi); rr = residual used by the update
        auto sweep = [&](double lo, double mid, double hi, double fv, int j, double& rr) {
            const double L = fpr_lane_up1z(mid), R = fpr_lane_down1z(mid);
            rr = ((((R + L) + hi) + lo) - C * mid) * _h2 - fv;
            double un = mid + facL * rr;
            if (j <= 0 || j >= ny - 1) { asm volatile("" ::: "memory"); un = mid; }   // uniform: first / last row of the grid
            return un;
        };
        auto step = [&](auto Tc, int r) {
            constexpr int T = decltype(Tc)::value;               // (r - rs) mod 12
            constexpr int Q = T % PF, M = T % 3, F = T % 6;
            constexpr int M1 = (M + 1) % 3, M2 = (M + 2) % 3;    // slots of rows r-2 (= r+1) and r-1
            auto fs = [](int k) { return (F - k + 6) % 6; };     // slot of f row r-k
            double an, fn;
            asm volatile("v_mov_b64 %0, %1" : "=v"(an) : "v"(pu[Q]));
            asm volatile("v_mov_b64 %0, %1" : "=v"(fn) : "v"(pfv[Q]));
            pu[Q] = ldu(std::integral_constant<int, T + 1 + PF>{}, r + 1 + PF);   // issue the loads of row r+1+PF
            pfv[Q] = ldf(r + 1 + PF);
            double rr;
            // ---- cycle k, post-smoothing (:142-143): sweeps 1 and 2 at rows r-1, r-2 ----
            wv[1][M2] = sweep(wv[0][M1], wv[0][M2], wv[0][M], fw[fs(1)], r - 1, rr);
            const int j2 = r - 2;
            double u2 = sweep(wv[1][M], wv[1][M1], wv[1][M2], fw[fs(2)], j2, rr);   // u at the end of cycle k (not stored)
            if constexpr (BCS) {
                // apply_boundary_conditions! between the cycles (multigrid.jl:60-62): Neumann columns copy their inner neighbour
                const double fromR = fpr_lane_down1z(u2), fromL = fpr_lane_up1z(u2);
                u2 = (gi == 0) ? fromR : ((gi == nx - 1) ? fromL : u2);
            }
            wv[2][M1] = u2;
            if (j2 >= y0 && j2 < y1 && j2 > 0 && j2 < ny - 1) {   // uniform; lanes that own nothing are dropped at the end
                asm volatile("" ::: "memory");
                acc = acc + rr * rr;                              // r_rms of cycle k (:252)
            }
            // ---- cycle k+1, pre-smoothing (:124-125): sweeps 3 and 4 at rows r-3, r-4 ----
            wv[3][M] = sweep(wv[2][M2], wv[2][M], wv[2][M1], fw[fs(3)], r - 3, rr);
            const int j4 = r - 4;
            const double u4 = sweep(wv[3][M1], wv[3][M2], wv[3][M], fw[fs(4)], j4, rr);
            {
                const bool row_own = j4 >= y0 && j4 < y1;      // uniform
                fpr_bst(rUout, vst, row_own ? (j4 - rs) * rowB : (int)FPR_OOR, u4);   // unconditional (see FPR_OOR)
            }
            wv[4][M2] = u4;
            // ---- residual of the pre-smoothed field at row r-5 (even in every odd step), injected at even columns (:128-132) ----
            if constexpr ((T & 1) == 1) {
                const int j5 = r - 5;
                const double mid = wv[4][M1];
                const double L = fpr_lane_up1z(mid), R = fpr_lane_down1z(mid);
                const double rres = ((((R + L) + wv[4][M2]) + wv[4][M]) - C * mid) * _h2 - fw[fs(5)];
                const int jc = j5 >> 1;
                const bool cint = cint_col && jc >= 1 && jc <= nyc - 2;
                const bool row_inj = j5 >= y0 && j5 < y1;       // uniform
                const int sc = row_inj ? jc * crow : (int)FPR_OOR;
                fpr_bst(rResC, vstr, sc, cint ? rres : 0.0);
                if constexpr (BCS) fpr_bst(rResC, vstn, sc, cint ? rres : 0.0);
                fpr_bst(rCorC, vstc, sc, 0.0);
            }
            wv[0][M1] = an;            // row r+1 takes the slot of row r-2
            fw[(F + 1) % 6] = fn;      // row r+1 takes the slot of row r-5
        };
        const int rend = y1 + 4;
        int r = rs;
#define FPR_SEAM_STEP(T) step(std::integral_constant<int, T>{}, r + T)
        for (; r + 11 <= rend; r += 12) {
            FPR_SEAM_STEP(0); FPR_SEAM_STEP(1); FPR_SEAM_STEP(2); FPR_SEAM_STEP(3); FPR_SEAM_STEP(4); FPR_SEAM_STEP(5);
            FPR_SEAM_STEP(6); FPR_SEAM_STEP(7); FPR_SEAM_STEP(8); FPR_SEAM_STEP(9); FPR_SEAM_STEP(10); FPR_SEAM_STEP(11);
        }
#undef FPR_SEAM_STEP
#define FPR_SEAM_TAIL(T) if (r <= rend) { step(std::integral_constant<int, T>{}, r); ++r; }
        FPR_SEAM_TAIL(0) FPR_SEAM_TAIL(1) FPR_SEAM_TAIL(2) FPR_SEAM_TAIL(3) FPR_SEAM_TAIL(4) FPR_SEAM_TAIL(5)
        FPR_SEAM_TAIL(6) FPR_SEAM_TAIL(7) FPR_SEAM_TAIL(8) FPR_SEAM_TAIL(9) FPR_SEAM_TAIL(10)
#undef FPR_SEAM_TAIL
        acc = (owner && !col_bnd) ? acc : 0.0;
    }
    const double sblk = fpr_block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
}

// ---- k_seam_march_v3 (round 5): TWO columns per lane ----------------------------------------------------------------------------
// A strip of k_seam_march_v2 is 64 columns of which 10 are feeders (5 on each side: four sweeps and the residual eat one column per
// stage): 15.6 % of every load, store and vector instruction of the pass works on a column another strip owns -- most of the 1.15 x
// the pass moves over its compulsory bytes.  Here a lane owns columns g and g + 64 of a 128-column strip: 118 owned of 128 (7.8 %).
// Every access is still a wave-wide run of 64 consecutive 8-byte words (rows of a 2^k + 1 wide array are only 8-byte aligned; the
// 16-byte form of round 2, k_smooth2_march2, needed a lane shift per row and was slower).  The x-neighbours of the two halves are two
// wave rotations and two wave shifts whose vacant lane takes the OTHER half's rotated value (lane 63 of the lower half <- lane 0 of
// the upper half and vice versa): the same four DPP moves per double that two separate strips need.  Same operations on the same
// operands per point as the second version: bit-identical fields; the norm is summed in another order (other tiling).
template <int NC>
__device__ __forceinline__ void fpr_nbr_lr(const double (&m)[NC], double (&L)[NC], double (&R)[NC])
{
    if constexpr (NC == 1) {
        L[0] = fpr_lane_up1z(m[0]);
        R[0] = fpr_lane_down1z(m[0]);
    } else {
        int lo0 = __double2loint(m[0]), hi0 = __double2hiint(m[0]);
        int lo1 = __double2loint(m[1]), hi1 = __double2hiint(m[1]);
        // lower half, left neighbour: lane i <- lane i - 1, lane 0 <- lane 63 (a feeder lane: any finite value)
        const int l0lo = __builtin_amdgcn_update_dpp(0, lo0, 0x13C, 0xf, 0xf, true), l0hi = __builtin_amdgcn_update_dpp(0, hi0, 0x13C, 0xf, 0xf, true);
        // upper half, left neighbour: lane i <- lane i - 1, lane 0 <- lane 63 of the LOWER half (column 63 is the left neighbour of column 64)
        const int l1lo = __builtin_amdgcn_update_dpp(l0lo, lo1, 0x138, 0xf, 0xf, false), l1hi = __builtin_amdgcn_update_dpp(l0hi, hi1, 0x138, 0xf, 0xf, false);
        // upper half, right neighbour: lane i <- lane i + 1, lane 63 <- lane 0 (feeder)
        const int r1lo = __builtin_amdgcn_update_dpp(0, lo1, 0x134, 0xf, 0xf, true), r1hi = __builtin_amdgcn_update_dpp(0, hi1, 0x134, 0xf, 0xf, true);
        // lower half, right neighbour: lane i <- lane i + 1, lane 63 <- lane 0 of the UPPER half
        const int r0lo = __builtin_amdgcn_update_dpp(r1lo, lo0, 0x130, 0xf, 0xf, false), r0hi = __builtin_amdgcn_update_dpp(r1hi, hi0, 0x130, 0xf, 0xf, false);
        L[0] = __hiloint2double(l0hi, l0lo); L[1] = __hiloint2double(l1hi, l1lo);
        R[0] = __hiloint2double(r0hi, r0lo); R[1] = __hiloint2double(r1hi, r1lo);
    }
}

template <bool BCS, int PF = 4, int NC = 2>
__global__ __launch_bounds__(256) void k_seam_march_v3(const double* __restrict__ uin, const double* __restrict__ f,
                                                        double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                        double fac, int rows_per_chunk, int nstrips,
                                                        double* __restrict__ partials, const double* __restrict__ corr_c,
                                                        double* __restrict__ res_c_out, double* __restrict__ corr_c_out,
                                                        const int* __restrict__ skip)
{
    if (skip && *skip) return;
    __shared__ double red[16];
    constexpr int HX = 5;                                    // feeder columns on each side of a strip
    constexpr int SW = 64 * NC - 2 * HX;                     // columns owned by a strip
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + w;
    const bool active = strip < nstrips;
    const int y0 = blockIdx.y * rows_per_chunk;              // host: rows_per_chunk is even
    const int y1 = (y0 + rows_per_chunk < ny) ? y0 + rows_per_chunk : ny;  // output rows [y0, y1)
    const int rs = y0 - 6 < 0 ? 0 : y0 - 6;                  // EVEN: row rs + T has the parity of T
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    if (active) {
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        int gi[NC];
        bool col_bnd[NC], owner[NC], cint_col[NC];
        double wE[NC], wO[NC], facL[NC];
        unsigned vcl[NC], vch[NC], vld[NC], vst[NC], vstc[NC], vstr[NC], vstn[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            gi[c] = strip * SW - HX + lane + 64 * c;               // global column of this lane's column c
            const bool col_ok = gi[c] >= 0 && gi[c] < nx;
            const int gic = gi[c] < 0 ? 0 : (gi[c] > nx - 1 ? nx - 1 : gi[c]);  // clamped for loads
            col_bnd[c] = gi[c] <= 0 || gi[c] >= nx - 1;            // domain boundary column (or outside)
            const int sl = lane + 64 * c;                          // position inside the strip
            owner[c] = col_ok && sl >= HX && sl < 64 * NC - HX;
            int gis = gic;  // Neumann columns of the prolongated correction (part2_utils.jl:35-39)
            if (BCS) gis = (gic == 0) ? 1 : (gic == nx - 1 ? nx - 2 : gic);
            const int p_io = gis & 1, p_icl = gis >> 1, p_ich = (p_icl + 1 < nxc) ? p_icl + 1 : nxc - 1;
            const bool p_sx0 = p_icl >= 1 && p_icl <= nxc - 2, p_sx1 = p_io && (p_icl + 1 <= nxc - 2);
            const bool p_inx = gis >= 1 && gis <= nx - 2;
            wE[c] = p_io ? 0.5 : 1.0; wO[c] = p_io ? 0.25 : 0.5;   // prolong_bf's weight in an even / odd fine row
            vcl[c] = (p_inx && p_sx0) ? (unsigned)p_icl * 8u : FPR_OOR;   // excluded terms read 0
            vch[c] = (p_inx && p_sx1) ? (unsigned)p_ich * 8u : FPR_OOR;
            vld[c] = (unsigned)gic * 8u;
            vst[c] = owner[c] ? (unsigned)gi[c] * 8u : FPR_OOR;   // columns that are not owned store out of range
            vstc[c] = (owner[c] && !(gi[c] & 1)) ? (unsigned)(gi[c] >> 1) * 8u : FPR_OOR;
            // BCS: Neumann columns of the coarse right-hand side (:355-357), see k_smooth2_march
            vstr[c] = (BCS && (gi[c] == 0 || gi[c] == nx - 1)) ? FPR_OOR : vstc[c];
            vstn[c] = (BCS && owner[c] && (gi[c] == 2 || gi[c] == nx - 3)) ? (gi[c] == 2 ? 0u : (unsigned)(nxc - 1) * 8u) : FPR_OOR;
            facL[c] = col_bnd[c] ? 0.0 : fac;                      // a boundary column keeps its value: mid + 0 * rr
            cint_col[c] = (gi[c] >> 1) >= 1 && (gi[c] >> 1) <= nxc - 2;
        }
        const __amdgpu_buffer_rsrc_t rCor = fpr_rsrc(corr_c);
        const int crow = nxc * 8;
        double cs[NC][3][2];                                 // coarse rows j, j+1, j+2 (slot = (j - rs/2) mod 3): columns icl, ich
        auto ldc = [&](int slot, int j) {
            const int so = (j >= 1 && j <= nyc - 2) ? j * crow : (int)FPR_OOR;   // a boundary coarse row contributes nothing
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                cs[c][slot][0] = fpr_bld(rCor, vcl[c], so);
                cs[c][slot][1] = fpr_bld(rCor, vch[c], so);
            }
        };
        const int rowB = nx * 8;
        const __amdgpu_buffer_rsrc_t rUin = fpr_rsrc(uin + (size_t)nx * rs), rF = fpr_rsrc(f + (size_t)nx * rs);
        const __amdgpu_buffer_rsrc_t rUout = fpr_rsrc(uout + (size_t)nx * rs);
        // corrected input of row `row` = rs + LR (mod 12): u - P(corr), terms and order of prolong_bf
        auto ldu = [&](auto LRc, int row, double (&out)[NC]) {
            constexpr int LR = decltype(LRc)::value;
            constexpr int JR = LR >> 1;
            constexpr int SA = JR % 3, SB = (JR + 1) % 3, SN = (JR + 2) % 3;
            const int rc = row > ny - 1 ? ny - 1 : row;
            double v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = fpr_bld(rUin, vld[c], (rc - rs) * rowB);
            if constexpr ((LR & 1) == 0) ldc(SN, (row >> 1) + 2);   // used three rows from now; its slot held the row last used one row ago
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double pv;
                if constexpr ((LR & 1) == 0) {
                    pv = wE[c] * cs[c][SA][0];
                    pv = pv + wE[c] * cs[c][SA][1];
                } else {
                    pv = wO[c] * cs[c][SA][0];
                    pv = pv + wO[c] * cs[c][SA][1];
                    pv = pv + wO[c] * cs[c][SB][0];
                    pv = pv + wO[c] * cs[c][SB][1];
                }
                out[c] = v[c] - pv;
            }
        };
        auto ldf = [&](int r, double (&out)[NC]) {
            const int rc = r > ny - 1 ? ny - 1 : r;
#pragma unroll
            for (int c = 0; c < NC; ++c) out[c] = fpr_bld(rF, vld[c], (rc - rs) * rowB);
        };
        double wv[5][3][NC];
        double fw[6][NC];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < NC; ++c) wv[a][b][c] = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < NC; ++c) fw[a][c] = 0.0;
        using I0 = std::integral_constant<int, 0>;
        ldc(0, rs >> 1);
        ldc(1, (rs >> 1) + 1);
        ldu(I0{}, rs, wv[0][0]);
        ldf(rs, fw[0]);
        static_assert(PF == 4 || PF == 6 || PF == 12, "ring slots are compile-time constants of a loop unrolled by 12");
        double pu[PF][NC], pfv[PF][NC];
        ldu(std::integral_constant<int, 1>{}, rs + 1, pu[0]); ldf(rs + 1, pfv[0]);
        ldu(std::integral_constant<int, 2>{}, rs + 2, pu[1]); ldf(rs + 2, pfv[1]);
        ldu(std::integral_constant<int, 3>{}, rs + 3, pu[2]); ldf(rs + 3, pfv[2]);
        ldu(std::integral_constant<int, 4>{}, rs + 4, pu[3]); ldf(rs + 4, pfv[3]);
        if constexpr (PF >= 6) {
            ldu(std::integral_constant<int, 5>{}, rs + 5, pu[4]); ldf(rs + 5, pfv[4]);
            ldu(std::integral_constant<int, 6>{}, rs + 6, pu[5]); ldf(rs + 6, pfv[5]);
        }
        if constexpr (PF == 12) {      // (one wave per SIMD has the whole register file: twelve rows of both halves in flight)
            ldu(std::integral_constant<int, 7>{}, rs + 7, pu[6]); ldf(rs + 7, pfv[6]);
            ldu(std::integral_constant<int, 8>{}, rs + 8, pu[7]); ldf(rs + 8, pfv[7]);
            ldu(std::integral_constant<int, 9>{}, rs + 9, pu[8]); ldf(rs + 9, pfv[8]);
            ldu(std::integral_constant<int, 10>{}, rs + 10, pu[9]); ldf(rs + 10, pfv[9]);
            ldu(std::integral_constant<int, 11>{}, rs + 11, pu[10]); ldf(rs + 11, pfv[10]);
            ldu(std::integral_constant<int, 12>{}, rs + 12, pu[11]); ldf(rs + 12, pfv[11]);
        }
        const __amdgpu_buffer_rsrc_t rResC = fpr_rsrc(res_c_out), rCorC = fpr_rsrc(corr_c_out);
        // one Jacobi sweep at row j of a field whose rows j-1, j, j+1 are (lo, mid, hi); rr = residual used by the update
        auto sweep = [&](const double (&lo)[NC], const double (&mid)[NC], const double (&hi)[NC], const double (&fv)[NC], int j, double (&rr)[NC],
                         double (&un)[NC]) {
            double L[NC], R[NC];
            fpr_nbr_lr<NC>(mid, L, R);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                rr[c] = ((((R[c] + L[c]) + hi[c]) + lo[c]) - C * mid[c]) * _h2 - fv[c];
                un[c] = mid[c] + facL[c] * rr[c];
            }
            if (j <= 0 || j >= ny - 1) {   // uniform: first / last row of the grid
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < NC; ++c) un[c] = mid[c];
            }
        };
        auto step = [&](auto Tc, int r) {
            constexpr int T = decltype(Tc)::value;               // (r - rs) mod 12
            constexpr int Q = T % PF, M = T % 3, F = T % 6;
            constexpr int M1 = (M + 1) % 3, M2 = (M + 2) % 3;    // slots of rows r-2 (= r+1) and r-1
            auto fs = [](int k) { return (F - k + 6) % 6; };     // slot of f row r-k
            double an[NC], fn[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                asm volatile("v_mov_b64 %0, %1" : "=v"(an[c]) : "v"(pu[Q][c]));
                asm volatile("v_mov_b64 %0, %1" : "=v"(fn[c]) : "v"(pfv[Q][c]));
            }
            ldu(std::integral_constant<int, T + 1 + PF>{}, r + 1 + PF, pu[Q]);   // issue the loads of row r+1+PF
            ldf(r + 1 + PF, pfv[Q]);
            double rr[NC];
            // ---- cycle k, post-smoothing (:142-143): sweeps 1 and 2 at rows r-1, r-2 ----
            sweep(wv[0][M1], wv[0][M2], wv[0][M], fw[fs(1)], r - 1, rr, wv[1][M2]);
            const int j2 = r - 2;
            double u2[NC];
            sweep(wv[1][M], wv[1][M1], wv[1][M2], fw[fs(2)], j2, rr, u2);   // u at the end of cycle k (not stored)
            if constexpr (BCS) {
                // apply_boundary_conditions! between the cycles (multigrid.jl:60-62): Neumann columns copy their inner neighbour
                double fromL[NC], fromR[NC];
                fpr_nbr_lr<NC>(u2, fromL, fromR);
#pragma unroll
                for (int c = 0; c < NC; ++c) u2[c] = (gi[c] == 0) ? fromR[c] : ((gi[c] == nx - 1) ? fromL[c] : u2[c]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) wv[2][M1][c] = u2[c];
            if (j2 >= y0 && j2 < y1 && j2 > 0 && j2 < ny - 1) {   // uniform; columns that are not owned are dropped at the end
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = acc[c] + rr[c] * rr[c];   // r_rms of cycle k (:252)
            }
            // ---- cycle k+1, pre-smoothing (:124-125): sweeps 3 and 4 at rows r-3, r-4 ----
            sweep(wv[2][M2], wv[2][M], wv[2][M1], fw[fs(3)], r - 3, rr, wv[3][M]);
            const int j4 = r - 4;
            double u4[NC];
            sweep(wv[3][M1], wv[3][M2], wv[3][M], fw[fs(4)], j4, rr, u4);
            {
                const bool row_own = j4 >= y0 && j4 < y1;      // uniform
#pragma unroll
                for (int c = 0; c < NC; ++c) fpr_bst(rUout, vst[c], row_own ? (j4 - rs) * rowB : (int)FPR_OOR, u4[c]);   // unconditional (see FPR_OOR)
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) wv[4][M2][c] = u4[c];
            // ---- residual of the pre-smoothed field at row r-5 (even in every odd step), injected at even columns (:128-132) ----
            if constexpr ((T & 1) == 1) {
                const int j5 = r - 5;
                double L[NC], R[NC];
                fpr_nbr_lr<NC>(wv[4][M1], L, R);
                const int jc = j5 >> 1;
                const bool row_inj = j5 >= y0 && j5 < y1;       // uniform
                const int sc = row_inj ? jc * crow : (int)FPR_OOR;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double mid = wv[4][M1][c];
                    const double rres = ((((R[c] + L[c]) + wv[4][M2][c]) + wv[4][M][c]) - C * mid) * _h2 - fw[fs(5)][c];
                    const bool cint = cint_col[c] && jc >= 1 && jc <= nyc - 2;
                    fpr_bst(rResC, vstr[c], sc, cint ? rres : 0.0);
                    if constexpr (BCS) fpr_bst(rResC, vstn[c], sc, cint ? rres : 0.0);
                    fpr_bst(rCorC, vstc[c], sc, 0.0);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                wv[0][M1][c] = an[c];            // row r+1 takes the slot of row r-2
                fw[(F + 1) % 6][c] = fn[c];      // row r+1 takes the slot of row r-5
            }
        };
        const int rend = y1 + 4;
        int r = rs;
#define FPR_SEAM_STEP(T) step(std::integral_constant<int, T>{}, r + T)
        for (; r + 11 <= rend; r += 12) {
            FPR_SEAM_STEP(0); FPR_SEAM_STEP(1); FPR_SEAM_STEP(2); FPR_SEAM_STEP(3); FPR_SEAM_STEP(4); FPR_SEAM_STEP(5);
            FPR_SEAM_STEP(6); FPR_SEAM_STEP(7); FPR_SEAM_STEP(8); FPR_SEAM_STEP(9); FPR_SEAM_STEP(10); FPR_SEAM_STEP(11);
        }
#undef FPR_SEAM_STEP
#define FPR_SEAM_TAIL(T) if (r <= rend) { step(std::integral_constant<int, T>{}, r); ++r; }
        FPR_SEAM_TAIL(0) FPR_SEAM_TAIL(1) FPR_SEAM_TAIL(2) FPR_SEAM_TAIL(3) FPR_SEAM_TAIL(4) FPR_SEAM_TAIL(5)
        FPR_SEAM_TAIL(6) FPR_SEAM_TAIL(7) FPR_SEAM_TAIL(8) FPR_SEAM_TAIL(9) FPR_SEAM_TAIL(10)
#undef FPR_SEAM_TAIL
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = (owner[c] && !col_bnd[c]) ? acc[c] : 0.0;
    }
    double tot = acc[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) tot = tot + acc[c];
    const double sblk = fpr_block_sum<256>(tot, red);
    if (threadIdx.x == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
}

// ---- k_smooth2_march, second version (round 3): what k_seam_march_v2 does for the seam, for the two-sweep passes ------------
// Same fields as k_smooth2_march (same operations on the same operands per point).  Chunks start on an even row, so row
// parity is a compile-time constant of the loop unrolled by 12: the prolongation has two terms in even rows and four in odd
// ones, its coarse rows come through a buffer descriptor with the exclusion masks folded into the offsets (an excluded term
// reads 0) into a ring of three register pairs three rows ahead of their use; windows live in compile-time slots (no
// shifting of nine window rows per step); boundary columns keep their value through a per-lane factor, boundary rows
// through a uniform branch; the residual stage runs in the injected rows only; the norm is masked per lane once at the end.
// FSQ: the pass also leaves sum(f.^2) of its rows as block partials (the first pass of a solve: f_rms of multigrid.jl:53 without a pass of its own)
template <bool NORM, bool PROLONG, bool RESTRICT, bool FSQ = false>
__global__ __launch_bounds__(256) void k_smooth2_march_v2(const double* __restrict__ uin, const double* __restrict__ f,
                                                           double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                           double fac, int rows_per_chunk, int nstrips,
                                                           double* __restrict__ partials, const double* __restrict__ corr_c,
                                                           int apply_BCs, double* __restrict__ res_c_out,
                                                           double* __restrict__ corr_c_out, const int* __restrict__ skip,
                                                           FprFinishArgs fin)
{
    __shared__ double red[16];
    if (fin.partials && blockIdx.y == gridDim.y - 1) {
        // the extra workgroup row of a pass that also carries the finish of the cycle BEFORE it (fprx_cycle_finish_defer): norm, exit
        // test and record of cycle k while this pass -- the first of cycle k+1 below the finest level -- runs; the launches behind it
        // find `stop` as they would behind k_cycle_finish.  The other workgroups of THIS launch may or may not see it: what they
        // write is scratch of cycle k+1, which nothing reads once the loop has ended.
        if (blockIdx.x == 0) fpr_cycle_finish_body(fin, red);
        return;
    }
    if (skip && *skip) return;   // a cycle enqueued ahead of the exit test that ended the loop (FprCycleCtl)
    constexpr int HX = RESTRICT ? 3 : 2;                     // feeder lanes on each side of a strip
    constexpr int SW = 64 - 2 * HX;                          // columns owned by a strip
    const bool uin_zero = (apply_BCs & 512) != 0;            // bit 9: the input field is identically zero (see vcycle_level)
    apply_BCs &= 255;                                        // (bit 8, non-temporal stores, is ignored by this kernel)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + w;
    const bool active = strip < nstrips;
    const int gi = strip * SW - HX + lane;                   // global column of this lane
    const bool col_ok = active && gi >= 0 && gi < nx;
    const int gic = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);  // clamped for loads
    const bool col_bnd = gi <= 0 || gi >= nx - 1;            // domain boundary column (or outside)
    const bool owner = col_ok && lane >= HX && lane < 64 - HX;
    const int y0 = blockIdx.y * rows_per_chunk;              // host: rows_per_chunk is even
    const int y1 = (y0 + rows_per_chunk < ny) ? y0 + rows_per_chunk : ny;  // output rows [y0, y1)
    constexpr int HY = RESTRICT ? 4 : 2;                     // rows above the chunk (even: row rs + T has the parity of T)
    const int rs = y0 - HY < 0 ? 0 : y0 - HY;
    static_assert(!(NORM && FSQ), "one list of block partials");
    double acc = 0.0, accf = 0.0;
    if (active) {
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        int gis = gic;  // Neumann columns of the prolongated correction (part2_utils.jl:35-39)
        if (PROLONG && apply_BCs) gis = (gic == 0) ? 1 : (gic == nx - 1 ? nx - 2 : gic);
        const int p_io = gis & 1, p_icl = gis >> 1, p_ich = (p_icl + 1 < nxc) ? p_icl + 1 : nxc - 1;
        const bool p_sx0 = p_icl >= 1 && p_icl <= nxc - 2, p_sx1 = p_io && (p_icl + 1 <= nxc - 2);
        const bool p_inx = gis >= 1 && gis <= nx - 2;
        const double wE = p_io ? 0.5 : 1.0, wO = p_io ? 0.25 : 0.5;   // prolong_bf's weight in an even / odd fine row
        const unsigned vcl = (PROLONG && p_inx && p_sx0) ? (unsigned)p_icl * 8u : FPR_OOR;   // excluded terms read 0
        const unsigned vch = (PROLONG && p_inx && p_sx1) ? (unsigned)p_ich * 8u : FPR_OOR;
        const __amdgpu_buffer_rsrc_t rCor = fpr_rsrc(PROLONG ? (const void*)corr_c : (const void*)uin);
        const int crow = nxc * 8;
        double cs[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};   // coarse rows j, j+1, j+2 (slot = (j - rs/2) mod 3): columns icl, ich
        auto ldc = [&](int slot, int j) {
            if constexpr (PROLONG) {
                const int so = (j >= 1 && j <= nyc - 2) ? j * crow : (int)FPR_OOR;   // a boundary coarse row contributes nothing
                cs[slot][0] = fpr_bld(rCor, vcl, so);
                cs[slot][1] = fpr_bld(rCor, vch, so);
            }
        };
        // rows are addressed relative to the first row of the chunk: (rows_per_chunk + 8) * nx * 8 < 2^31
        const int rowB = nx * 8;
        // A coarse level starts from the zero guess its parent has just written (`corr_c .= 0`, multigrid.jl:132): with uin_zero the
        // descriptor of the input has no records, every load of it returns 0 without touching memory -- the same zeros, the same
        // operations on them, 8 bytes per point less to read (33.6 MB at 2049^2)
        const __amdgpu_buffer_rsrc_t rUin = uin_zero ? __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(uin), 0, 0, 0x00020000)
                                                     : fpr_rsrc(uin + (size_t)nx * rs);
        const __amdgpu_buffer_rsrc_t rF = fpr_rsrc(f + (size_t)nx * rs);
        const __amdgpu_buffer_rsrc_t rUout = fpr_rsrc(uout + (size_t)nx * rs);
        const unsigned vld = (unsigned)gic * 8u;
        const unsigned vst = owner ? (unsigned)gi * 8u : FPR_OOR;   // lanes that own nothing store out of range
        auto ldu = [&](auto LRc, int row) {
            constexpr int LR = decltype(LRc)::value;             // (row - rs) mod 12
            constexpr int JR = LR >> 1;
            constexpr int SA = JR % 3, SB = (JR + 1) % 3, SN = (JR + 2) % 3;
            const int rc = row > ny - 1 ? ny - 1 : row;
            const double v = fpr_bld(rUin, vld, (rc - rs) * rowB);
            if constexpr (PROLONG) {
                double pv;
                if constexpr ((LR & 1) == 0) {
                    ldc(SN, (row >> 1) + 2);       // used three rows from now; its slot held the row last used one row ago
                    pv = wE * cs[SA][0];
                    pv = pv + wE * cs[SA][1];
                } else {
                    pv = wO * cs[SA][0];
                    pv = pv + wO * cs[SA][1];
                    pv = pv + wO * cs[SB][0];
                    pv = pv + wO * cs[SB][1];
                }
                return v - pv;
            } else {
                return v;
            }
        };
        auto ldf = [&](int r) { const int rc = r > ny - 1 ? ny - 1 : r; return fpr_bld(rF, vld, (rc - rs) * rowB); };
        // windows in compile-time slots: row j of u (wa), of the once-smoothed field (wb), of the twice-smoothed field (wc,
        // RESTRICT) lives in slot (j - rs) mod 3, row j of f in slot (j - rs) mod 4
        double wa[3] = {0.0, 0.0, 0.0}, wb[3] = {0.0, 0.0, 0.0}, wc[3] = {0.0, 0.0, 0.0}, fw[4] = {0.0, 0.0, 0.0, 0.0};
        ldc(0, rs >> 1);
        ldc(1, (rs >> 1) + 1);
        wa[0] = ldu(std::integral_constant<int, 0>{}, rs);
        fw[0] = ldf(rs);
        constexpr int PF = 4;
        double pu[PF], pfv[PF];
        pu[0] = ldu(std::integral_constant<int, 1>{}, rs + 1); pfv[0] = ldf(rs + 1);
        pu[1] = ldu(std::integral_constant<int, 2>{}, rs + 2); pfv[1] = ldf(rs + 2);
        pu[2] = ldu(std::integral_constant<int, 3>{}, rs + 3); pfv[2] = ldf(rs + 3);
        pu[3] = ldu(std::integral_constant<int, 4>{}, rs + 4); pfv[3] = ldf(rs + 4);
        // RESTRICT: coarse arrays (whole-array descriptors: nxc * nyc * 8 < 2^31), even owned columns only
        const __amdgpu_buffer_rsrc_t rResC = fpr_rsrc(RESTRICT ? (const void*)res_c_out : (const void*)uout);
        const __amdgpu_buffer_rsrc_t rCorC = fpr_rsrc(RESTRICT ? (const void*)corr_c_out : (const void*)uout);
        const unsigned vstc = (RESTRICT && owner && !(gi & 1)) ? (unsigned)(gi >> 1) * 8u : FPR_OOR;
        // RESTRICT with apply_BCs: Neumann columns of the coarse right-hand side (:355-357), see k_smooth2_march
        const bool nbc = RESTRICT && apply_BCs != 0;
        const unsigned vstr = (nbc && (gi == 0 || gi == nx - 1)) ? FPR_OOR : vstc;
        const unsigned vstn = (nbc && owner && (gi == 2 || gi == nx - 3)) ? (gi == 2 ? 0u : (unsigned)(nxc - 1) * 8u) : FPR_OOR;
        const double facL = col_bnd ? 0.0 : fac;             // a boundary column keeps its value: mid + 0 * rr
        const bool cint_col = (gi >> 1) >= 1 && (gi >> 1) <= nxc - 2;
        auto sweep = [&](double lo, double mid, double hi, double fv, int j, double& rr) {
            const double L = fpr_lane_up1z(mid), R = fpr_lane_down1z(mid);
            rr = ((((R + L) + hi) + lo) - C * mid) * _h2 - fv;
            double un = mid + facL * rr;
            if (j <= 0 || j >= ny - 1) { asm volatile("" ::: "memory"); un = mid; }   // uniform: first / last row of the grid
            return un;
        };
        auto step = [&](auto Tc, int r) {
            constexpr int T = decltype(Tc)::value;               // (r - rs) mod 12
            constexpr int Q = T % 4, M = T % 3, F = T % 4;
            constexpr int M1 = (M + 1) % 3, M2 = (M + 2) % 3;    // slots of rows r-2 (= r+1) and r-1
            auto fs = [](int k) { return (F - k + 4) % 4; };     // slot of f row r-k
            double an, fn;
            asm volatile("v_mov_b64 %0, %1" : "=v"(an) : "v"(pu[Q]));
            asm volatile("v_mov_b64 %0, %1" : "=v"(fn) : "v"(pfv[Q]));
            pu[Q] = ldu(std::integral_constant<int, T + 1 + PF>{}, r + 1 + PF);   // issue the loads of row r+1+PF
            pfv[Q] = ldf(r + 1 + PF);
            double rr;
            // ---- sweep 1 at row r-1 (u rows r-2, r-1, r) ----
            wb[M2] = sweep(wa[M1], wa[M2], wa[M], fw[fs(1)], r - 1, rr);
            // ---- sweep 2 at row r-2 (once-smoothed rows r-3, r-2, r-1) ----
            const int j2 = r - 2;
            const double u2 = sweep(wb[M], wb[M1], wb[M2], fw[fs(2)], j2, rr);
            {
                const bool row_own = j2 >= y0 && j2 < y1;        // uniform
                fpr_bst(rUout, vst, row_own ? (j2 - rs) * rowB : (int)FPR_OOR, u2);   // unconditional (see FPR_OOR)
            }
            if constexpr (NORM) {
                if (j2 >= y0 && j2 < y1 && j2 > 0 && j2 < ny - 1) {   // uniform; lanes that own nothing are dropped at the end
                    asm volatile("" ::: "memory");
                    acc = acc + rr * rr;
                }
            }
            if constexpr (FSQ) {
                if (j2 >= y0 && j2 < y1) {   // uniform: every row of the grid once, boundary rows and columns included (sum(f.^2), :53)
                    const double fv2 = fw[fs(2)];
                    accf = accf + fv2 * fv2;
                }
            }
            if constexpr (RESTRICT) {
                wc[M1] = u2;
                // ---- residual of the twice-smoothed field at row r-3 (even in every odd step), injected at even columns ----
                if constexpr ((T & 1) == 1) {
                    const int j3 = r - 3;
                    const double mid = wc[M];
                    const double L = fpr_lane_up1z(mid), R = fpr_lane_down1z(mid);
                    const double rres = ((((R + L) + wc[M1]) + wc[M2]) - C * mid) * _h2 - fw[fs(3)];
                    const int jc = j3 >> 1;
                    const bool cint = cint_col && jc >= 1 && jc <= nyc - 2;
                    const bool row_inj = j3 >= y0 && j3 < y1;   // uniform
                    const int sc = row_inj ? jc * crow : (int)FPR_OOR;
                    fpr_bst(rResC, vstr, sc, cint ? rres : 0.0);
                    fpr_bst(rResC, vstn, sc, cint ? rres : 0.0);
                    fpr_bst(rCorC, vstc, sc, 0.0);
                }
            }
            wa[M1] = an;               // row r+1 takes the slot of row r-2
            fw[(F + 1) % 4] = fn;      // row r+1 takes the slot of row r-3
        };
        const int rend = y1 + (RESTRICT ? 2 : 1);
        int r = rs;
#define FPR_M2_STEP(T) step(std::integral_constant<int, T>{}, r + T)
        for (; r + 11 <= rend; r += 12) {
            FPR_M2_STEP(0); FPR_M2_STEP(1); FPR_M2_STEP(2); FPR_M2_STEP(3); FPR_M2_STEP(4); FPR_M2_STEP(5);
            FPR_M2_STEP(6); FPR_M2_STEP(7); FPR_M2_STEP(8); FPR_M2_STEP(9); FPR_M2_STEP(10); FPR_M2_STEP(11);
        }
#undef FPR_M2_STEP
#define FPR_M2_TAIL(T) if (r <= rend) { step(std::integral_constant<int, T>{}, r); ++r; }
        FPR_M2_TAIL(0) FPR_M2_TAIL(1) FPR_M2_TAIL(2) FPR_M2_TAIL(3) FPR_M2_TAIL(4) FPR_M2_TAIL(5)
        FPR_M2_TAIL(6) FPR_M2_TAIL(7) FPR_M2_TAIL(8) FPR_M2_TAIL(9) FPR_M2_TAIL(10)
#undef FPR_M2_TAIL
        acc = (owner && !col_bnd) ? acc : 0.0;
        accf = owner ? accf : 0.0;
    }
    if constexpr (NORM) {
        const double sblk = fpr_block_sum<256>(acc, red);
        if (threadIdx.x == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
    }
    if constexpr (FSQ) {
        const double sblk = fpr_block_sum<256>(accf, red);
        if (threadIdx.x == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
    }
}
