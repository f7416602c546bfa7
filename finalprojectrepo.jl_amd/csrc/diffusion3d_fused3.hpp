// diffusion3d_fused3.hpp -- THREE pseudo-transient iterations of the fused 7-point update in one pass over memory
// (temporal blocking of scripts-part1/part1_kernel_programming.jl:179-192: three trips through the `while` body,
// step(A -> B), swap, step(B -> A), swap, step(A -> B), without materialising the two fields in between).
//
// Per launch and interior cell the kernel reads the field and Ht once and writes the field after three iterations and
// the residual of the third: 32 B for three iterations.  Results are bit-identical to three launches of k_diff3_march
// (same diff3_point expression on the same operands).
//
// Levels: L0 = the field in buffer X, L1 / L2 = the fields after one / two iterations (never leave the chip), L3 = the
// field after three, written to the interior of buffer Y; dH receives the residual of the third iteration.  The reference
// never writes the boundary cells of its two ping-pong buffers, so each keeps its own for ever: L1 and L3 live in the
// reference's OTHER buffer -- L1's boundary cells are read from `Bnd` (any array that carries that buffer's boundary; Y
// itself does, and only its boundary cells are read), and Y keeps them for L3 -- while L2 lives in X's buffer again and its
// boundary cells are X's own, already in registers.  An odd depth thus needs no third work buffer: X and Y alternate
// exactly as Htau and Htau2 do in the reference, three iterations at a time.
//
// Geometry (wave64, 2 cells per lane, RY = 3 rows per lane, 8 waves stacked in y = block tile 128 x 24):
//   * the block marches in z; iteration m computes L1 on plane m for the whole tile (ring of 4 register planes of L0:
//     m-1, m, m+1 and m+2 in flight), L2 on plane m-1 from the L1 planes m-2, m-1, m, and L3 on plane m-2 from the L2
//     planes m-3, m-2, m-1; Ht has a ring of 4 (every level needs it; plane m+1 in flight); all rings are indexed at
//     compile time (loop unrolled by 4; an L1 / L2 ring holds three live planes);
//   * x-neighbours by DPP wave shifts at every level; y-neighbours between the waves through ONE LDS exchange + raw
//     s_barrier per iteration that carries the first / last rows of L0(m), L1(m-1) and L2(m-2);
//   * L1 is valid on the whole tile (L0 halo cells / rows come from global memory), L2 on the tile shrunk by one cell, L3
//     by two: tiles overlap by 4 in x and y, chunks by 4 planes in z;
//   * 260 (tile, chunk) units at 512^3 on 256 compute units: the launch has one workgroup per unit of the first round and
//     slices the left-over units thinly over all of them (the reserved form of k_diff3_march2 without tickets).
// Requirements (the caller falls back to shallower launches otherwise): nx even and >= 128, ny >= 24, nz >= 5, all arrays
// 16-byte aligned, (zc + 12) planes below 2 GiB.
#pragma once
#include "diffusion3d_fused2.hpp"

#ifndef DIFF3_M3_ROW_FENCE
#define DIFF3_M3_ROW_FENCE 0      // 1: scheduling barrier between the rows of a level (no effect on the register count: harness only)
#endif
// When the next planes are requested.  The register budget (256 at two waves per SIMD) holds twelve planes of 12 registers beside the
// temporaries of the point update, and L0 + Ht + L1 + L2 need three each: a plane in flight beside them is a thirteenth, which the
// compiler spills -- with a wait for the load it has just issued (measured: 1.165 ms per launch, no overlap of loads and arithmetic).
//   0 = early: L0 plane m+3 behind the second step, Ht plane m+2 at the end of iteration m (two planes in flight all the time)
//   1 = late:  L0 plane m+2 and Ht plane m+1 behind the second step of iteration m, into the slots that died one iteration earlier; they
//       are needed at the first step of iteration m+1 (the third step and the exchange lie between); twelve planes at any time
#ifndef DIFF3_M3_SCHED
#define DIFF3_M3_SCHED 0
#endif
// 1: the L2 plane below the one the third step updates (its z-minus operand, idle for a whole iteration between its use as the centre
// and this one) waits in LDS instead of registers: 3 ds_write_b128 + 3 ds_read_b128 per wave and iteration buy the twelve registers
// the budget is short of
#ifndef DIFF3_M3_PARK
#define DIFF3_M3_PARK 0
#endif
// 1: the three running sums of squares (one per level and lane) live in LDS -- one ds_add_f64 per level and iteration -- instead of six
// registers
#ifndef DIFF3_M3_LDS_ACC
#define DIFF3_M3_LDS_ACC 1
#endif

struct Diff3Args3 {
    const double* Ht;
    const double* X;        // L0
    const double* Bnd;      // boundary cells of L1 (the reference's other buffer; may be Y)
    double* Y;              // L3 (interior of the box)
    double* dH;             // residual of the third step (nullptr: not stored -- its norm is still reduced)
    int nx, ny, nz;
    int lo[3], hi[3];       // output box, clipped to the interior [1, n-1)
    double dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
    double scale;
    double* partials1;      // per-unit sum((r*scale)^2) over owned cells, first / second / third step
    double* partials2;
    double* partials3;
    int zc, ntx, nby, ntz, sx;
    int xalign, xcd_remap;
    int nfull, bal_sp, bal_q;   // workgroups that serve a whole unit; slices per left-over unit, planes per slice
    const int* skip;        // return at once if *skip (nullptr = unconditional)
    int lane_off;
#ifdef FPR_TUNE
    int dbg;
#endif
};

template <bool NORM, bool WRES = true>
__global__ __launch_bounds__(512, 1) void k_diff3_march3(Diff3Args3 a)
{
    constexpr int NW = 8, VX = 2, RY = 3, TXW = 128, SYB = NW * RY - 4;
    constexpr int NR = 4;                         // ring length = loop unroll
    constexpr int SLOT = 6 * TXW;                 // doubles per wave slot: first / last row of L0, L1, L2
    constexpr unsigned OOR = 0x7fffffffu;
#if DIFF3_M3_LDS_ACC
    __shared__ double accl[3][64 * NW];     // per level and lane: running sum of squares over the rows / planes the workgroup owns
#endif
    __shared__ double red[3 * NW];          // per wave and level: sum over the wave's owned cells of every item served so far
    extern __shared__ __attribute__((aligned(16))) double xrow[];   // [parity][slot 0..NW+1][row kind][TXW]: 2 * (NW + 2) * SLOT doubles

    if (a.skip && *a.skip) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    if (NORM && tid < 3 * NW) red[tid] = 0.0;    // (ordered before its first use by the barriers of the march)
#if DIFF3_M3_LDS_ACC
    if (NORM) { accl[0][tid] = 0.0; accl[1][tid] = 0.0; accl[2][tid] = 0.0; }     // (a lane's own slots: program order suffices)
#endif

    // Units: (x/y tile, z-chunk) pairs.  The first a.nfull workgroups serve one unit each -- as many as the device holds at once -- and
    // the units beyond them are cut into a.bal_sp thin slices of a.bal_q planes, one workgroup per slice, which the dispatcher places as
    // the whole units finish: 260 units on 256 compute units cost one round and a few iterations, not two rounds.
    const int unit = blockIdx.x;
    int tx, by, k0, k1;
    int bid = unit, slice = -1;
    if (bid >= a.nfull) {
        const int j = (bid - a.nfull) / a.bal_sp;
        slice = (bid - a.nfull) - j * a.bal_sp;
        bid = a.nfull + j;
    } else if (a.xcd_remap == 1) {
        const int q = a.nfull >> 3, rem = a.nfull & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        bid = xcd * q + (xcd < rem ? xcd : rem) + slot;
    }
    tx = bid % a.ntx;
    by = (bid / a.ntx) % a.nby;
    const int tz = bid / (a.ntx * a.nby);
    k0 = a.lo[2] + tz * a.zc;
    k1 = (k0 + a.zc < a.hi[2]) ? k0 + a.zc : a.hi[2];
    if (slice >= 0) {
        k0 += slice * a.bal_q;
        k1 = (k0 + a.bal_q < k1) ? k0 + a.bal_q : k1;
    }
    const bool idle = k1 <= k0;              // block-uniform: a slice beyond the end of its unit (its partial sums are zero)

    // ---- x: owned output cells [ol, oh); own cells [s, s+128) ----
    const int e0 = a.lo[0] & ~1;
    auto cut = [&](int t) {
        const int c = e0 + t * a.sx;
        return a.xalign ? (c & ~15) + 8 : c;
    };
    const int olr = tx == 0 ? a.lo[0] : cut(tx), ohr = tx == a.ntx - 1 ? a.hi[0] : cut(tx + 1);
    const int ol = olr > a.lo[0] ? olr : a.lo[0];
    const int oh = ohr < a.hi[0] ? ohr : a.hi[0];
    int s = ol >= 2 ? (ol - 2) & ~1 : 0;
    s = s < nx - TXW ? s : nx - TXW;
    const int ib = s + lane * VX;
    int ilast = (oh + 2) & ~1;                       // last pair that is needed (L0 at oh+2)
    ilast = ilast < nx - 2 ? ilast : nx - 2;
    int ifirst = ol >= 3 ? (ol - 3) & ~1 : 0;        // first pair that is needed (L0 at ol-3)
    const int ibc = ib < ifirst ? ifirst : (ib < ilast ? ib : ilast);
    const unsigned voff = (unsigned)ibc * 8u;
    const bool xb_tile = (s == 0) || (s + TXW == nx);
    const bool bndL = (ib == 0);
    const bool bndR = (ib + 1 == nx - 1);
    int ie = (lane == 0) ? ib - 1 : ib + VX;
    ie = ie < ifirst ? ifirst : (ie > ilast + 1 ? ilast + 1 : ie);
    const unsigned eoff = ((lane == 0 || lane == 63) && !(bndL || bndR)) ? (unsigned)ie * 8u : OOR;   // from X
    const unsigned boff = bndL ? 0u : (bndR ? (unsigned)(nx - 1) * 8u : OOR);                     // from Bnd

    // ---- y: owned rows [oly, ohy); block rows y1 .. y1 + 23 ----
    const int oly = a.lo[1] + by * SYB;
    const int ohy = (oly + SYB < a.hi[1]) ? oly + SYB : a.hi[1];
    int y1 = oly >= 2 ? oly - 2 : 0;
    y1 = y1 < ny - NW * RY ? y1 : ny - NW * RY;
    const int j0 = y1 + w * RY;
    const bool bb = (w == 0) && (y1 == 0);
    const bool bt = (w == NW - 1) && (y1 + NW * RY - 1 == ny - 1);
    const int jd = bb ? 0 : (j0 > 0 ? j0 - 1 : 0);
    const int ju = bt ? ny - 1 : (j0 + RY < ny - 1 ? j0 + RY : ny - 1);
    const bool hwave = (w == 0) || (w == NW - 1);
    const double* Hsrc = (w == 0) ? (bb ? a.Bnd : a.X) : (bt ? a.Bnd : a.X);
    const int hrow = (w == 0) ? jd : ju;

    // ---- z: owned planes [k0, k1); iterations m0 .. m1 ----
    const int m0 = k0 - 2, m1 = k1 + 1;

    bool cm[VX], rm[RY];
#pragma unroll
    for (int v = 0; v < VX; ++v) cm[v] = (ib + v >= ol) && (ib + v < oh);
#pragma unroll
    for (int r = 0; r < RY; ++r) rm[r] = (j0 + r >= oly) && (j0 + r < ohy);   // uniform
    const bool has_split = ((ol | oh) & 1) != 0;
    const unsigned sv4 = (cm[0] && cm[1]) ? (unsigned)ib * 8u : OOR;
    const unsigned sv2 = (cm[0] != cm[1]) ? (unsigned)(ib + (cm[1] ? 1 : 0)) * 8u : OOR;

    const Diff3Coef cf{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};
    auto kcl = [&](int k) { return k < 0 ? 0 : (k > nz - 1 ? nz - 1 : k); };

    // one sum per level and lane: a lane owns both cells of its pair or neither, except at an odd end of the owned range (tiles with
    // has_split: there the cell that is not owned is left out as it goes); lanes that own nothing are dropped at the end
#if !DIFF3_M3_LDS_ACC
    double acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#endif
    if (!idle && (!a.lane_off || (ib >= ifirst && ib <= ilast))) {
    // ---- buffer descriptors (see k_diff3_march2): every access of iteration m uses the scalar offset so + r * rs ----
    const int ps = (int)(sz * 8), rs = (int)(sy * 8);
    const int pbA = m0 - 1 > 0 ? m0 - 1 : 0;
    const long array_bytes = (long)sz * (long)nz * 8;
    auto rsrc_at = [&](const double* Xp, int pshift, int row) -> __amdgpu_buffer_rsrc_t {
        const long off = ((long)(pbA + pshift) * (long)sz + (long)row * (long)sy) * 8;
        long rem = array_bytes - off;
        rem = rem < 0 ? 0 : (rem > 0x7ffffff0L ? 0x7ffffff0L : rem);
        return diff3_rsrc((uintptr_t)Xp + (uintptr_t)off, (unsigned)rem);
    };
    const __amdgpu_buffer_rsrc_t rA = rsrc_at(a.X, 0, j0);        // L0 plane m+3
    const __amdgpu_buffer_rsrc_t rHt = rsrc_at(a.Ht, -1, j0);     // Ht plane m+2
    const __amdgpu_buffer_rsrc_t rEB = rsrc_at(a.Bnd, -2, j0);    // Bnd x-boundary cells, plane m+1
    const __amdgpu_buffer_rsrc_t rH = rsrc_at(Hsrc, -2, hrow);    // halo row, plane m+1
    const __amdgpu_buffer_rsrc_t rC = rsrc_at(a.Y, -5, j0);       // L3 plane m-2
    const __amdgpu_buffer_rsrc_t rD = rsrc_at(WRES ? a.dH : a.Y, -5, j0);

    DVec<VX> P[NR][RY];       // L0 planes m-1, m, m+1, m+2; plane p lives in slot (p - m0 + 1) % NR
    DVec<VX> HT[NR][RY];      // Ht planes m-2, m-1, m, m+1; plane p in slot (p - m0 + 2) % NR
    DVec<VX> Q1[NR][RY];      // L1 planes m-2, m-1, m; plane p in slot (p - m0 + 2) % NR
    DVec<VX> Q2[NR][RY];      // L2 planes m-3, m-2, m-1; plane p in slot (p - m0 + 3) % NR
    DVec<VX> YH;              // global L0 halo row of plane m (bottom / top wave; the Bnd boundary row if bb / bt)
    double ED[RY];            // L0 tile-edge cells of plane m (or the Bnd boundary cell)
#if DIFF3_M3_LDS_ACC
    double acc1, acc2, acc3;          // (unused names: the sums live in accl)
    auto flush = [&](int lv, double sl, double&) { atomicAdd(&accl[lv][tid], sl); };   // result unused: ds_add_f64
#else
    auto flush = [&](int, double sl, double& acc) { acc += sl; };
#endif
    // A lane owns both cells of its pair or neither (dropped at the end), except the lane that holds an x-boundary cell (the launch covers
    // the whole interior: an odd end of the owned range is the domain's): that cell's residual is zeroed in the x-boundary branch below
    auto accum = [&](double& acc, const double (&r)[VX]) {
        acc = __builtin_fma(r[0], r[0], acc);
        acc = __builtin_fma(r[1], r[1], acc);
    };
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int v = 0; v < VX; ++v) { Q1[q][r].v[v] = 0.0; Q2[q][r].v[v] = 0.0; HT[q][r].v[v] = 0.0; }

#if DIFF3_M3_PARK
    double* const park = xrow + (size_t)2 * (NW + 2) * SLOT + (size_t)w * (RY * TXW) + lane * VX;   // this wave's parked L2 plane
#endif
    auto row_off = [&](int soff, int r) { return soff == (int)OOR ? (int)OOR : soff + r * rs; };
    auto load_rows = [&](DVec<VX>(&dst)[RY], __amdgpu_buffer_rsrc_t rsrc, int soff) {
#pragma unroll
        for (int r = 0; r < RY; ++r) dst[r] = diff3_bld2(rsrc, voff, row_off(soff, r));
    };
    // soff: plane offset for the descriptors based two planes below rA (rEB, rH); soffA: the same plane relative to rA
    auto load_halo = [&](DVec<VX>& yh, double (&e)[RY], int soff, int soffA) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const double ea = diff3_bld1(rA, eoff, row_off(soffA, r));
            const double eb = diff3_bld1(rEB, boff, row_off(soff, r));
            e[r] = __longlong_as_double(__double_as_longlong(ea) | __double_as_longlong(eb));
        }
        yh = diff3_bld2(rH, voff, hwave ? soff : (int)OOR);
    };

    // slots: L0 plane m0-1 -> 0 ... m0+2 -> 3; Ht plane m0 -> 2, m0+1 -> 3 (m0-2, m0-1 are not needed)
    load_rows(P[0], rA, (kcl(m0 - 1) - pbA) * ps);
    load_rows(P[1], rA, (kcl(m0) - pbA) * ps);
    load_rows(P[2], rA, (m0 + 1 - pbA) * ps);
    load_rows(HT[2], rHt, (kcl(m0) - pbA + 1) * ps);
#if DIFF3_M3_SCHED == 0 || DIFF3_M3_SCHED == 2
    load_rows(P[3], rA, (m0 + 2 - pbA) * ps);
#endif
#if DIFF3_M3_SCHED == 0
    load_rows(HT[3], rHt, (m0 + 1 - pbA + 1) * ps);
#endif
    load_halo(YH, ED, (kcl(m0) - pbA + 2) * ps, (kcl(m0) - pbA) * ps);
    // scalar offset of iteration m: plane m+3 of rA = plane m+2 of rHt = plane m+1 of rEB / rH = plane m-2 of rC / rD
    int so = (m0 + 3 - pbA) * ps;

    auto step = [&](auto Sc, auto Do2c, auto Do3c, int m) {
        constexpr int S = decltype(Sc)::value;       // (m - m0) % NR
        constexpr bool DO2 = decltype(Do2c)::value;
        constexpr bool DO3 = decltype(Do3c)::value;
        DVec<VX>(&zmR)[RY] = P[S % NR];
        DVec<VX>(&cR)[RY] = P[(S + 1) % NR];
        DVec<VX>(&zpR)[RY] = P[(S + 2) % NR];
        DVec<VX>(&Q1m)[RY] = Q1[S % NR];
        DVec<VX>(&Q1c)[RY] = Q1[(S + 1) % NR];
        DVec<VX>(&Q1n)[RY] = Q1[(S + 2) % NR];
#if !DIFF3_M3_PARK
        DVec<VX>(&Q2m)[RY] = Q2[S % NR];
#endif
        DVec<VX>(&Q2c)[RY] = Q2[(S + 1) % NR];
        DVec<VX>(&Q2n)[RY] = Q2[(S + 2) % NR];

        // ---- one LDS exchange for all three levels: rows of L0(m), L1(m-1), L2(m-2) ----
        double* buf = xrow + (size_t)(S & 1) * ((NW + 2) * SLOT);     // (parity of the iteration count: a compile-time offset)
        typedef double d2l __attribute__((ext_vector_type(2)));
        {
            double* mine = buf + (size_t)(w + 1) * SLOT + lane * VX;
            d2l t;
            t.x = cR[0].v[0]; t.y = cR[0].v[1];             *reinterpret_cast<d2l*>(mine) = t;
            t.x = cR[RY - 1].v[0]; t.y = cR[RY - 1].v[1];   *reinterpret_cast<d2l*>(mine + TXW) = t;
            if constexpr (DO2) {
                t.x = Q1c[0].v[0]; t.y = Q1c[0].v[1];           *reinterpret_cast<d2l*>(mine + 2 * TXW) = t;
                t.x = Q1c[RY - 1].v[0]; t.y = Q1c[RY - 1].v[1]; *reinterpret_cast<d2l*>(mine + 3 * TXW) = t;
            }
            if constexpr (DO3) {
                t.x = Q2c[0].v[0]; t.y = Q2c[0].v[1];           *reinterpret_cast<d2l*>(mine + 4 * TXW) = t;
                t.x = Q2c[RY - 1].v[0]; t.y = Q2c[RY - 1].v[1]; *reinterpret_cast<d2l*>(mine + 5 * TXW) = t;
            }
            if (hwave) {   // bottom wave: slot 0 "L0 last row"; top wave: slot NW+1 "L0 first row"
                t.x = YH.v[0]; t.y = YH.v[1];
                *reinterpret_cast<d2l*>(buf + ((w == 0) ? TXW : (NW + 1) * SLOT) + lane * VX) = t;
            }
        }
        diff3_lds_barrier();
        // the neighbours' rows are read where they are used (level by level: six rows held from here on were 24 registers too many)
        const double* od = buf + (size_t)w * SLOT + lane * VX;         // slot below: rows 1, 3, 5 (last rows)
        const double* ou = buf + (size_t)(w + 2) * SLOT + lane * VX;   // slot above: rows 0, 2, 4 (first rows)
        auto lds_row = [&](const double* q) {
            const d2l t = *reinterpret_cast<const d2l*>(q);
            DVec<VX> o;
            o.v[0] = t.x; o.v[1] = t.y;
            return o;
        };

        // ---- first step: L1 on plane m ----
        const bool zb = (m <= 0) || (m >= nz - 1);   // block-uniform: a z-boundary plane of L1 comes from Bnd
        if (zb) {
            const long offB = ((long)sz * kcl(m) + (long)sy * j0) * 8;
            const long remB = array_bytes - offB;
            const __amdgpu_buffer_rsrc_t rB = diff3_rsrc((uintptr_t)a.Bnd + (uintptr_t)offB, (unsigned)(remB > 0x7ffffff0L ? 0x7ffffff0L : remB));
            load_rows(Q1n, rB, 0);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): rare (first / last chunk), no load of this branch pending at the join
        } else {
            const bool own_plane = (m >= k0) && (m < k1);
            const DVec<VX> yd0 = lds_row(od + TXW), yu0 = lds_row(ou);
            double sl1 = 0.0;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const double xl0 = diff3_lane_up1_edge(cR[r].v[VX - 1], ED[r]);
                const double xrL = diff3_lane_down1_edge(cR[r].v[0], ED[r]);
                double r1[VX];
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    const double xm = (v == 0) ? xl0 : cR[r].v[v == 0 ? 0 : v - 1];
                    const double xp = (v == VX - 1) ? xrL : cR[r].v[v == VX - 1 ? v : v + 1];
                    const double ym = (r == 0) ? yd0.v[v] : cR[r == 0 ? 0 : r - 1].v[v];
                    const double yp = (r == RY - 1) ? yu0.v[v] : cR[r == RY - 1 ? r : r + 1].v[v];
                    r1[v] = diff3_point<false>(cR[r].v[v], xm, xp, ym, yp, zmR[r].v[v], zpR[r].v[v],
                                               HT[(S + 2) % NR][r].v[v], cf, Q1n[r].v[v]);
                }
                if (xb_tile) {   // x-boundary own cells of L1 come from Bnd (through the edge register); they have no residual
                    asm volatile("" ::: "memory");
                    Q1n[r].v[0] = bndL ? xl0 : Q1n[r].v[0];
                    Q1n[r].v[VX - 1] = bndR ? xrL : Q1n[r].v[VX - 1];
                    if constexpr (NORM) { r1[0] = bndL ? 0.0 : r1[0]; r1[VX - 1] = bndR ? 0.0 : r1[VX - 1]; }
                }
                if constexpr (NORM) {
                    if (own_plane && rm[r]) accum(sl1, r1);
                }
#if DIFF3_M3_ROW_FENCE
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (bb) { asm volatile("" ::: "memory"); Q1n[0] = YH; }
            if (bt) { asm volatile("" ::: "memory"); Q1n[RY - 1] = YH; }
            if constexpr (NORM) flush(0, sl1, acc1);
        }

        // the halo registers of plane m are dead: refill with the halos of plane m+1 (nothing once the chunk ends)
        {
            const bool more = m + 1 <= m1 && !DIFF3_DBG(a, 2);
            load_halo(YH, ED, more ? so : (int)OOR, more ? so - 2 * ps : (int)OOR);
        }
#if DIFF3_M3_SCHED == 2
        // Ht plane m+1 (needed by the first step of iteration m+1) into the slot plane m-3 left at the end of the last iteration
        load_rows(HT[(S + 3) % NR], rHt, (m + 1 <= m1 && !DIFF3_DBG(a, 2)) ? so - ps : (int)OOR);
#endif

        // ---- second step: L2 on plane m-1 (its boundary cells are X's own: L0 plane m-1 is still in zmR) ----
        if constexpr (DO2) {
            const bool zb2 = (m - 1 <= 0) || (m - 1 >= nz - 1);
            if (zb2) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < RY; ++r) Q2n[r] = zmR[r];
            } else {
                const bool own2 = (m - 1 >= k0) && (m - 1 < k1);
                const DVec<VX> yd1 = lds_row(od + 3 * TXW), yu1 = lds_row(ou + 2 * TXW);
                double sl2 = 0.0;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const double fromL = diff3_lane_up1_z(Q1c[r].v[VX - 1]);
                    const double fromR = diff3_lane_down1_z(Q1c[r].v[0]);
                    double r2[VX];
#pragma unroll
                    for (int v = 0; v < VX; ++v) {
                        const double xm = (v == 0) ? fromL : Q1c[r].v[v == 0 ? 0 : v - 1];
                        const double xp = (v == VX - 1) ? fromR : Q1c[r].v[v == VX - 1 ? v : v + 1];
                        const double ym = (r == 0) ? yd1.v[v] : Q1c[r == 0 ? 0 : r - 1].v[v];
                        const double yp = (r == RY - 1) ? yu1.v[v] : Q1c[r == RY - 1 ? r : r + 1].v[v];
                        r2[v] = diff3_point<false>(Q1c[r].v[v], xm, xp, ym, yp, Q1m[r].v[v], Q1n[r].v[v], HT[(S + 1) % NR][r].v[v], cf, Q2n[r].v[v]);
                    }
                    if (xb_tile) {
                        asm volatile("" ::: "memory");
                        Q2n[r].v[0] = bndL ? zmR[r].v[0] : Q2n[r].v[0];
                        Q2n[r].v[VX - 1] = bndR ? zmR[r].v[VX - 1] : Q2n[r].v[VX - 1];
                        if constexpr (NORM) { r2[0] = bndL ? 0.0 : r2[0]; r2[VX - 1] = bndR ? 0.0 : r2[VX - 1]; }
                    }
                    if constexpr (NORM) {
                        if (own2 && rm[r]) accum(sl2, r2);
                    }
#if DIFF3_M3_ROW_FENCE
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                if (bb) { asm volatile("" ::: "memory"); Q2n[0] = zmR[0]; }
                if (bt) { asm volatile("" ::: "memory"); Q2n[RY - 1] = zmR[RY - 1]; }
                if constexpr (NORM) flush(1, sl2, acc2);
            }
        }

#if DIFF3_M3_SCHED == 0 || DIFF3_M3_SCHED == 2
        // L0 plane m-1 is dead: refill with plane m+3 (L0 is needed up to plane k1 + 2 = m1 + 1)
        load_rows(P[S % NR], rA, (m + 3 <= m1 + 1 && m + 3 <= nz - 1 && !DIFF3_DBG(a, 2)) ? so : (int)OOR);
#else
        // the slots that died one iteration ago (L0 plane m-2, Ht plane m-3) receive L0 plane m+2 and Ht plane m+1
        __builtin_amdgcn_sched_barrier(0);      // (not hoisted above the second step: the registers they land in are the budget's last)
        load_rows(P[(S + 3) % NR], rA, (m + 2 <= m1 + 1 && m + 2 <= nz - 1 && !DIFF3_DBG(a, 2)) ? so - ps : (int)OOR);
        load_rows(HT[(S + 3) % NR], rHt, (m + 1 <= m1 && !DIFF3_DBG(a, 2)) ? so - ps : (int)OOR);
#endif

        // ---- third step: L3 on plane m-2 ----
        if constexpr (DO3) {
            __builtin_amdgcn_sched_barrier(0);
            const DVec<VX> yd2 = lds_row(od + 5 * TXW), yu2 = lds_row(ou + 4 * TXW);
#if DIFF3_M3_PARK
            DVec<VX> Q2m[RY];
#pragma unroll
            for (int r = 0; r < RY; ++r) Q2m[r] = lds_row(park + r * TXW);
#endif
            double sl3 = 0.0;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const double fromL = diff3_lane_up1_z(Q2c[r].v[VX - 1]);
                const double fromR = diff3_lane_down1_z(Q2c[r].v[0]);
                double res[VX], h3[VX];
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    const double xm = (v == 0) ? fromL : Q2c[r].v[v == 0 ? 0 : v - 1];
                    const double xp = (v == VX - 1) ? fromR : Q2c[r].v[v == VX - 1 ? v : v + 1];
                    const double ym = (r == 0) ? yd2.v[v] : Q2c[r == 0 ? 0 : r - 1].v[v];
                    const double yp = (r == RY - 1) ? yu2.v[v] : Q2c[r == RY - 1 ? r : r + 1].v[v];
                    res[v] = diff3_point<false>(Q2c[r].v[v], xm, xp, ym, yp, Q2m[r].v[v], Q2n[r].v[v], HT[S % NR][r].v[v], cf, h3[v]);
                }
                const int sor = (rm[r] && !DIFF3_DBG(a, 1)) ? so + r * rs : (int)OOR;
                double r3 = res[0], g3 = h3[0];
                if (has_split) { asm volatile("" ::: "memory"); r3 = cm[0] ? res[0] : res[1]; g3 = cm[0] ? h3[0] : h3[1]; }
                if constexpr (WRES) diff3_bst2_nt(rD, sv4, sor, res[0], res[1]);
                diff3_bst2_nt(rC, sv4, sor, h3[0], h3[1]);
                if constexpr (WRES) diff3_bst1(rD, sv2, sor, r3);
                diff3_bst1(rC, sv2, sor, g3);
#if DIFF3_STORE_NOP >= 0
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop %0" ::"n"(DIFF3_STORE_NOP));
                __builtin_amdgcn_sched_barrier(0);
#endif
                if constexpr (NORM) {
                    if (xb_tile) { asm volatile("" ::: "memory"); res[0] = bndL ? 0.0 : res[0]; res[VX - 1] = bndR ? 0.0 : res[VX - 1]; }
                    if (rm[r]) accum(sl3, res);
                }
            }
            if constexpr (NORM) flush(2, sl3, acc3);
        }
#if DIFF3_M3_PARK
        if constexpr (DO2) {      // L2 plane m-2 is the next iteration's z-minus plane: to LDS (this wave's own rows; in order behind the reads above)
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                d2l t;
                t.x = Q2c[r].v[0]; t.y = Q2c[r].v[1];
                *reinterpret_cast<d2l*>(park + r * TXW) = t;
            }
        }
#endif
#if DIFF3_M3_SCHED == 0
        // Ht plane m-2 is dead: refill with plane m+2 (first needed by L1 on plane m+2 <= m1)
        load_rows(HT[S % NR], rHt, (m + 2 <= m1 && !DIFF3_DBG(a, 2)) ? so : (int)OOR);
#endif
        so += ps;
    };

    using T = std::true_type;
    using F = std::false_type;
    // four warm-up iterations (L1 planes k0-2 .. k0+1, L2 planes k0-1, k0), then the steady state
    step(std::integral_constant<int, 0>{}, F{}, F{}, m0);
    step(std::integral_constant<int, 1>{}, F{}, F{}, m0 + 1);
    step(std::integral_constant<int, 2>{}, T{}, F{}, m0 + 2);
    step(std::integral_constant<int, 3>{}, T{}, F{}, m0 + 3);
    int m = m0 + 4;
    for (; m + NR - 1 <= m1; m += NR) {
        step(std::integral_constant<int, 0>{}, T{}, T{}, m);
        step(std::integral_constant<int, 1>{}, T{}, T{}, m + 1);
        step(std::integral_constant<int, 2>{}, T{}, T{}, m + 2);
        step(std::integral_constant<int, 3>{}, T{}, T{}, m + 3);
    }
    if (m <= m1) { step(std::integral_constant<int, 0>{}, T{}, T{}, m); ++m; }
    if (m <= m1) { step(std::integral_constant<int, 1>{}, T{}, T{}, m); ++m; }
    if (m <= m1) { step(std::integral_constant<int, 2>{}, T{}, T{}, m); ++m; }
    }

    if constexpr (NORM) {
        // every lane takes part (lanes switched off for the march hold zeros)
        const bool any = cm[0] || cm[1];
#if DIFF3_M3_LDS_ACC
        const double acc1 = accl[0][tid], acc2 = accl[1][tid], acc3 = accl[2][tid];
        accl[0][tid] = 0.0; accl[1][tid] = 0.0; accl[2][tid] = 0.0;
#endif
        const double w1 = diff3_wave_sum(any ? acc1 : 0.0), w2 = diff3_wave_sum(any ? acc2 : 0.0), w3 = diff3_wave_sum(any ? acc3 : 0.0);
        if (lane == 0) { red[w] += w1; red[NW + w] += w2; red[2 * NW + w] += w3; }
    }

    if constexpr (NORM) {
        const double sc2 = a.scale * a.scale;
        __syncthreads();
        if (tid == 0) {
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int i = 0; i < NW; ++i) { s1 += red[i]; s2 += red[NW + i]; s3 += red[2 * NW + i]; }
            a.partials1[unit] = s1 * sc2; a.partials2[unit] = s2 * sc2; a.partials3[unit] = s3 * sc2;
        }
    }
}

constexpr size_t DIFF3_MARCH3_LDS = ((size_t)2 * (8 + 2) * 6 * 128 + (size_t)DIFF3_M3_PARK * 8 * 3 * 128) * sizeof(double);   // 122 880 (+ 24 576) bytes

// true if the fused three-step kernel can serve this problem
static inline bool diff3_can_fuse3(const double* Ht, const double* X, const double* Bnd, const double* Y, const double* dH,
                                   int nx, int ny, int nz)
{
    const uintptr_t al = (uintptr_t)Ht | (uintptr_t)X | (uintptr_t)Bnd | (uintptr_t)Y | (uintptr_t)dH;   // dH may be null
    return (nx % 2 == 0) && nx >= 128 && ny >= 24 && nz >= 5 && (al & 15) == 0 && (long)nx * ny * 8 * 16 < (1L << 31);
}

// Launch on `stream`; *nparts = number of per-unit partials written to each of partials1/2/3 (norm only).
// zc_opt: planes per z-chunk (0 = auto); ncu: compute units of the device.
static inline hipError_t diff3_launch3(Diff3Args3 a, bool norm, int zc_opt, int xcd_opt, hipStream_t stream, int max_partials,
                                       int* nparts, int ncu = 256, long* bal_info = nullptr)
{
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    *nparts = 0;
    if (bal_info) *bal_info = 0;
    if (wx <= 0 || wy <= 0 || wz <= 0) return hipSuccess;
    if (!diff3_can_fuse3(a.Ht, a.X, a.Bnd, a.Y, a.dH, a.nx, a.ny, a.nz)) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipSuccess;
#define FPR_M3_ATTR(...) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diff3_march3<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIFF3_MARCH3_LDS)
        FPR_M3_ATTR(true, true);
#ifndef DIFF3_M3_FEW
        FPR_M3_ATTR(false, true); FPR_M3_ATTR(true, false); FPR_M3_ATTR(false, false);
#endif
#undef FPR_M3_ATTR
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int span = a.hi[0] - (a.lo[0] & ~1);
    a.ntx = (span + 121) / 122;
    a.sx = (span + a.ntx - 1) / a.ntx;
    a.sx += a.sx & 1;
    a.xalign = (a.sx >= 32 && a.sx <= 106 && a.nx % 16 == 0 && ((((uintptr_t)a.X | (uintptr_t)a.Ht)) & 127) == 0) ? 1 : 0;
    const long psb = (long)a.nx * a.ny * 8;
    const int zc_max = (int)((1L << 31) / psb) - 12;
    if (zc_max < 1) return hipErrorInvalidValue;
    constexpr int SYB = 20;
    a.nby = (wy + SYB - 1) / SYB;
    const long tiles = (long)a.ntx * a.nby;
    const long slots = ncu > 0 ? ncu : 256;
    // chunking: a chunk of zc planes costs zc + 4 plane-iterations (+ prologue); the grid is `slots` workgroups when the units of
    // ntz chunks per tile are a few more than that (left-over units sliced over all workgroups), else the plain grid
    int zc = zc_opt;
    if (zc <= 0) {
        long best = -1;
        for (int ntz = 1; ntz <= wz; ++ntz) {
            const int z = (wz + ntz - 1) / ntz;
            if (z > zc_max) continue;
            if (z < 8 && ntz > 1) break;
            const long nb = tiles * ((wz + z - 1) / z);
            const long rounds = (nb + slots - 1) / slots;
            long cost = rounds * (z + 8);
            const long r = nb - slots;
            if (r > 0 && r <= slots / 4) {                        // one round + thin slices of the left-over units
                const long sp = slots / r, q = (z + sp - 1) / sp;
                cost = (z + 8) + (q + 8);
            }
            if (best < 0 || cost <= best) { best = cost; zc = z; }
        }
        if (zc <= 0) zc = wz < zc_max ? wz : zc_max;
    }
    if (zc > wz) zc = wz;
    if (zc > zc_max) zc = zc_max;
    a.zc = zc;
    a.ntz = (wz + zc - 1) / zc;
    const long nblk = tiles * a.ntz;
    if (nblk > 0x7fffffffL || (norm && nblk > max_partials)) return hipErrorInvalidValue;
#ifdef FPR_TUNE
    a.dbg = xcd_opt >> 4;
#endif
    xcd_opt &= 15;
    if (xcd_opt == 0) xcd_opt = (nblk >= 64 && nblk <= 2 * slots) ? 1 : 3;
    a.xcd_remap = (xcd_opt == 1 && nblk >= 64) ? 1 : 0;
    const bool wres = a.dH != nullptr;
    a.nfull = (int)nblk; a.bal_sp = 1; a.bal_q = zc;
    long grid = nblk;
    {
        const long r = nblk - slots;
        if (r > 0 && r <= slots / 4) {
            a.nfull = (int)slots;
            a.bal_sp = (int)(slots / r);
            a.bal_q = (zc + a.bal_sp - 1) / a.bal_sp;
            grid = slots + r * a.bal_sp;
            if (bal_info) *bal_info = r * 1000000 + (long)a.bal_sp * 1000 + a.bal_q;
        }
    }
    if (norm && grid > max_partials) return hipErrorInvalidValue;
    const size_t lds = DIFF3_MARCH3_LDS;
#define FPR_M3_GO(N_, W_) k_diff3_march3<N_, W_><<<(int)grid, 512, lds, stream>>>(a)
#ifdef DIFF3_M3_FEW
    if (!wres || !norm) return hipErrorInvalidValue;
    FPR_M3_GO(true, true);
#else
    if (wres) { if (norm) FPR_M3_GO(true, true); else FPR_M3_GO(false, true); }
    else { if (norm) FPR_M3_GO(true, false); else FPR_M3_GO(false, false); }
#endif
#undef FPR_M3_GO
    *nparts = (int)grid;
    return hipGetLastError();
}
