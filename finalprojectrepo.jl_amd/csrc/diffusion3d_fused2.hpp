// diffusion3d_fused2.hpp -- TWO pseudo-transient iterations of the fused 7-point update in one pass over
// memory (temporal blocking of scripts-part1/part1_kernel_programming.jl:179-192: two trips through the
// `while` body, i.e. step(A -> B), swap, step(B -> A'), without materialising the intermediate field B).
//
// Per launch and interior cell the kernel reads Htau and Ht once and writes the twice-updated field and
// the residual of the second step: 32 B for two iterations instead of 64 B.  Results are bit-identical
// to two launches of k_diff3_march (same diff3_point expression, evaluated on the same operands).
//
// Levels: L0 = Htau (buffer A), L1 = field after the first step, L2 = field after the second step
// (written to buffer C; dHdtau receives the residual of the second step).  The reference never writes the
// boundary cells of its two ping-pong buffers, so each keeps its own boundary values forever: L1's boundary
// cells are the boundary cells of the reference's *other* buffer (`B` here, only its boundary is read), and
// C must carry A's boundary (the caller copies it once).
//
// Geometry (wave64, 2 cells per lane, 4 rows per lane, 4 waves stacked in y = block tile 128 x 16):
//   * the block marches in z; iteration m computes L1 on plane m for its whole tile (from the L0 ring of 4
//     register planes m-1..m+2, exactly as k_diff3_march<PIPE>), then L2 on plane m-1 from the three L1
//     planes m-2, m-1, m held in registers;
//   * x-neighbours by DPP wave shifts for both levels; y-neighbours between the 4 waves through ONE LDS
//     exchange + raw s_barrier per iteration that carries the first/last rows of L0(m) and of L1(m-1);
//   * L1 is valid on the whole 128 x 16 tile (L0 halo cells / rows come from global memory as before), L2 on
//     the tile shrunk by one cell: tiles overlap by 2 in x and y and chunks by 2 planes in z (redundant L1
//     work: 16/14 in y, (zc+2)/zc in z; x uses 5 tiles of ~102 owned cells for a 512-cell line);
//   * domain-boundary cells of L1 are taken from B: the registers that would hold the (non-existent) L0 halo
//     beyond the boundary carry the B value instead, so the steady-state loop has no extra loads.
// Requirements (the caller falls back to two single-step launches otherwise): nx even, all arrays 16-byte
// aligned, ny >= 16, nz >= 3.
#pragma once
#include "diffusion3d_kernels.hpp"

struct Diff3Args2 {
    const double* __restrict__ Ht;
    const double* __restrict__ A;   // L0 (Htau)
    const double* __restrict__ B;   // boundary values of L1 (the reference's Htau2); interior never read
    double* __restrict__ C;         // L2 (interior of the box)
    double* __restrict__ dH;        // residual of the second step
    int nx, ny, nz;
    int lo[3], hi[3];               // output box, clipped to the interior [1, n-1)
    double dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
    double scale;
    double* partials1;              // per-block sum((r1*scale)^2) over owned cells (first step)
    double* partials2;              // same for the second step
    int zc, ntx, nby, ntz, sx;      // planes per chunk, tile counts, owned cells per tile in x
    int xcd_remap;
};

template <bool NORM>
__global__ __launch_bounds__(256) void k_diff3_march2(Diff3Args2 a)
{
    constexpr int VX = 2, RY = 4, TXW = 128, SYB = 4 * RY - 2;
    __shared__ double red[8];
    // [parity][wave][L0 first row, L0 last row, L1 first row, L1 last row][TXW]
    __shared__ __attribute__((aligned(16))) double xrow[2 * 4 * 4 * TXW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;

    int bid = blockIdx.x;
    if (a.xcd_remap == 1) {
        const int nblk = gridDim.x;
        const int q = nblk >> 3, rem = nblk & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        bid = xcd * q + (xcd < rem ? xcd : rem) + slot;
    }
    const int tx = bid % a.ntx;
    const int by = (bid / a.ntx) % a.nby;
    const int tz = bid / (a.ntx * a.nby);

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;

    // ---- x: owned output cells [ol, oh); the tile's own cells start at the even index s <= ol-1 ----
    const int ol = a.lo[0] + tx * a.sx;
    const int oh = (ol + a.sx < a.hi[0]) ? ol + a.sx : a.hi[0];
    const int s = (ol - 1) & ~1;
    const int ib = s + lane * VX;
    int ilast = (oh + 1) & ~1;                       // last pair that is needed (L0 at oh+1)
    ilast = ilast < nx - 2 ? ilast : nx - 2;
    const int ibc = ib < ilast ? ib : ilast;         // lanes beyond re-read the last needed pair
    const bool bndL = (ib == 0);                     // own cell v=0 is the x-boundary
    const bool bndR = (ib + 1 == nx - 1);            // own cell v=1 is the x-boundary
    const bool is_edge = (lane == 0) || (lane == 63) || bndR;
    const double* __restrict__ Esrc = (bndL || bndR) ? a.B : a.A;
    int ie = bndL ? 0 : (bndR ? nx - 1 : (lane == 0 ? ib - 1 : ib + VX));
    ie = ie < 0 ? 0 : (ie > nx - 1 ? nx - 1 : ie);

    // ---- y: owned rows [oly, ohy); block rows y1 .. y1+15 ----
    const int oly = a.lo[1] + by * SYB;
    const int ohy = (oly + SYB < a.hi[1]) ? oly + SYB : a.hi[1];
    const int y1 = (oly - 1 < ny - 4 * RY) ? oly - 1 : ny - 4 * RY;
    const int j0 = y1 + w * RY;
    const bool bb = (w == 0) && (y1 == 0);                    // own row 0 of wave 0 is the y-boundary
    const bool bt = (w == 3) && (y1 + 4 * RY - 1 == ny - 1);  // own last row of wave 3 is the y-boundary
    const bool need_gd = (w == 0), need_gu = (w == 3);
    const int jd = bb ? 0 : (j0 > 0 ? j0 - 1 : 0);
    const int ju = bt ? ny - 1 : (j0 + RY < ny - 1 ? j0 + RY : ny - 1);
    const double* __restrict__ YDsrc = bb ? a.B : a.A;
    const double* __restrict__ YUsrc = bt ? a.B : a.A;

    // ---- z: owned planes [k0, k1); iterations m0 .. m1 ----
    const int k0 = a.lo[2] + tz * a.zc;
    const int k1 = (k0 + a.zc < a.hi[2]) ? k0 + a.zc : a.hi[2];
    const int m0 = k0 - 1, m1 = k1;

    bool cm[VX], rm[RY];
#pragma unroll
    for (int v = 0; v < VX; ++v) cm[v] = (ib + v >= ol) && (ib + v < oh);
#pragma unroll
    for (int r = 0; r < RY; ++r) rm[r] = (j0 + r >= oly) && (j0 + r < ohy);

    const Diff3Coef cf{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};
    const double* __restrict__ H = a.A;
    auto kcl = [&](int k) { return k < 0 ? 0 : (k > nz - 1 ? nz - 1 : k); };

    DVec<VX> P[4][RY];        // L0 planes (m-1, m, m+1, m+2) in slots ((m-m0)+{0,1,2,3}) & 3
    DVec<VX> HT[2][RY];       // Ht of planes m, m+1 in slots (m-m0) & 1
    DVec<VX> HTo[2][RY];      // Ht of plane m-1 (copied from HT before its slot is refilled)
    DVec<VX> YD[2], YU[2];    // global L0 halo rows of planes m, m+1 (or the B boundary row, see bb / bt)
    double ED[2][RY];         // L0 tile-edge cells of planes m, m+1 (or the B boundary cell, see bndL / bndR)
    DVec<VX> Qm[RY], Qc[RY], Qn[RY];  // L1 planes m-2, m-1, m
    double acc1 = 0.0, acc2 = 0.0;
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int v = 0; v < VX; ++v) { Qm[r].v[v] = 0.0; Qc[r].v[v] = 0.0; HTo[0][r].v[v] = 0.0; HTo[1][r].v[v] = 0.0; }

    auto load_plane = [&](DVec<VX>(&dst)[RY], int k) {
        const size_t base = (size_t)ibc + sz * kcl(k);
#pragma unroll
        for (int r = 0; r < RY; ++r) dst[r] = diff3_ldv<VX>(H + base + sy * (j0 + r));
    };
    auto load_aux = [&](DVec<VX>(&ht)[RY], DVec<VX>& yd, DVec<VX>& yu, double (&e)[RY], int k) {
        const size_t kc = sz * kcl(k);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            ht[r] = diff3_ldv<VX>(a.Ht + (size_t)ibc + sy * (j0 + r) + kc);
            e[r] = is_edge ? Esrc[(size_t)ie + sy * (j0 + r) + kc] : 0.0;
        }
        if (need_gd) yd = diff3_ldv<VX>(YDsrc + (size_t)ibc + sy * jd + kc);
        if (need_gu) yu = diff3_ldv<VX>(YUsrc + (size_t)ibc + sy * ju + kc);
    };

    load_plane(P[0], m0 - 1);
    load_plane(P[1], m0);
    load_plane(P[2], m0 + 1);
    load_plane(P[3], m0 + 2);
    load_aux(HT[0], YD[0], YU[0], ED[0], m0);
    load_aux(HT[1], YD[1], YU[1], ED[1], m0 + 1);

    auto step = [&](auto Sc, int m) {
        constexpr int S = decltype(Sc)::value;
        DVec<VX>(&zmR)[RY] = P[S & 3];
        DVec<VX>(&cR)[RY] = P[(S + 1) & 3];
        DVec<VX>(&zpR)[RY] = P[(S + 2) & 3];
        constexpr int hsl = S & 1;

        // ---- one LDS exchange for both levels: L0 rows of plane m, L1 rows of plane m-1 ----
        double* buf = xrow + (size_t)(m & 1) * (4 * 4 * TXW);
        double* mine = buf + (size_t)w * (4 * TXW) + lane * VX;
#pragma unroll
        for (int v = 0; v < VX; ++v) {
            mine[v] = cR[0].v[v];
            mine[TXW + v] = cR[RY - 1].v[v];
            mine[2 * TXW + v] = Qc[0].v[v];
            mine[3 * TXW + v] = Qc[RY - 1].v[v];
        }
        diff3_lds_barrier();
        const double* od = buf + (size_t)(w > 0 ? w - 1 : 0) * (4 * TXW) + lane * VX;  // wave below: rows 1 (L0 last), 3 (L1 last)
        const double* ou = buf + (size_t)(w < 3 ? w + 1 : 3) * (4 * TXW) + lane * VX;  // wave above: rows 0 (L0 first), 2 (L1 first)
        DVec<VX> yd0 = YD[hsl], yu0 = YU[hsl], yd1, yu1;
#pragma unroll
        for (int v = 0; v < VX; ++v) {
            const double ld0 = od[TXW + v], lu0 = ou[v];
            yd0.v[v] = (w > 0) ? ld0 : yd0.v[v];
            yu0.v[v] = (w < 3) ? lu0 : yu0.v[v];
            yd1.v[v] = od[3 * TXW + v];
            yu1.v[v] = ou[2 * TXW + v];
        }

        // ---- first step: L1 on plane m ----
        const bool zb = (m <= 0) || (m >= nz - 1);   // block-uniform: a z-boundary plane of L1 comes from B
        if (zb) {
            const size_t base = (size_t)ibc + sz * kcl(m);
#pragma unroll
            for (int r = 0; r < RY; ++r) Qn[r] = diff3_ldv<VX>(a.B + base + sy * (j0 + r));
        } else {
            const bool own_plane = (m >= k0) && (m < k1);
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const double fromL = diff3_lane_up1(cR[r].v[VX - 1]);
                const double fromR = diff3_lane_down1(cR[r].v[0]);
                const double xl0 = (lane == 0) ? ED[hsl][r] : fromL;
                const double xrL = (lane == 63) ? ED[hsl][r] : fromR;
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    const double xm = (v == 0) ? xl0 : cR[r].v[v == 0 ? 0 : v - 1];
                    const double xp = (v == VX - 1) ? xrL : cR[r].v[v == VX - 1 ? v : v + 1];
                    const double ym = (r == 0) ? yd0.v[v] : cR[r == 0 ? 0 : r - 1].v[v];
                    const double yp = (r == RY - 1) ? yu0.v[v] : cR[r == RY - 1 ? r : r + 1].v[v];
                    double h1;
                    const double r1 = diff3_point(cR[r].v[v], xm, xp, ym, yp, zmR[r].v[v], zpR[r].v[v],
                                                  HT[hsl][r].v[v], cf, h1);
                    Qn[r].v[v] = h1;
                    if constexpr (NORM) {
                        if (own_plane && rm[r] && cm[v]) { const double t = r1 * a.scale; acc1 += t * t; }
                    }
                }
                // x-boundary own cells of L1 come from B (carried in the edge register)
                Qn[r].v[0] = bndL ? ED[hsl][r] : Qn[r].v[0];
                Qn[r].v[VX - 1] = bndR ? ED[hsl][r] : Qn[r].v[VX - 1];
            }
            // y-boundary own rows of L1 come from B (carried in the halo-row registers)
#pragma unroll
            for (int v = 0; v < VX; ++v) {
                Qn[0].v[v] = bb ? YD[hsl].v[v] : Qn[0].v[v];
                Qn[RY - 1].v[v] = bt ? YU[hsl].v[v] : Qn[RY - 1].v[v];
            }
        }

        // plane m-1 of L0 and the aux registers of plane m are dead: keep Ht(m) for the second step of the next
        // iteration, then refill (L0 plane m+3, aux of plane m+2) so the loads fly during the second step
#pragma unroll
        for (int r = 0; r < RY; ++r) HTo[hsl][r] = HT[hsl][r];
        if (m + 2 <= m1) {
            load_plane(P[S & 3], m + 3);
            load_aux(HT[hsl], YD[hsl], YU[hsl], ED[hsl], m + 2);
        }

        // ---- second step: L2 on plane m-1 ----
        if (m - 1 >= k0) {
            const int k = m - 1;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const double fromL = diff3_lane_up1(Qc[r].v[VX - 1]);
                const double fromR = diff3_lane_down1(Qc[r].v[0]);
                double res[VX], h2[VX];
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    const double xm = (v == 0) ? fromL : Qc[r].v[v == 0 ? 0 : v - 1];
                    const double xp = (v == VX - 1) ? fromR : Qc[r].v[v == VX - 1 ? v : v + 1];
                    const double ym = (r == 0) ? yd1.v[v] : Qc[r == 0 ? 0 : r - 1].v[v];
                    const double yp = (r == RY - 1) ? yu1.v[v] : Qc[r == RY - 1 ? r : r + 1].v[v];
                    res[v] = diff3_point(Qc[r].v[v], xm, xp, ym, yp, Qm[r].v[v], Qn[r].v[v], HTo[hsl ^ 1][r].v[v], cf,
                                         h2[v]);
                }
                if (rm[r]) {
                    const size_t id = (size_t)ib + sy * (size_t)(j0 + r) + sz * (size_t)k;
                    if (cm[0] && cm[1]) {
                        typedef double d2v __attribute__((ext_vector_type(2)));
                        d2v rv, hv;
                        rv.x = res[0]; rv.y = res[1];
                        hv.x = h2[0]; hv.y = h2[1];
                        __builtin_nontemporal_store(rv, reinterpret_cast<d2v*>(a.dH + id));
                        __builtin_nontemporal_store(hv, reinterpret_cast<d2v*>(a.C + id));
                    } else {
                        if (cm[0]) { a.dH[id] = res[0]; a.C[id] = h2[0]; }
                        if (cm[1]) { a.dH[id + 1] = res[1]; a.C[id + 1] = h2[1]; }
                    }
                    if constexpr (NORM) {
#pragma unroll
                        for (int v = 0; v < VX; ++v)
                            if (cm[v]) { const double t = res[v] * a.scale; acc2 += t * t; }
                    }
                }
            }
        }

        // ---- rotate the L1 window (plain register copies: none of these is a load destination) ----
#pragma unroll
        for (int r = 0; r < RY; ++r) { Qm[r] = Qc[r]; Qc[r] = Qn[r]; }
    };

    int m = m0;
    for (; m + 3 <= m1; m += 4) {
        step(std::integral_constant<int, 0>{}, m);
        step(std::integral_constant<int, 1>{}, m + 1);
        step(std::integral_constant<int, 2>{}, m + 2);
        step(std::integral_constant<int, 3>{}, m + 3);
    }
    if (m <= m1) { step(std::integral_constant<int, 0>{}, m); ++m; }
    if (m <= m1) { step(std::integral_constant<int, 1>{}, m); ++m; }
    if (m <= m1) { step(std::integral_constant<int, 2>{}, m); ++m; }

    if constexpr (NORM) {
        const double s1 = diff3_block_sum256(acc1, red, tid);
        const double s2 = diff3_block_sum256(acc2, red + 4, tid);
        if (tid == 0) { a.partials1[blockIdx.x] = s1; a.partials2[blockIdx.x] = s2; }
    }
}

// true if the fused two-step kernel can serve this problem
static inline bool diff3_can_fuse2(const double* Ht, const double* A, const double* B, const double* C, const double* dH,
                                   int nx, int ny, int nz)
{
    const uintptr_t al = (uintptr_t)Ht | (uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)dH;
    return (nx % 2 == 0) && nx >= 4 && ny >= 16 && nz >= 3 && (al & 15) == 0;
}

#ifndef DIFF3_TARGET_BLOCKS2
#define DIFF3_TARGET_BLOCKS2 3072
#endif

// Launch on `stream`; *nparts = number of per-block partials written to each of partials1/partials2 (norm only).
static inline hipError_t diff3_launch2(Diff3Args2 a, bool norm, int zc_opt, int xcd_opt, hipStream_t stream,
                                       int max_partials, int* nparts)
{
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    *nparts = 0;
    if (wx <= 0 || wy <= 0 || wz <= 0) return hipSuccess;
    if (!diff3_can_fuse2(a.Ht, a.A, a.B, a.C, a.dH, a.nx, a.ny, a.nz)) return hipErrorInvalidValue;
    // owned cells per x-tile: the tile's own cells [s, s+128) with s = (ol-1)&~1 must contain ol-1 .. oh
    const int maxsx = ((a.lo[0] - 1) & 1) ? 124 : 126;
    a.ntx = (wx + maxsx - 1) / maxsx;
    a.sx = (wx + a.ntx - 1) / a.ntx;
    a.sx += a.sx & 1;   // even, so every tile start has the parity of the first
    a.nby = (wy + 13) / 14;
    const long tiles_xy = (long)a.ntx * a.nby;
    int zc = zc_opt;
    if (zc <= 0) {
        long want = (DIFF3_TARGET_BLOCKS2 + tiles_xy - 1) / tiles_xy;
        if (want < 1) want = 1;
        zc = (int)((wz + want - 1) / want);
        if (zc < 16) zc = wz < 16 ? wz : 16;
    }
    if (zc > wz) zc = wz;
    a.zc = zc;
    a.ntz = (wz + zc - 1) / zc;
    const long nblk = tiles_xy * a.ntz;
    if (nblk > 0x7fffffffL || (norm && nblk > max_partials)) return hipErrorInvalidValue;
    a.xcd_remap = (xcd_opt == 1 && nblk >= 64) ? 1 : 0;
    if (norm) k_diff3_march2<true><<<(int)nblk, 256, 0, stream>>>(a);
    else k_diff3_march2<false><<<(int)nblk, 256, 0, stream>>>(a);
    *nparts = (int)nblk;
    return hipGetLastError();
}
