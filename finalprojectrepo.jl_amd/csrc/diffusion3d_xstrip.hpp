// diffusion3d_xstrip.hpp -- the one-cell shell next to an x-neighbour of a decomposed run, worked on in COMPACT STRIPS
// (role of @hide_communication's boundary width in x, part1_kernel_programming.jl:185-188, and of update_halo!'s x send /
// receive buffers, :182,187, for a fused pair of iterations).
//
// In a column-major array the cells of one x-column lie nx*8 bytes apart: a kernel whose lanes run along y touches one 64-byte
// sector per cell, and beside a launch that saturates the memory system every one of those sector requests waits its turn --
// k_diff3_slab2 took 280-350 us per x-face beside the core launch of a 512^3 pair (49 us alone), the pack / unpack kernels
// 17-50 us each (profiles/r4_xface_timeline_before.txt).  Here the few columns next to the face are first GATHERED (one
// sector-per-row pass, independent loads) into strips -- one array of ny*nz doubles per column, element (j, k) at j + ny*k, so
// that the lanes of a wave read 512 contiguous bytes -- and both iterations of the shell cells, the planes that travel to the
// neighbour and the planes that arrive from it live in strips too:
//
//   turn     : (between two core launches, alone on the chip) the pair that ends: received CR -> field Hout column H, CS ->
//              column O, RS -> dHdtau column O; the pair that follows: columns H O I J of its level 0 -> strips AH AO AI AJ
//              (first pair of a chain: a plain gather, with Ht columns O I -> HTO HTI)
//   xstrip1  : level 1 of column O on every interior (j, k)                -> strip L1S  (sent as it is: no pack kernel)
//   xstrip2  : level 2 of column O, residual of the second iteration       -> strips CS (sent as it is), RS
//              (level 1 of O from L1S, of H from the received strip L1R, of I recomputed from level 0 exactly as the core does)
//
// H = the halo column (x = 0 or nx-1), O = the owned shell column next to it, I and J the two columns further inside.  Same
// diff3_point expression on the same operands as every other kernel of the path: bit-identical results.  The strips cover the
// whole interior (j, k) range of the column whatever other faces have neighbours (rows / planes that belong to a y- or z-shell
// box are computed twice, to the same bits; the norms count the peeled box only), so the plane sent to the x-neighbour is complete.
#pragma once
#include "diffusion3d_fused2.hpp"

enum { XS_AH = 0, XS_AO, XS_AI, XS_AJ, XS_HTO, XS_HTI, XS_L1S, XS_L1R, XS_CS, XS_CR, XS_RS, XS_COUNT };

struct Diff3StripFace {
    double* s;        // this face's strips: strip q at s + q * stride
    int xo;           // field column of the owned cells (1 or nx-2)
    int high;         // 0: H = xo-1, I = xo+1, J = xo+2;  1: H = xo+1, I = xo-1, J = xo-2
    int jlo, jhi;     // rows [jlo, jhi) and ...
    int klo, khi;     // ... planes [klo, khi) of the face's (peeled) shell box: the cells its norms count
};

struct Diff3StripArgs {
    Diff3StripFace f[2];
    const double* __restrict__ A;    // level 0 (Htau)
    const double* __restrict__ Ht;
    double* __restrict__ B;          // the reference's second work buffer: level-1 boundary / halo cells
    double* __restrict__ C;          // level 2 (Hout)
    double* __restrict__ dH;         // residual of the second iteration (nullptr: not stored)
    int nx, ny, nz;
    long stride;                     // doubles between two strips of a face
    double dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
    double scale;
    double* partials;                // [face][pcap] per-workgroup sums of the launch
    int pcap;
    int nyt, ntz;                    // 62-row tiles, XS_ZC-plane chunks
};

constexpr int XS_ZC = 8;             // planes per wave: every load of a wave is issued before its first store (one round trip)

// The strided part, kept OUT of the chain that runs beside the core launch (there every sector request of a strided kernel waits
// behind the core's traffic: gather 430 us, scatter 260 us on 16 units; alone on the chip a few us): one launch on the core stream
// BETWEEN two core launches.  Thread (j, k) of a face:
//   SCATTER (the pair that ends): interior (j, k): received CR -> field C column H, CS -> column O, RS -> dH column O;
//   then the strips of the NEXT pair's level 0: AH AO from CR CS (interior) or from the field (boundary rows / planes: cells a
//   y- / z-neighbour or the physical boundary owns), AI AJ from the field (the core launch wrote them);
//   HT: Ht columns O I as well (first pair of a chain; Ht does not change inside one).
// !SCATTER = the plain gather from the field `a.C` (first pair of a chain).  blockIdx.z = face.
typedef double xs_d2 __attribute__((ext_vector_type(2)));
template <bool SCATTER, bool HT>
__global__ __launch_bounds__(256) void k_xstrip_turn(Diff3StripArgs a)
{
    const int j = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y;
    if (j >= a.ny || k >= a.nz) return;
    const Diff3StripFace F = a.f[blockIdx.z];
    const int d = F.high ? -1 : 1;
    const size_t row = (size_t)a.nx * ((size_t)j + (size_t)a.ny * k);
    const size_t e = (size_t)j + (size_t)a.ny * k;
    double* __restrict__ C = a.C;
    const bool interior = j >= 1 && j <= a.ny - 2 && k >= 1 && k <= a.nz - 2;
    // columns (0,1) (2,3) / (nx-4,nx-3) (nx-2,nx-1) are aligned 16-byte pairs
    const bool pairs = (a.nx & 1) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    double h, o, i, jj;
    // inner pair first: the loads are what the launch waits for
    if (pairs) {
        const xs_d2 v = *reinterpret_cast<const xs_d2*>(C + row + (F.high ? F.xo - 2 : F.xo + 1));
        i = F.high ? v.y : v.x; jj = F.high ? v.x : v.y;
    } else {
        i = C[row + (F.xo + d)]; jj = C[row + (F.xo + 2 * d)];
    }
    double to = 0.0, ti = 0.0;
    if constexpr (HT) { to = a.Ht[row + F.xo]; ti = a.Ht[row + (F.xo + d)]; }
    if (SCATTER && interior) {
        h = F.s[XS_CR * a.stride + e];
        o = F.s[XS_CS * a.stride + e];
        if (pairs) {
            xs_d2 v;
            v.x = F.high ? o : h; v.y = F.high ? h : o;
            *reinterpret_cast<xs_d2*>(C + row + (F.high ? F.xo : F.xo - 1)) = v;
        } else {
            C[row + (F.xo - d)] = h;
            C[row + F.xo] = o;
        }
        if (a.dH) a.dH[row + F.xo] = F.s[XS_RS * a.stride + e];
    } else if (pairs) {
        const xs_d2 v = *reinterpret_cast<const xs_d2*>(C + row + (F.high ? F.xo : F.xo - 1));
        h = F.high ? v.y : v.x; o = F.high ? v.x : v.y;
    } else {
        h = C[row + (F.xo - d)]; o = C[row + F.xo];
    }
    F.s[XS_AH * a.stride + e] = h;
    F.s[XS_AO * a.stride + e] = o;
    F.s[XS_AI * a.stride + e] = i;
    F.s[XS_AJ * a.stride + e] = jj;
    if constexpr (HT) {
        F.s[XS_HTO * a.stride + e] = to;
        F.s[XS_HTI * a.stride + e] = ti;
    }
}

// received level 1 of the halo column (L1R) -> field B column H on the frame rows 1, ny-2 and planes 1, nz-2 only: the cells the
// fused launches on the y- / z-shell boxes read (everything else of that column is read from the strip)
__global__ __launch_bounds__(256) void k_xstrip_frame(Diff3StripArgs a)
{
    const int j = 1 + blockIdx.x * 64 + threadIdx.x, k = 1 + blockIdx.y * 4 + threadIdx.y;
    if (j >= a.ny - 1 || k >= a.nz - 1) return;
    if (!(j == 1 || j == a.ny - 2 || k == 1 || k == a.nz - 2)) return;
    const Diff3StripFace F = a.f[blockIdx.z];
    const int d = F.high ? -1 : 1;
    a.B[(size_t)a.nx * ((size_t)j + (size_t)a.ny * k) + (F.xo - d)] = F.s[XS_L1R * a.stride + (size_t)j + (size_t)a.ny * k];
}

// One wave per (62-row tile, XS_ZC-plane chunk) of a face; lane l holds row oly-1+l, lanes 1..62 own theirs; 4 waves per workgroup.
// LEVEL = 1: level 1 of column O (xstrip1);  LEVEL = 2: level 2 of column O (xstrip2).  blockIdx.y = face.
template <int LEVEL, bool NORM>
__global__ __launch_bounds__(256) void k_diff3_xstrip(Diff3StripArgs a)
{
    constexpr unsigned OOR = 0x7fffffffu;
    constexpr int ZC = XS_ZC;
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = blockIdx.y;
    const Diff3StripFace F = a.f[fi];
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long item = (long)blockIdx.x * 4 + wv;
    double acc = 0.0;
    if (item < (long)a.nyt * a.ntz) {
        const int ty = (int)(item % a.nyt), tz = (int)(item / a.nyt);
        const int j = ty * 62 + lane;                        // rows oly-1 .. oly+62, oly = 1 + 62*ty
        const bool row_in = j <= ny - 1;
        const bool row_bnd = j == 0 || j == ny - 1;
        const bool row_own = lane >= 1 && lane <= 62 && j <= ny - 2;
        const bool row_cnt = row_own && j >= F.jlo && j < F.jhi;
        const int k0 = 1 + tz * ZC;
        const int k1 = k0 + ZC < nz - 1 ? k0 + ZC : nz - 1;  // planes [k0, k1)
        const Diff3Coef cf{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};
        const unsigned sbytes = (unsigned)((long)ny * nz * 8);
        auto strip = [&](int q) { return diff3_rsrc((uintptr_t)(F.s + (long)q * a.stride), sbytes); };
        const int ps = ny * 8;                               // bytes between planes of a strip
        const unsigned vld = row_in ? (unsigned)(j * 8) : OOR;
        const unsigned vst = row_own ? (unsigned)(j * 8) : OOR;
        auto pl = [&](int k) { return (k >= 0 && k <= nz - 1 && k <= k1) ? k * ps : (int)OOR; };   // uniform
        const bool hi = F.high != 0;

        if constexpr (LEVEL == 1) {
            const __amdgpu_buffer_rsrc_t rH = strip(XS_AH), rO = strip(XS_AO), rI = strip(XS_AI), rT = strip(XS_HTO), rS = strip(XS_L1S);
            double O[ZC + 2], H[ZC], I[ZC], T[ZC];
#pragma unroll
            for (int q = 0; q < ZC + 2; ++q) O[q] = diff3_bld1(rO, vld, pl(k0 - 1 + q));
#pragma unroll
            for (int q = 0; q < ZC; ++q) {
                const int so = k0 + q < k1 ? (k0 + q) * ps : (int)OOR;
                H[q] = diff3_bld1(rH, vld, so);
                I[q] = diff3_bld1(rI, vld, so);
                T[q] = diff3_bld1(rT, vld, so);
            }
#pragma unroll
            for (int q = 0; q < ZC; ++q) {
                const int k = k0 + q;
                const double h = O[q + 1];
                const double ym = diff3_lane_up1(h), yp = diff3_lane_down1(h);
                double l1;
                const double r = diff3_point(h, hi ? I[q] : H[q], hi ? H[q] : I[q], ym, yp, O[q], O[q + 2], T[q], cf, l1);
                diff3_bst1(rS, vst, k < k1 ? k * ps : (int)OOR, l1);
                if constexpr (NORM) {
                    if (row_cnt && k < k1 && k >= F.klo && k < F.khi) { const double t = r * a.scale; acc += t * t; }
                }
            }
        } else {
            const __amdgpu_buffer_rsrc_t rL = strip(XS_L1S), rR = strip(XS_L1R), rO = strip(XS_AO), rI = strip(XS_AI), rJ = strip(XS_AJ),
                                         rTO = strip(XS_HTO), rTI = strip(XS_HTI), rC = strip(XS_CS), rD = strip(XS_RS);
            // level 1 of O on boundary rows / planes comes from the field B (physical boundary values, or the halo cells that
            // arrived from a y- / z-neighbour): descriptor based at plane k0-1 of the field
            const long fsz = (long)nx * ny * 8;
            const long fbase = (long)(k0 - 1) * fsz;
            const long frem = fsz * nz - fbase;
            const __amdgpu_buffer_rsrc_t rB = diff3_rsrc((uintptr_t)a.B + (uintptr_t)fbase, (unsigned)(frem > 0x7ffffff0L ? 0x7ffffff0L : frem));
            const unsigned vfb = row_in ? (unsigned)(((long)j * nx + F.xo) * 8) : OOR;
            double L[ZC + 2], I[ZC + 2], O[ZC], J[ZC], TO[ZC], TI[ZC], R[ZC];
#pragma unroll
            for (int q = 0; q < ZC + 2; ++q) {
                const int k = k0 - 1 + q;
                const bool pv = k <= k1;                                  // k1 <= nz-1
                const bool zb = k == 0 || k == nz - 1;
                const double fromS = diff3_bld1(rL, row_bnd ? OOR : vld, (pv && !zb) ? k * ps : (int)OOR);
                const double fromB = diff3_bld1(rB, (zb || row_bnd) ? vfb : OOR, pv ? (int)(q * fsz) : (int)OOR);
                L[q] = __longlong_as_double(__double_as_longlong(fromS) | __double_as_longlong(fromB));
                I[q] = diff3_bld1(rI, vld, pl(k));
            }
#pragma unroll
            for (int q = 0; q < ZC; ++q) {
                const int so = k0 + q < k1 ? (k0 + q) * ps : (int)OOR;
                O[q] = diff3_bld1(rO, vld, so);
                J[q] = diff3_bld1(rJ, vld, so);
                TO[q] = diff3_bld1(rTO, vld, so);
                TI[q] = diff3_bld1(rTI, vld, so);
                R[q] = diff3_bld1(rR, vld, so);
            }
#pragma unroll
            for (int q = 0; q < ZC; ++q) {
                const int k = k0 + q;
                // level 1 of column I, as the core launch computes it
                const double hi0 = I[q + 1];
                double l1i;
                (void)diff3_point(hi0, hi ? J[q] : O[q], hi ? O[q] : J[q], diff3_lane_up1(hi0), diff3_lane_down1(hi0), I[q], I[q + 2], TI[q], cf, l1i);
                const double h = L[q + 1];
                double h2;
                const double res = diff3_point(h, hi ? l1i : R[q], hi ? R[q] : l1i, diff3_lane_up1(h), diff3_lane_down1(h), L[q], L[q + 2], TO[q], cf, h2);
                const int so = k < k1 ? k * ps : (int)OOR;
                diff3_bst1(rC, vst, so, h2);
                diff3_bst1(rD, vst, so, res);
                if constexpr (NORM) {
                    if (row_cnt && k < k1 && k >= F.klo && k < F.khi) { const double t = res * a.scale; acc += t * t; }
                }
            }
        }
    }
    if constexpr (NORM) {
        const double w = diff3_wave_sum(acc);
        if (lane == 0) red[wv] = w;
        __syncthreads();
        if (tid == 0) a.partials[(size_t)fi * a.pcap + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}
