// diffusion3d_kernels.hpp -- gfx950 kernels for the fused 7-point pseudo-transient diffusion update
// (reference: scripts-part1/part1_kernel_programming.jl:46-58, flux macros :12-20).
//
// Shared by libfpr_hip.so (diffusion3d.hip) and the tuning harness (tools/diffusion_tune.hip).
// All kernels compute, for interior cells inside the box [lo, hi):
//     r      = (((qx+ - qx-)*_dx + (qy+ - qy-)*_dy) + (qz+ - qz-)*_dz) + (H - Ht)*_dt
//     dHdtau = r ;  Htau2 = H - dtau*r
// with q+ = -D_d*(H[+1]-H), q- = -D_d*(H-H[-1]); expression order as the reference so that, built
// with -ffp-contract=off, results are bit-identical to the CPU expressions.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

struct Diff3Args {
    const double* __restrict__ Ht;
    const double* __restrict__ Htau;
    double* __restrict__ Htau2;
    double* __restrict__ dHdtau;
    int nx, ny, nz;
    int lo[3], hi[3];  // box, already clipped to the interior [1, n-1)
    double dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
    double scale;       // norm: sum((r*scale)^2)
    double* partials;   // one double per block (NORM kernels)
    int zc;             // planes per z-chunk (marching kernels)
    int ntx, nty, ntz;  // tile counts (marching kernels)
    int xcd_remap;      // 1: give each XCD a contiguous range of tiles
    int fma = 0;        // 1: the contracted form of diff3_point where an instantiation exists (option fp_contract)
};

struct Diff3Coef {
    double dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
};

// FMA = false (the default and the only form parity claims are made for): every operation rounds as the reference's expression
// does on a CPU (the library is built with -ffp-contract=off).  FMA = true (option fp_contract = 1, opt-in): the contraction the
// reference itself notes -- "(or 1 * fma)", part1_kernel_programming.jl:55,94 -- written out, so that it is the SAME sequence on
// every compiler and in the oracle's orc_diffusion3d_step_fma: 18 instead of 25 FP64 instructions per cell and iteration, results
// within 1e-12 relative of the exact form (tests/test_gpu_part1.py).
template <bool FMA = false>
__device__ __forceinline__ double diff3_point(double h, double xm, double xp, double ym, double yp, double zm,
                                              double zp, double ht, const Diff3Coef& c, double& h2)
{
    if constexpr (FMA) {
        const double qxm = -c.D_dx * (h - xm), qym = -c.D_dy * (h - ym), qzm = -c.D_dz * (h - zm);
        const double fx = __builtin_fma(-c.D_dx, xp - h, -qxm);      // qxp - qxm
        const double fy = __builtin_fma(-c.D_dy, yp - h, -qym);
        const double fz = __builtin_fma(-c.D_dz, zp - h, -qzm);
        double r = fx * c._dx;
        r = __builtin_fma(fy, c._dy, r);
        r = __builtin_fma(fz, c._dz, r);
        r = __builtin_fma(h - ht, c._dt, r);
        h2 = __builtin_fma(-c.dtau, r, h);
        return r;
    }
    const double qxp = -c.D_dx * (xp - h), qxm = -c.D_dx * (h - xm);
    const double qyp = -c.D_dy * (yp - h), qym = -c.D_dy * (h - ym);
    const double qzp = -c.D_dz * (zp - h), qzm = -c.D_dz * (h - zm);
    const double r = (((qxp - qxm) * c._dx + (qyp - qym) * c._dy) + (qzp - qzm) * c._dz) + (h - ht) * c._dt;
    h2 = h - c.dtau * r;
    return r;
}

// neighbour-lane moves by DPP wave shifts (see fpr_internal.hpp); edge lanes keep their own value
__device__ __forceinline__ double diff3_lane_up1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double diff3_lane_down1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double diff3_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a memory fence, for
// which hipcc emits `s_waitcnt vmcnt(0)`: inside the z-march that would drain the prefetched global loads
// of the next planes at every plane.  Here only the LDS counter is waited on, so global loads issued
// before the barrier stay in flight across it.
__device__ __forceinline__ void diff3_lds_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) alone (gfx9 encoding: vmcnt and expcnt fields left at max)
    __builtin_amdgcn_s_barrier();
}

// fixed-order block sum for 256-thread blocks; result in thread 0
__device__ __forceinline__ double diff3_block_sum256(double v, double* red, int tid)
{
    v = diff3_wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (tid == 0) ? ((red[0] + red[1]) + red[2]) + red[3] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// V0: one thread per cell, 7 global loads (cache-served reuse).  Baseline / correctness anchor.
// block (64,4,1), grid (ceil(wx/64), ceil(wy/4), wz)
// ------------------------------------------------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(256) void k_diff3_naive(Diff3Args a)
{
    __shared__ double red[4];
    const int i = a.lo[0] + blockIdx.x * 64 + threadIdx.x;
    const int j = a.lo[1] + blockIdx.y * 4 + threadIdx.y;
    const int k = a.lo[2] + blockIdx.z;
    double acc = 0.0;
    if (i < a.hi[0] && j < a.hi[1] && k < a.hi[2]) {
        const size_t sx = 1, sy = (size_t)a.nx, sz = (size_t)a.nx * a.ny;
        const size_t id = (size_t)i + sy * j + sz * k;
        const Diff3Coef c{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};
        double h2;
        const double r = diff3_point(a.Htau[id], a.Htau[id - sx], a.Htau[id + sx], a.Htau[id - sy], a.Htau[id + sy],
                                     a.Htau[id - sz], a.Htau[id + sz], a.Ht[id], c, h2);
        a.dHdtau[id] = r;
        a.Htau2[id] = h2;
        if constexpr (NORM) { const double t = r * a.scale; acc = t * t; }
    }
    if constexpr (NORM) {
        const int tid = threadIdx.x + 64 * threadIdx.y;
        const double s = diff3_block_sum256(acc, red, tid);
        if (tid == 0) a.partials[blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Z-marching kernels (variants 2 and 3).
//
// A wave owns a tile of 64*VX cells in x (VX consecutive cells per lane: VX = 2 -> one 16-byte
// global_load_dwordx4 per lane and row, 1 KiB per wave-instruction) by RY rows in y, and marches
// through a chunk of z-planes keeping planes k-1, k, k+1 (and the prefetched k+2) in registers:
//   * z-neighbours        : registers (each Htau value is loaded once per chunk)
//   * y-neighbours        : registers inside the tile; the two halo rows come either from global
//                           memory (LDSY = false: neighbouring waves' rows, L1/L2 hits) or, inside a
//                           workgroup of 4 waves stacked in y, from an LDS exchange of the waves'
//                           first/last rows (LDSY = true, one s_barrier per plane, 2 LDS buffers)
//   * x-neighbours        : wavefront shuffles of the lane-adjacent registers; only lanes 0 and 63
//                           fetch the cell beyond the tile edge (one predicated 8-byte load per row)
// Loads for plane k+2 / halos of plane k+1 are issued before plane k is computed, so every load has a
// full iteration of slack; nothing in the loop waits on a just-issued load.
// Tiles start at an even x index so that 16-byte accesses are aligned when nx is even.
// ------------------------------------------------------------------------------------------------
template <int VX>
struct DVec;
template <>
struct DVec<1> {
    double v[1];
};
template <>
struct DVec<2> {
    double v[2];
};

template <int VX>
__device__ __forceinline__ DVec<VX> diff3_ldv(const double* __restrict__ p)
{
    DVec<VX> o;
    if constexpr (VX == 2) {
        const double2 t = *reinterpret_cast<const double2*>(p);
        o.v[0] = t.x;
        o.v[1] = t.y;
    } else {
        o.v[0] = *p;
    }
    return o;
}

// PIPE = true: the z-pipeline is written as a ring of 4 register sets with compile-time roles (the loop
// is unrolled by 4), so no register of an in-flight load is ever copied: the loads of plane k+3 (and of
// the halo rows / Ht of plane k+2) are issued right after plane k has been computed, into the registers
// plane k-1 occupied, and stay in flight for two full iterations -- across the LDS barrier as well.
// (With PIPE = false the rotation `zm=c; c=zp; zp=zpp` and the merge of the LDS / global halo rows make
// hipcc wait for the just-issued loads: `s_waitcnt vmcnt(0)` once per plane.)
template <int VX, int RY, bool NORM, bool LDSY, bool NT, bool PIPE = false, bool FMA = false>
__global__ __launch_bounds__(256) void k_diff3_march(Diff3Args a)
{
    constexpr int TXW = 64 * VX;  // tile width in cells
    __shared__ double red[4];
    // LDS exchange rows: [parity][wave][first/last][TXW]
    __shared__ __attribute__((aligned(16))) double xrow[LDSY ? 2 * 4 * 2 * TXW : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;

    // ---- block -> tile mapping (optionally XCD-contiguous) ----
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if (a.xcd_remap == 1) {
        const int q = nblk >> 3, rem = nblk & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        bid = xcd * q + (xcd < rem ? xcd : rem) + slot;
    }
    int tx, ty, tz;
    bool wave_active;
    if constexpr (LDSY) {
        // block = 4 waves stacked in y on the same x-tile / z-chunk
        const int nby = (a.nty + 3) >> 2;
        int by;
        if (a.xcd_remap >= 2) {
            // y-band ownership: groups of G = 2^(xcd_remap-2) consecutive block-rows are dealt round-robin
            // to the 8 XCDs, so all x-tiles (and G block-rows) of a band share one L2 (host checks divisibility)
            const int G = 1 << (a.xcd_remap - 2);
            const int ng8 = (nby / G) >> 3;  // groups per XCD
            const int xcd = blockIdx.x & 7;
            int r = blockIdx.x >> 3;
            tx = r % a.ntx; r /= a.ntx;
            const int byl = r % G; r /= G;
            const int gl = r % ng8;
            tz = r / ng8;
            by = (gl * 8 + xcd) * G + byl;
        } else {
            tx = bid % a.ntx;
            by = (bid / a.ntx) % nby;
            tz = bid / (a.ntx * nby);
        }
        ty = by * 4 + w;
        wave_active = true;  // all waves run the loop (barriers); rows are masked
    } else {
        const long wt = (long)bid * 4 + w;
        const long ntile = (long)a.ntx * a.nty * a.ntz;
        wave_active = wt < ntile;
        const long wtc = wave_active ? wt : 0;
        tx = (int)(wtc % a.ntx);
        ty = (int)((wtc / a.ntx) % a.nty);
        tz = (int)(wtc / ((long)a.ntx * a.nty));
    }

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    const int xorg = a.lo[0] & ~(VX - 1);
    const int ib = xorg + tx * TXW + lane * VX;        // first cell of this lane
    const int ibc = ib > nx - VX ? nx - VX : ib;       // clamped for loads
    const int j0 = a.lo[1] + ty * RY;
    const int k0 = a.lo[2] + tz * a.zc;
    const int k1 = (k0 + a.zc < a.hi[2]) ? k0 + a.zc : a.hi[2];

    bool cm[VX];
#pragma unroll
    for (int v = 0; v < VX; ++v) cm[v] = (ib + v >= a.lo[0]) && (ib + v < a.hi[0]);
    int jr[RY];       // clamped row index per register row
    bool rm[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        rm[r] = (j0 + r < a.hi[1]);
        jr[r] = (j0 + r < ny - 1) ? j0 + r : ny - 1;
    }
    const int jd = j0 - 1;  // >= 0 because lo[1] >= 1
    const int ju = (j0 + RY < ny - 1) ? j0 + RY : ny - 1;
    // edge cell beyond the tile in x: lane 0 -> left, lane 63 -> right (clamped into the row)
    const bool is_edge = (lane == 0) || (lane == 63);
    int ie = (lane == 0) ? ib - 1 : ib + VX;
    ie = ie < 0 ? 0 : (ie > nx - 1 ? nx - 1 : ie);

    const Diff3Coef cf{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};
    const double* __restrict__ H = a.Htau;

    DVec<VX> zm[RY], c[RY], zp[RY], zpp[RY], htc[RY], htn[RY];
    DVec<VX> ydc, yuc, ydn, yun;
    double ec[RY], en[RY];
    double acc = 0.0;

    auto kcl = [&](int k) { return k > nz - 1 ? nz - 1 : k; };
    const bool need_gd = !LDSY || (w == 0);   // bottom halo row from global
    const bool need_gu = !LDSY || (w == 3);   // top halo row from global

    if constexpr (PIPE) {
        const int kendp = LDSY ? k1 : (wave_active ? k1 : k0);
        DVec<VX> P[4][RY];          // planes (k-1, k, k+1, k+2) in slots ((k-k0)+{0,1,2,3}) & 3
        DVec<VX> HT[2][RY];         // Ht of planes k, k+1 in slots (k-k0) & 1
        DVec<VX> YD[2], YU[2];      // global halo rows of planes k, k+1
        double ED[2][RY];           // tile-edge cells of planes k, k+1
        auto load_plane = [&](DVec<VX>(&dst)[RY], int k) {
#pragma unroll
            for (int r = 0; r < RY; ++r) dst[r] = diff3_ldv<VX>(H + (size_t)ibc + sy * jr[r] + sz * kcl(k));
        };
        auto load_aux = [&](DVec<VX>(&ht)[RY], DVec<VX>& yd, DVec<VX>& yu, double (&e)[RY], int k) {
            const int kc = kcl(k);
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                ht[r] = diff3_ldv<VX>(a.Ht + (size_t)ibc + sy * jr[r] + sz * kc);
                e[r] = is_edge ? H[(size_t)ie + sy * jr[r] + sz * kc] : 0.0;
            }
            if (need_gd) yd = diff3_ldv<VX>(H + (size_t)ibc + sy * jd + sz * kc);
            if (need_gu) yu = diff3_ldv<VX>(H + (size_t)ibc + sy * ju + sz * kc);
        };
        if (kendp > k0) {
            load_plane(P[0], k0 - 1);
            load_plane(P[1], k0);
            load_plane(P[2], k0 + 1);
            load_plane(P[3], k0 + 2);
            load_aux(HT[0], YD[0], YU[0], ED[0], k0);
            load_aux(HT[1], YD[1], YU[1], ED[1], k0 + 1);
        }
        auto step = [&](auto Sc, int k) {
            constexpr int S = decltype(Sc)::value;
            DVec<VX>(&zmR)[RY] = P[S & 3];
            DVec<VX>(&cR)[RY] = P[(S + 1) & 3];
            DVec<VX>(&zpR)[RY] = P[(S + 2) & 3];
            constexpr int hsl = S & 1;
            DVec<VX> ydv = YD[hsl], yuv = YU[hsl];
            if constexpr (LDSY) {
                double* buf = xrow + (size_t)(k & 1) * (4 * 2 * TXW);
                double* mine = buf + (size_t)w * (2 * TXW) + lane * VX;
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    mine[v] = cR[0].v[v];
                    mine[TXW + v] = cR[RY - 1].v[v];
                }
                diff3_lds_barrier();
                const double* od = buf + (size_t)(w > 0 ? w - 1 : 0) * (2 * TXW) + TXW + lane * VX;  // last row of the wave below
                const double* ou = buf + (size_t)(w < 3 ? w + 1 : 3) * (2 * TXW) + lane * VX;        // first row of the wave above
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    const double ld = od[v], lu = ou[v];
                    ydv.v[v] = (w > 0) ? ld : ydv.v[v];
                    yuv.v[v] = (w < 3) ? lu : yuv.v[v];
                }
            }
            double res[RY][VX], h2[RY][VX];
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
                    const double ym = (r == 0) ? ydv.v[v] : cR[r == 0 ? 0 : r - 1].v[v];
                    const double yp = (r == RY - 1) ? yuv.v[v] : cR[r == RY - 1 ? r : r + 1].v[v];
                    res[r][v] = diff3_point<FMA>(cR[r].v[v], xm, xp, ym, yp, zmR[r].v[v], zpR[r].v[v], HT[hsl][r].v[v], cf, h2[r][v]);
                }
            }
            // plane k-1, and the halo rows / Ht of plane k, are dead now: refill them (planes k+3 and k+2)
            if (k + 1 < kendp) {
                load_plane(P[S & 3], k + 3);
                load_aux(HT[hsl], YD[hsl], YU[hsl], ED[hsl], k + 2);
            }
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                if (rm[r]) {
                    const size_t id = (size_t)ib + sy * (size_t)(j0 + r) + sz * (size_t)k;
                    if constexpr (VX == 2) {
                        if (cm[0] && cm[1]) {
                            if constexpr (NT) {
                                typedef double d2v __attribute__((ext_vector_type(2)));
                                d2v rv, hv;
                                rv.x = res[r][0]; rv.y = res[r][1];
                                hv.x = h2[r][0]; hv.y = h2[r][1];
                                __builtin_nontemporal_store(rv, reinterpret_cast<d2v*>(a.dHdtau + id));
                                __builtin_nontemporal_store(hv, reinterpret_cast<d2v*>(a.Htau2 + id));
                            } else {
                                *reinterpret_cast<double2*>(a.dHdtau + id) = make_double2(res[r][0], res[r][1]);
                                *reinterpret_cast<double2*>(a.Htau2 + id) = make_double2(h2[r][0], h2[r][1]);
                            }
                        } else {
                            if (cm[0]) { a.dHdtau[id] = res[r][0]; a.Htau2[id] = h2[r][0]; }
                            if (cm[1]) { a.dHdtau[id + 1] = res[r][1]; a.Htau2[id + 1] = h2[r][1]; }
                        }
                    } else {
                        if (cm[0]) { a.dHdtau[id] = res[r][0]; a.Htau2[id] = h2[r][0]; }
                    }
                    if constexpr (NORM) {
#pragma unroll
                        for (int v = 0; v < VX; ++v)
                            if (cm[v]) { const double t = res[r][v] * a.scale; acc += t * t; }
                    }
                }
            }
        };
        int k = k0;
        for (; k + 3 < kendp; k += 4) {
            step(std::integral_constant<int, 0>{}, k);
            step(std::integral_constant<int, 1>{}, k + 1);
            step(std::integral_constant<int, 2>{}, k + 2);
            step(std::integral_constant<int, 3>{}, k + 3);
        }
        if (k < kendp) { step(std::integral_constant<int, 0>{}, k); ++k; }
        if (k < kendp) { step(std::integral_constant<int, 1>{}, k); ++k; }
        if (k < kendp) { step(std::integral_constant<int, 2>{}, k); ++k; }
        if constexpr (NORM) {
            const double s = diff3_block_sum256(acc, red, tid);
            if (tid == 0) a.partials[blockIdx.x] = s;
        }
        return;
    }

    if (wave_active && k0 < k1) {
        // ---- prologue: planes k0-1, k0, k0+1 ----
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const size_t rowoff = (size_t)ibc + sy * jr[r];
            zm[r] = diff3_ldv<VX>(H + rowoff + sz * (k0 - 1));
            c[r] = diff3_ldv<VX>(H + rowoff + sz * k0);
            zp[r] = diff3_ldv<VX>(H + rowoff + sz * kcl(k0 + 1));
            htc[r] = diff3_ldv<VX>(a.Ht + rowoff + sz * k0);
            ec[r] = is_edge ? H[(size_t)ie + sy * jr[r] + sz * k0] : 0.0;
        }
        if (need_gd) ydc = diff3_ldv<VX>(H + (size_t)ibc + sy * jd + sz * k0);
        if (need_gu) yuc = diff3_ldv<VX>(H + (size_t)ibc + sy * ju + sz * k0);
    }

    const int kend = LDSY ? k1 : (wave_active ? k1 : k0);
    for (int k = k0; k < kend; ++k) {
        // ---- issue next loads: centre of plane k+2, halos/edges/Ht of plane k+1 ----
        const int kn = kcl(k + 1), knn = kcl(k + 2);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const size_t rowoff = (size_t)ibc + sy * jr[r];
            zpp[r] = diff3_ldv<VX>(H + rowoff + sz * knn);
            htn[r] = diff3_ldv<VX>(a.Ht + rowoff + sz * kn);
            en[r] = is_edge ? H[(size_t)ie + sy * jr[r] + sz * kn] : 0.0;
        }
        if (need_gd) ydn = diff3_ldv<VX>(H + (size_t)ibc + sy * jd + sz * kn);
        if (need_gu) yun = diff3_ldv<VX>(H + (size_t)ibc + sy * ju + sz * kn);

        if constexpr (LDSY) {
            // exchange first/last rows of plane k between the 4 waves through LDS
            double* buf = xrow + (size_t)(k & 1) * (4 * 2 * TXW);
            double* mine = buf + (size_t)w * (2 * TXW) + lane * VX;
#pragma unroll
            for (int v = 0; v < VX; ++v) {
                mine[v] = c[0].v[v];
                mine[TXW + v] = c[RY - 1].v[v];
            }
            diff3_lds_barrier();
            if (w > 0) {
                const double* o = buf + (size_t)(w - 1) * (2 * TXW) + TXW + lane * VX;  // last row of wave below
#pragma unroll
                for (int v = 0; v < VX; ++v) ydc.v[v] = o[v];
            }
            if (w < 3) {
                const double* o = buf + (size_t)(w + 1) * (2 * TXW) + lane * VX;        // first row of wave above
#pragma unroll
                for (int v = 0; v < VX; ++v) yuc.v[v] = o[v];
            }
        }

        // ---- compute plane k ----
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            // x-neighbours by wavefront shuffle; edge lanes use the fetched edge cell
            const double fromL = diff3_lane_up1(c[r].v[VX - 1]);
            const double fromR = diff3_lane_down1(c[r].v[0]);
            const double xl0 = (lane == 0) ? ec[r] : fromL;
            const double xrL = (lane == 63) ? ec[r] : fromR;
            double res[VX], h2[VX];
#pragma unroll
            for (int v = 0; v < VX; ++v) {
                const double xm = (v == 0) ? xl0 : c[r].v[v - 1];
                const double xp = (v == VX - 1) ? xrL : c[r].v[v + 1];
                const double ym = (r == 0) ? ydc.v[v] : c[r == 0 ? 0 : r - 1].v[v];
                const double yp = (r == RY - 1) ? yuc.v[v] : c[r == RY - 1 ? r : r + 1].v[v];
                res[v] = diff3_point<FMA>(c[r].v[v], xm, xp, ym, yp, zm[r].v[v], zp[r].v[v], htc[r].v[v], cf, h2[v]);
            }
            if (rm[r]) {
                const size_t id = (size_t)ib + sy * (size_t)(j0 + r) + sz * (size_t)k;
                if constexpr (VX == 2) {
                    if (cm[0] && cm[1]) {
                        if constexpr (NT) {
                            typedef double d2v __attribute__((ext_vector_type(2)));
                            d2v rv, hv;
                            rv.x = res[0]; rv.y = res[1];
                            hv.x = h2[0]; hv.y = h2[1];
                            __builtin_nontemporal_store(rv, reinterpret_cast<d2v*>(a.dHdtau + id));
                            __builtin_nontemporal_store(hv, reinterpret_cast<d2v*>(a.Htau2 + id));
                        } else {
                            *reinterpret_cast<double2*>(a.dHdtau + id) = make_double2(res[0], res[1]);
                            *reinterpret_cast<double2*>(a.Htau2 + id) = make_double2(h2[0], h2[1]);
                        }
                    } else {
                        if (cm[0]) { a.dHdtau[id] = res[0]; a.Htau2[id] = h2[0]; }
                        if (cm[1]) { a.dHdtau[id + 1] = res[1]; a.Htau2[id + 1] = h2[1]; }
                    }
                } else {
                    if (cm[0]) {
                        if constexpr (NT) {
                            __builtin_nontemporal_store(res[0], a.dHdtau + id);
                            __builtin_nontemporal_store(h2[0], a.Htau2 + id);
                        } else {
                            a.dHdtau[id] = res[0];
                            a.Htau2[id] = h2[0];
                        }
                    }
                }
                if constexpr (NORM) {
#pragma unroll
                    for (int v = 0; v < VX; ++v)
                        if (cm[v]) { const double t = res[v] * a.scale; acc += t * t; }
                }
            }
        }

        // ---- rotate the register pipeline ----
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            zm[r] = c[r];
            c[r] = zp[r];
            zp[r] = zpp[r];
            htc[r] = htn[r];
            ec[r] = en[r];
        }
        if (need_gd) ydc = ydn;
        if (need_gu) yuc = yun;
    }

    if constexpr (NORM) {
        const double s = diff3_block_sum256(acc, red, tid);
        if (tid == 0) a.partials[blockIdx.x] = s;
    }
}
