// diffusion3d_launch.hpp -- host-side variant dispatch for the fused diffusion kernels.
// Shared by diffusion3d.hip (library) and tools/diffusion_tune.hip (tuning harness).
#pragma once
#include <cstdint>

#include "diffusion3d_kernels.hpp"

struct Diff3Tuning {
    int variant = 0;     // 0 = library default; 1 = naive; 2 = z-march (registers + shuffles);
                         // 3 = z-march with LDS exchange of the inter-wave halo rows;
                         // 4 / 5 = 2 / 3 with the ring-pipelined z-loop (16-byte path, ry 2 or 4; else falls back)
    int zc = 0;          // planes per z-chunk (0 = auto)
    int xcd_remap = -1;  // -1 auto, 0 off, 1 on
    int ry = 0;          // rows per lane (0 = auto; 1, 2 or 4)
    int nt = -1;         // non-temporal stores: -1 auto, 0 off, 1 on
    int vx = 0;          // cells per lane in x (0 = auto; 1 or 2)
};

#ifndef DIFF3_DEFAULT_VARIANT_ID
#define DIFF3_DEFAULT_VARIANT_ID 5
#endif
#ifndef DIFF3_DEFAULT_RY
#define DIFF3_DEFAULT_RY 4
#endif
#ifndef DIFF3_DEFAULT_NT
#define DIFF3_DEFAULT_NT 1
#endif
#ifndef DIFF3_DEFAULT_XCD
#define DIFF3_DEFAULT_XCD 0
#endif
#ifndef DIFF3_TARGET_BLOCKS
#define DIFF3_TARGET_BLOCKS 4096
#endif

template <int VX, int RY, bool LDSY, bool NT>
static inline void diff3_march_go(const Diff3Args& a, bool norm, int nblk, hipStream_t stream)
{
    if (norm) k_diff3_march<VX, RY, true, LDSY, NT><<<nblk, 256, 0, stream>>>(a);
    else k_diff3_march<VX, RY, false, LDSY, NT><<<nblk, 256, 0, stream>>>(a);
}

// ring-pipelined form (PIPE = true), instantiated for the 16-byte path with 2 or 4 rows per lane
template <int RY>
static inline void diff3_pipe_go(const Diff3Args& a, bool norm, bool ldsy, bool nt, int nblk, hipStream_t stream)
{
#define DIFF3_PIPE_CASE(N_, L_, T_)                                                              \
    if (norm == N_ && ldsy == L_ && nt == T_) {                                                    \
        k_diff3_march<2, RY, N_, L_, T_, true><<<nblk, 256, 0, stream>>>(a);                        \
        return;                                                                                    \
    }
    DIFF3_PIPE_CASE(false, false, false) DIFF3_PIPE_CASE(false, false, true) DIFF3_PIPE_CASE(false, true, false)
    DIFF3_PIPE_CASE(false, true, true) DIFF3_PIPE_CASE(true, false, false) DIFF3_PIPE_CASE(true, false, true)
    DIFF3_PIPE_CASE(true, true, false) DIFF3_PIPE_CASE(true, true, true)
#undef DIFF3_PIPE_CASE
}

template <int VX, int RY>
static inline void diff3_march_go2(const Diff3Args& a, bool norm, bool ldsy, bool nt, int nblk, hipStream_t stream)
{
    if (ldsy) {
        if (nt) diff3_march_go<VX, RY, true, true>(a, norm, nblk, stream);
        else diff3_march_go<VX, RY, true, false>(a, norm, nblk, stream);
    } else {
        if (nt) diff3_march_go<VX, RY, false, true>(a, norm, nblk, stream);
        else diff3_march_go<VX, RY, false, false>(a, norm, nblk, stream);
    }
}

// Launches the selected variant on `stream`.  *nparts = number of block partials written (norm only).
static inline hipError_t diff3_launch(Diff3Args a, bool norm, const Diff3Tuning& t, hipStream_t stream,
                                      int max_partials, int* nparts)
{
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    int variant = t.variant ? t.variant : DIFF3_DEFAULT_VARIANT_ID;
    *nparts = 0;
    if (wx <= 0 || wy <= 0 || wz <= 0) return hipSuccess;
#ifndef FPR_TUNE
    // the library: the default form only (variant 5 = ring-pipelined march with the LDS row exchange, non-temporal stores; 2 cells per
    // lane where nx is even and the arrays are 16-byte aligned, else 1; 4, 2 or 1 rows per lane by the box's height)
    if (t.variant || t.zc || t.xcd_remap >= 0 || t.ry || t.nt >= 0 || t.vx) return hipErrorInvalidValue;
#else
    if (variant == 1) {
        const dim3 grid((wx + 63) / 64, (wy + 3) / 4, wz);
        const size_t nb = (size_t)grid.x * grid.y * grid.z;
        if (norm && nb > (size_t)max_partials) return hipErrorInvalidValue;
        if (norm) k_diff3_naive<true><<<grid, dim3(64, 4, 1), 0, stream>>>(a);
        else k_diff3_naive<false><<<grid, dim3(64, 4, 1), 0, stream>>>(a);
        *nparts = (int)nb;
        return hipGetLastError();
    }
#endif
    if (variant < 2 || variant > 5) return hipErrorInvalidValue;
    bool pipe = variant >= 4;  // 4 = variant 2 + ring pipeline, 5 = variant 3 + ring pipeline
    const bool ldsy = (variant == 3 || variant == 5);
    // 16-byte path needs even nx and 16-byte aligned arrays
    const bool can16 = (a.nx % 2 == 0) &&
                       ((((uintptr_t)a.Ht | (uintptr_t)a.Htau | (uintptr_t)a.Htau2 | (uintptr_t)a.dHdtau) & 15) == 0);
    int vx = t.vx ? t.vx : 2;
    if (!can16 || wx < 32) vx = 1;
    int ry = t.ry ? t.ry : DIFF3_DEFAULT_RY;
    if (ry != 1 && ry != 2 && ry != 4) return hipErrorInvalidValue;
    while (ry > 1 && wy < ry * 2) ry >>= 1;
    const int txw = 64 * vx;
    const int xorg = a.lo[0] & ~(vx - 1);
    a.ntx = (a.hi[0] - xorg + txw - 1) / txw;
    a.nty = (wy + ry - 1) / ry;
    const long tiles_xy = ldsy ? (long)a.ntx * ((a.nty + 3) / 4) : ((long)a.ntx * a.nty + 3) / 4;  // blocks per chunk
    int zc = t.zc;
    if (zc <= 0) {
        long want = (DIFF3_TARGET_BLOCKS + tiles_xy - 1) / tiles_xy;  // chunks wanted
        if (want < 1) want = 1;
        zc = (int)((wz + want - 1) / want);
        if (zc < 8) zc = wz < 8 ? wz : 8;
    }
    if (zc > wz) zc = wz;
    a.zc = zc;
    a.ntz = (wz + zc - 1) / zc;
    long nblk;
    if (ldsy) nblk = (long)a.ntx * ((a.nty + 3) / 4) * a.ntz;
    else nblk = ((long)a.ntx * a.nty * a.ntz + 3) / 4;
    if (nblk > 0x7fffffffL || (norm && nblk > max_partials)) return hipErrorInvalidValue;
    a.xcd_remap = t.xcd_remap < 0 ? (nblk >= 64 ? DIFF3_DEFAULT_XCD : 0) : t.xcd_remap;
    if (a.xcd_remap >= 2) {  // y-band ownership needs LDSY blocks and (block rows / G) divisible by 8
        const int G = 1 << (a.xcd_remap - 2);
        const int nby = (a.nty + 3) / 4;
        if (!ldsy || a.xcd_remap > 6 || nby % (8 * G) != 0) a.xcd_remap = 0;
    }
    const bool nt = t.nt < 0 ? DIFF3_DEFAULT_NT : (t.nt != 0);
    if (a.fma && pipe && vx == 2 && ry == 4 && ldsy && nt) {
        // opt-in contracted arithmetic (option fp_contract): instantiated for the default form of the march only; every other
        // geometry runs the exact kernel below
        if (norm) k_diff3_march<2, 4, true, true, true, true, true><<<(int)nblk, 256, 0, stream>>>(a);
        else k_diff3_march<2, 4, false, true, true, true, true><<<(int)nblk, 256, 0, stream>>>(a);
    }
#ifndef FPR_TUNE
    else if (vx == 2 && ry == 4) { if (norm) k_diff3_march<2, 4, true, true, true, true><<<(int)nblk, 256, 0, stream>>>(a); else k_diff3_march<2, 4, false, true, true, true><<<(int)nblk, 256, 0, stream>>>(a); }
    else if (vx == 2 && ry == 2) { if (norm) k_diff3_march<2, 2, true, true, true, true><<<(int)nblk, 256, 0, stream>>>(a); else k_diff3_march<2, 2, false, true, true, true><<<(int)nblk, 256, 0, stream>>>(a); }
    else if (vx == 2) diff3_march_go<2, 1, true, true>(a, norm, (int)nblk, stream);
    else if (ry == 4) diff3_march_go<1, 4, true, true>(a, norm, (int)nblk, stream);
    else if (ry == 2) diff3_march_go<1, 2, true, true>(a, norm, (int)nblk, stream);
    else diff3_march_go<1, 1, true, true>(a, norm, (int)nblk, stream);
#else
    else if (pipe && vx == 2 && (ry == 4 || ry == 2)) {
        if (ry == 4) diff3_pipe_go<4>(a, norm, ldsy, nt, (int)nblk, stream);
        else diff3_pipe_go<2>(a, norm, ldsy, nt, (int)nblk, stream);
    } else if (vx == 2) {
        if (ry == 4) diff3_march_go2<2, 4>(a, norm, ldsy, nt, (int)nblk, stream);
        else if (ry == 2) diff3_march_go2<2, 2>(a, norm, ldsy, nt, (int)nblk, stream);
        else diff3_march_go2<2, 1>(a, norm, ldsy, nt, (int)nblk, stream);
    } else {
        if (ry == 4) diff3_march_go2<1, 4>(a, norm, ldsy, nt, (int)nblk, stream);
        else if (ry == 2) diff3_march_go2<1, 2>(a, norm, ldsy, nt, (int)nblk, stream);
        else diff3_march_go2<1, 1>(a, norm, ldsy, nt, (int)nblk, stream);
    }
#endif
    *nparts = (int)nblk;
    return hipGetLastError();
}
