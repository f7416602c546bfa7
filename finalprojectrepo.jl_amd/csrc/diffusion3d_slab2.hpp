// diffusion3d_slab2.hpp -- TWO pseudo-transient iterations on a box that is only a few cells wide in x
// (same update, same operands, bit-identical to k_diff3_march2 and to two launches of k_diff3_march).
//
// k_diff3_march2 spends a whole 128-cell wave tile on whatever x-range it is given, so the one-cell-wide slab next to
// an x-neighbour of a decomposed run (role of @hide_communication's boundary width,
// part1_kernel_programming.jl:185-188) would cost as much as 128 columns.  This kernel serves boxes up to a few cells
// wide.  (It was also tried for the columns a row of full 126-cell tiles leaves over -- 510 = 4 x 126 + 6 at
// nx = 512, four tiles instead of five: 18 % fewer wave-instructions in the main kernel, and 10 % MORE time, because
// tile seams that are not on 64-byte sectors turn the two neighbours' stores into partial-sector writes;
// profiles/r2_diffusion_fused2_retile.txt.  Sector-aligned seams allow 120 owned cells per tile, i.e. five tiles.)
// Here the lanes of a wave run along y: lane l holds, for ITS row, the W owned columns plus two more on each side of
// level 0 (Htau) in registers, marches in z with three-plane windows of level 0, Ht and level 1 like the main kernel,
// takes y-neighbours from the adjacent lanes by DPP wave shifts and z-neighbours from the windows.  Level 1 (the field
// after the first iteration) is valid on lanes 1..62 and one column beyond the owned ones, level 2 on lanes 2..61:
// a wave owns 60 rows.  Domain-boundary cells of level 1 come from B (the reference's second work buffer), as in the
// main kernel.  Rows of one column are 8*nx bytes apart, so every wave-instruction touches 64 cache lines: this kernel
// trades coalescing for not wasting lanes, and is meant for <= ~3 % of the cells.
#pragma once
#include "diffusion3d_fused2.hpp"

// One 64-thread wave per (column group, 60-row y-tile, z-chunk); 4 independent waves per 256-thread workgroup.
template <bool NORM, int W>
__global__ __launch_bounds__(256) void k_diff3_slab2(Diff3Args2 a)
{
    constexpr int C0 = W + 4;   // level-0 columns xs-2 .. xs+W+1
    constexpr int C1 = W + 2;   // level-1 / Ht columns xs-1 .. xs+W
    constexpr unsigned OOR = 0x7fffffffu;
    __shared__ double red[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long sy = nx, sz = (long)nx * ny;
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    const int ncg = (wx + W - 1) / W, nyt = (wy + 59) / 60, ntz = (wz + a.zc - 1) / a.zc;
    const long item = (long)blockIdx.x * 4 + wv;
    const bool live = item < (long)ncg * nyt * ntz;
    double acc1 = 0.0, acc2 = 0.0;
    if (live) {
        const int cg = (int)(item % ncg), ty = (int)((item / ncg) % nyt), tz = (int)(item / ((long)ncg * nyt));
        const int xs = a.lo[0] + cg * W;
        const int xe = xs + W < a.hi[0] ? xs + W : a.hi[0];              // owned columns [xs, xe)
        const int oly = a.lo[1] + ty * 60;
        const int ohy = oly + 60 < a.hi[1] ? oly + 60 : a.hi[1];         // owned rows [oly, ohy): lanes 2 ..
        const int j = oly - 2 + lane;                                    // this lane's row
        const int jc = j < 0 ? 0 : (j > ny - 1 ? ny - 1 : j);
        const bool row_in = j >= 0 && j <= ny - 1;
        const bool row_bnd = j == 0 || j == ny - 1;                      // level 1 of this row comes from B
        const bool row_own = lane >= 2 && lane <= 61 && j >= oly && j < ohy;
        const int k0 = a.lo[2] + tz * a.zc;
        const int k1 = k0 + a.zc < a.hi[2] ? k0 + a.zc : a.hi[2];
        const int m0 = k0 - 1, m1 = k1;
        const Diff3Coef cf{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};

        // descriptors based at the first plane this chunk touches: num_records = bytes left in the array from there
        // (capped below the sentinel); chunk-relative plane offsets stay below 2 GiB (host: zc <= zc_max)
        const long array_bytes = sz * nz * 8;
        const long pbase = (long)(m0 > 0 ? m0 - 1 : 0) * sz * 8;
        auto mk = [&](const double* X) {
            const long rem = array_bytes - pbase;
            return diff3_rsrc((uintptr_t)X + (uintptr_t)pbase, (unsigned)(rem > 0x7ffffff0L ? 0x7ffffff0L : rem));
        };
        const bool wres = a.dH != nullptr;   // uniform; the residual store is dropped (sentinel offset) without a buffer
        const __amdgpu_buffer_rsrc_t rA = mk(a.A), rHt = mk(a.Ht), rB = mk(a.B), rC = mk(a.C), rD = mk(wres ? a.dH : a.C);
        // per-lane byte offset of (row, column xs-1) -- or the sentinel for rows outside the array (loads) / rows this
        // lane does not own (stores); columns add a compile-time constant, column and plane validity ride in the
        // scalar offset.  Column xs-2 (window index 0) gets an offset of its own: for xs = 1 it does not exist.
        const long rowoff1 = ((long)jc * sy + (xs - 1)) * 8;                 // >= 0
        const unsigned vld = row_in ? (unsigned)rowoff1 : OOR;               // window column 1
        const unsigned vld0 = (row_in && xs >= 2) ? (unsigned)(rowoff1 - 8) : OOR;   // window column 0
        const unsigned vst = row_own ? (unsigned)rowoff1 : OOR;
        auto poff = [&](int k) { return (int)((long)k * sz * 8 - pbase); };
        auto colL0 = [&](int c) { const int x = xs - 2 + c; return x >= 0 && x <= nx - 1 && x <= xe + 1; };   // uniform
        auto colL1 = [&](int c) { const int x = xs - 1 + c; return x >= 0 && x <= nx - 1 && x <= xe; };
        auto colB = [&](int c) { const int x = xs - 1 + c; return x == 0 || x == nx - 1; };

        double P[3][C0], HT[3][C1], Q[3][C1];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < C1; ++c) Q[q][c] = 0.0;
        auto loadP = [&](double (&dst)[C0], int k) {
            const bool pv = k >= 0 && k <= nz - 1;
#pragma unroll
            for (int c = 0; c < C0; ++c) dst[c] = diff3_bld1(rA, c == 0 ? vld0 : vld + 8u * (c - 1), (pv && colL0(c)) ? poff(k) : (int)OOR);
        };
        auto loadHt = [&](double (&dst)[C1], int k) {
            const bool pv = k >= 0 && k <= nz - 1;
#pragma unroll
            for (int c = 0; c < C1; ++c) dst[c] = diff3_bld1(rHt, vld + 8u * c, (pv && colL1(c)) ? poff(k) : (int)OOR);
        };
        // ring slots: plane p lives in slot (p - m0 + 1) % 3 for level 0 and Ht, level-1 plane p in slot (p - m0 + 2) % 3
        loadP(P[0], m0 - 1 < 0 ? 0 : m0 - 1);
        loadP(P[1], m0);
        loadP(P[2], m0 + 1);
        loadHt(HT[0], m0 - 1 < 0 ? 0 : m0 - 1);
        loadHt(HT[1], m0);
        loadHt(HT[2], m0 + 1);
        auto step = [&](auto Sc, int m) {
            constexpr int S = decltype(Sc)::value;   // (m - m0) % 3
            double(&Pm)[C0] = P[S % 3];
            double(&Pc)[C0] = P[(S + 1) % 3];
            double(&Pp)[C0] = P[(S + 2) % 3];
            double(&Qm)[C1] = Q[S % 3];              // level 1, plane m-2
            double(&Qc)[C1] = Q[(S + 1) % 3];        // plane m-1
            double(&Qn)[C1] = Q[(S + 2) % 3];        // plane m (written here)
            // ---- first step: level 1 on plane m ----
            const bool zb = m <= 0 || m >= nz - 1;
            const bool own_plane = m >= k0 && m < k1;
            double Bv[C1];
            {
                const int so = poff(m);
#pragma unroll
                for (int c = 0; c < C1; ++c) {   // boundary cells only; everything else is dropped by the range check
                    const bool need = colL1(c) && (zb || colB(c));                     // uniform part
                    Bv[c] = diff3_bld1(rB, (need || row_bnd) ? vld + 8u * c : OOR, colL1(c) ? so : (int)OOR);
                }
            }
#pragma unroll
            for (int c = 0; c < C1; ++c) {
                const double h = Pc[c + 1];
                const double ym = diff3_lane_up1(h), yp = diff3_lane_down1(h);
                double l1;
                const double r1 = diff3_point(h, Pc[c], Pc[c + 2], ym, yp, Pm[c + 1], Pp[c + 1], HT[(S + 1) % 3][c], cf, l1);
                if constexpr (NORM) {
                    if (own_plane && row_own && c >= 1 && c <= W && xs + c - 1 < xe) { const double t = r1 * a.scale; acc1 += t * t; }
                }
                Qn[c] = (zb || row_bnd || colB(c)) ? Bv[c] : l1;
            }
            // level-0 plane m-1 is dead: refill its slot with plane m+2 (needed up to m1 + 1)
            loadP(Pm, m + 2 <= m1 + 1 ? m + 2 : -1);
            // ---- second step: level 2 on plane m-1 (from level 1 of planes m-2, m-1, m) ----
            if (m - 1 >= k0) {
                const int so = poff(m - 1);
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    const double h = Qc[c + 1];
                    const double ym = diff3_lane_up1(h), yp = diff3_lane_down1(h);
                    double h2;
                    const double res = diff3_point(h, Qc[c], Qc[c + 2], ym, yp, Qm[c + 1], Qn[c + 1], HT[S % 3][c + 1], cf, h2);
                    const int soc = xs + c < xe ? so : (int)OOR;
                    diff3_bst1(rD, vst + 8u * (c + 1), wres ? soc : (int)OOR, res);
                    diff3_bst1(rC, vst + 8u * (c + 1), soc, h2);
                    if constexpr (NORM) {
                        if (row_own && xs + c < xe) { const double t = res * a.scale; acc2 += t * t; }
                    }
                }
            }
            // Ht plane m-1 is dead: refill with plane m+2
            loadHt(HT[S % 3], m + 2 <= m1 ? m + 2 : -1);
        };
        int m = m0;
        for (; m + 2 <= m1; m += 3) {
            step(std::integral_constant<int, 0>{}, m);
            step(std::integral_constant<int, 1>{}, m + 1);
            step(std::integral_constant<int, 2>{}, m + 2);
        }
        if (m <= m1) { step(std::integral_constant<int, 0>{}, m); ++m; }
        if (m <= m1) { step(std::integral_constant<int, 1>{}, m); ++m; }
    }
    if constexpr (NORM) {
        // fixed-order sum over the four waves of the workgroup
        const double w1 = diff3_wave_sum(acc1), w2 = diff3_wave_sum(acc2);
        if (lane == 0) { red[wv] = w1; red[4 + wv] = w2; }
        __syncthreads();
        if (tid == 0) {
            a.partials1[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
            a.partials2[blockIdx.x] = ((red[4] + red[5]) + red[6]) + red[7];
        }
    }
}

// Launch the slab kernel on [lo, hi) of `a` (hi[0] - lo[0] small).  *nparts = partials written per list.
static inline hipError_t diff3_launch_slab2(Diff3Args2 a, bool norm, hipStream_t stream, int max_partials, int* nparts)
{
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    *nparts = 0;
    if (wx <= 0 || wy <= 0 || wz <= 0) return hipSuccess;
    const long psb = (long)a.nx * a.ny * 8;
    const int zc_max = (int)((1L << 31) / psb) - 8;
    if (zc_max < 1) return hipErrorInvalidValue;
    const int W = wx <= 2 ? 2 : 3;   // owned columns per wave: more would not fit two waves per SIMD (W = 4: 245 VGPRs, 6: > 256)
    const long ncg = (wx + W - 1) / W, nyt = (wy + 59) / 60;
    // enough waves to spread over the chip (~4 per CU), chunks of at least 8 planes (each costs 2 extra iterations)
    int zc = wz;
    while (zc > 8 && ncg * nyt * ((wz + zc - 1) / zc) < 1024) zc = (zc + 1) / 2;
    if (zc > zc_max) zc = zc_max;
    a.zc = zc;
    const long items = ncg * nyt * ((wz + zc - 1) / zc);
    const long nblk = (items + 3) / 4;
    if (nblk > 0x7fffffffL || (norm && nblk > max_partials)) return hipErrorInvalidValue;
    if (W == 2) {
        if (norm) k_diff3_slab2<true, 2><<<(int)nblk, 256, 0, stream>>>(a);
        else k_diff3_slab2<false, 2><<<(int)nblk, 256, 0, stream>>>(a);
    } else {
        if (norm) k_diff3_slab2<true, 3><<<(int)nblk, 256, 0, stream>>>(a);
        else k_diff3_slab2<false, 3><<<(int)nblk, 256, 0, stream>>>(a);
    }
    *nparts = (int)nblk;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// ONE pseudo-transient iteration on a narrow box (the x-slab next to an x-neighbour: level 1 there has to exist in
// memory before the halo exchange of a fused pair, and the wave-tile kernel k_diff3_march would spend a 128-cell tile
// on it).  Same lane layout as k_diff3_slab2: lanes along y, W owned columns per lane, z-march with a three-plane
// window of Htau; no second level, so every lane but the two outermost owns its row (62 rows per wave).
// Writes Htau2 and dHdtau on the box, exactly like fpr_diffusion3d_step_box.
template <bool NORM, int W>
__global__ __launch_bounds__(256) void k_diff3_slab1(Diff3Args a)
{
    constexpr int C0 = W + 2;   // Htau columns xs-1 .. xs+W
    constexpr unsigned OOR = 0x7fffffffu;
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long sy = nx, sz = (long)nx * ny;
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    const int ncg = (wx + W - 1) / W, nyt = (wy + 61) / 62, ntz = (wz + a.zc - 1) / a.zc;
    const long item = (long)blockIdx.x * 4 + wv;
    double acc = 0.0;
    if (item < (long)ncg * nyt * ntz) {
        const int cg = (int)(item % ncg), ty = (int)((item / ncg) % nyt), tz = (int)(item / ((long)ncg * nyt));
        const int xs = a.lo[0] + cg * W;
        const int xe = xs + W < a.hi[0] ? xs + W : a.hi[0];
        const int oly = a.lo[1] + ty * 62;
        const int ohy = oly + 62 < a.hi[1] ? oly + 62 : a.hi[1];
        const int j = oly - 1 + lane;
        const int jc = j < 0 ? 0 : (j > ny - 1 ? ny - 1 : j);
        const bool row_in = j >= 0 && j <= ny - 1;
        const bool row_own = lane >= 1 && lane <= 62 && j >= oly && j < ohy;
        const int k0 = a.lo[2] + tz * a.zc;
        const int k1 = k0 + a.zc < a.hi[2] ? k0 + a.zc : a.hi[2];
        const Diff3Coef cf{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};
        const long array_bytes = sz * nz * 8;
        const long pbase = (long)(k0 - 1) * sz * 8;                          // k0 >= 1
        auto mk = [&](const double* X) {
            const long rem = array_bytes - pbase;
            return diff3_rsrc((uintptr_t)X + (uintptr_t)pbase, (unsigned)(rem > 0x7ffffff0L ? 0x7ffffff0L : rem));
        };
        const __amdgpu_buffer_rsrc_t rA = mk(a.Htau), rHt = mk(a.Ht), rC = mk(a.Htau2), rD = mk(a.dHdtau);
        const long rowoff = ((long)jc * sy + (xs - 1)) * 8;                 // column xs-1 >= 0
        const unsigned vld = row_in ? (unsigned)rowoff : OOR;
        const unsigned vst = row_own ? (unsigned)rowoff : OOR;
        auto poff = [&](int k) { return (int)((long)k * sz * 8 - pbase); };
        auto colv = [&](int c) { const int x = xs - 1 + c; return x <= nx - 1 && x <= xe; };   // uniform
        double P[3][C0];
        auto loadP = [&](double (&dst)[C0], int k) {
            const bool pv = k >= 0 && k <= nz - 1;
#pragma unroll
            for (int c = 0; c < C0; ++c) dst[c] = diff3_bld1(rA, vld + 8u * c, (pv && colv(c)) ? poff(k) : (int)OOR);
        };
        loadP(P[0], k0 - 1);
        loadP(P[1], k0);
        loadP(P[2], k0 + 1);
        auto step = [&](auto Sc, int k) {
            constexpr int S = decltype(Sc)::value;   // (k - k0) % 3: plane k-1 in slot S, k in S+1, k+1 in S+2
            double(&Pm)[C0] = P[S % 3];
            double(&Pc)[C0] = P[(S + 1) % 3];
            double(&Pp)[C0] = P[(S + 2) % 3];
            double ht[W];
            const int so = poff(k);
#pragma unroll
            for (int c = 0; c < W; ++c) ht[c] = diff3_bld1(rHt, vld + 8u * (c + 1), xs + c < xe ? so : (int)OOR);
#pragma unroll
            for (int c = 0; c < W; ++c) {
                const double h = Pc[c + 1];
                const double ym = diff3_lane_up1(h), yp = diff3_lane_down1(h);
                double h2;
                const double r = diff3_point(h, Pc[c], Pc[c + 2], ym, yp, Pm[c + 1], Pp[c + 1], ht[c], cf, h2);
                const int soc = xs + c < xe ? so : (int)OOR;
                diff3_bst1(rD, vst + 8u * (c + 1), soc, r);
                diff3_bst1(rC, vst + 8u * (c + 1), soc, h2);
                if constexpr (NORM) {
                    if (row_own && xs + c < xe) { const double t = r * a.scale; acc += t * t; }
                }
            }
            loadP(Pm, k + 2 <= k1 ? k + 2 : -1);   // plane k-1 is dead: refill with plane k+2
        };
        int k = k0;
        for (; k + 2 < k1; k += 3) {
            step(std::integral_constant<int, 0>{}, k);
            step(std::integral_constant<int, 1>{}, k + 1);
            step(std::integral_constant<int, 2>{}, k + 2);
        }
        if (k < k1) { step(std::integral_constant<int, 0>{}, k); ++k; }
        if (k < k1) { step(std::integral_constant<int, 1>{}, k); ++k; }
    }
    if constexpr (NORM) {
        const double w1 = diff3_wave_sum(acc);
        if (lane == 0) red[wv] = w1;
        __syncthreads();
        if (tid == 0) a.partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

static inline hipError_t diff3_launch_slab1(Diff3Args a, bool norm, hipStream_t stream, int max_partials, int* nparts)
{
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    *nparts = 0;
    if (wx <= 0 || wy <= 0 || wz <= 0) return hipSuccess;
    const long psb = (long)a.nx * a.ny * 8;
    const int zc_max = (int)((1L << 31) / psb) - 8;
    if (zc_max < 1) return hipErrorInvalidValue;
    const int W = wx <= 2 ? 2 : 4;
    const long ncg = (wx + W - 1) / W, nyt = (wy + 61) / 62;
    int zc = wz;
    while (zc > 8 && ncg * nyt * ((wz + zc - 1) / zc) < 1024) zc = (zc + 1) / 2;
    if (zc > zc_max) zc = zc_max;
    a.zc = zc;
    const long nblk = (ncg * nyt * ((wz + zc - 1) / zc) + 3) / 4;
    if (nblk > 0x7fffffffL || (norm && nblk > max_partials)) return hipErrorInvalidValue;
    if (W == 2) {
        if (norm) k_diff3_slab1<true, 2><<<(int)nblk, 256, 0, stream>>>(a);
        else k_diff3_slab1<false, 2><<<(int)nblk, 256, 0, stream>>>(a);
    } else {
        if (norm) k_diff3_slab1<true, 4><<<(int)nblk, 256, 0, stream>>>(a);
        else k_diff3_slab1<false, 4><<<(int)nblk, 256, 0, stream>>>(a);
    }
    *nparts = (int)nblk;
    return hipGetLastError();
}
