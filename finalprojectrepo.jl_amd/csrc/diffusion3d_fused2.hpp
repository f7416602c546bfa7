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
// Geometry (wave64, 2 cells per lane, 4 rows per lane, NW = 8 or 4 waves stacked in y = block tile 128 x 4*NW):
//   * the block marches in z; iteration m computes L1 on plane m for its whole tile (from a ring of 3 register
//     planes of L0: m-1, m, m+1, refilled one iteration ahead), then L2 on plane m-1 from the three L1 planes
//     m-2, m-1, m held in a second ring; Ht has its own ring of 3 (both steps need it);
//   * x-neighbours by DPP wave shifts for both levels; y-neighbours between the waves through ONE LDS exchange +
//     raw s_barrier per iteration that carries the first/last rows of L0(m) and of L1(m-1);
//   * L1 is valid on the whole tile (L0 halo cells / rows come from global memory as in k_diff3_march), L2 on the
//     tile shrunk by one cell: tiles overlap by 2 in x and y and chunks by 2 planes in z (redundant L1 work:
//     32/30 or 16/14 in y, (zc+2)/zc in z; x uses 5 tiles of ~102 owned cells for a 512-cell line);
//   * domain-boundary cells of L1 are taken from B: the registers that would hold the (non-existent) L0 halo
//     beyond the boundary carry the B value instead, so the steady-state loop has no extra loads;
//   * <= 256 registers, so every SIMD holds two waves (NW = 8: one 512-thread workgroup per CU; NW = 4: two).
// Requirements (the caller falls back to two single-step launches otherwise): nx even and >= 128, all
// arrays 16-byte aligned, ny >= 16, nz >= 3, nx * ny * 96 < 2^31.
#pragma once
#include "diffusion3d_kernels.hpp"

// Wait states after the stores of a row (s_nop N = N+1 states; -1 = none, harness only).  Round 1 found corrupted
// upper halves of 16-byte store data with a VALU write 1-2 instructions behind the store and used s_nop 7.  Round 2
// soak (tools/hazard_soak.hip: 200-400 launches each at 512^3, 768^3, 640x200x300, every cell compared, second stream
// copying; profiles/r2_hazard_soak.txt): with today's instruction order -- the two 16-byte stores of a row are
// followed by its two 8-byte stores, pinned by sched_barrier -- every count from none to 7 is clean.  Kept: s_nop 1 =
// the two wait states the gfx9 rule asks for when the SGPR-offset exemption is NOT assumed.
#ifndef DIFF3_STORE_NOP
#define DIFF3_STORE_NOP 1
#endif
// 1: lanes whose pair of cells lies outside what the tile needs (level 0 on [ol - 2, oh + 1]) are switched OFF for the whole march
// (EXEC) instead of re-reading the nearest needed pair and computing on it: 5 x-tiles of 128 cells cover a 510-cell line, so about
// a sixth of the lanes of every vector instruction works on nothing -- at the same issue rate, but the kernel runs at the card's
// power cap (profiles/r4_power_probe.json) and idle lanes draw less.  (Diff3Args2::lane_off selects it at run time.)
#ifndef DIFF3_LANE_OFF
#define DIFF3_LANE_OFF 1
#endif

// Every workgroup of a ticketed launch passes here once, behind its other counter updates; the last one leaves the counters at zero
// for the next launch.
__device__ __forceinline__ void diff3_ticket_leave(int* ticket, int nwg)
{
    if (atomicAdd(ticket + 8, 1) == nwg - 1)
        for (int i = 0; i < 11; ++i) __hip_atomic_store(ticket + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef FPR_CU_KEY_DEFINED
#define FPR_CU_KEY_DEFINED
// key of the compute unit the calling wave runs on: XCC_ID (3 bits) | HW_ID[15:8] = SE_ID, SH_ID, CU_ID
__device__ __forceinline__ unsigned fpr_cu_key()
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
}
#endif

struct Diff3Args2 {
    const double* __restrict__ Ht;
    const double* __restrict__ A;   // L0 (Htau)
    const double* __restrict__ B;   // boundary values of L1 (the reference's Htau2); interior never read
    double* __restrict__ C;         // L2 (interior of the box)
    double* __restrict__ dH;        // residual of the second step (nullptr: not stored -- its norm is still reduced)
    int nx, ny, nz;
    int lo[3], hi[3];               // output box, clipped to the interior [1, n-1)
    double dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
    double scale;
    double* partials1;              // per-block sum((r1*scale)^2) over owned cells (first step)
    double* partials2;              // same for the second step
    int zc, ntx, nby, ntz, sx;      // planes per chunk, tile counts, owned cells per tile in x
    int nw;                         // waves per workgroup (4, or 8: 32-row blocks)
    int xalign;                     // 1: x-tile cut points sit in the middle of 128-byte lines
    int zb_lo, zb_hi, ntz_a;        // optional second z-range [zb_lo, zb_hi) with the same x/y box: chunks tz >= ntz_a
    int xcd_remap;
    int fma = 0;                    // 1: the contracted form of diff3_point (option fp_contract; plain grids of 4 or 8 waves only)
    // reserved form (k_diff3_march2<..., BAL = true>; decomposed runs that leave compute units to the halo exchange): the
    // launch has G = gridDim.x workgroups where the plain grid has G + bal_r (tile, chunk) units.  Workgroup g first serves
    // unit g like the plain grid (same tiles marching in lockstep, same block -> XCD mapping), then a slice of bal_q planes
    // of one of the bal_r left-over units: unit G + g / bal_sp, slice g % bal_sp.
    int bal_r, bal_sp, bal_q;
    // Tickets (reserved form on a device whose compute units are split between streams, fpr_reserve_comm_cus): the launch has one
    // workgroup per device slot, MORE than its bal_g units; a workgroup takes a ticket from the counter of the XCD it runs on
    // (then, if that XCD's share of the units is taken, from the others') and the ones that find none leave at once.
    // Whichever slots the stream's CU mask takes away -- the dispatcher deals a workgroup to a shader engine and lets it wait
    // THERE -- the workgroups that cannot be placed are then the ones without work, so the mask needs no symmetry (8 or 16
    // units instead of 32), and every unit is served whatever the placement.  The last workgroup to leave zeroes the counters.
    int* ticket;                    // 8 unit counters + 1 exit counter, all zero between launches; nullptr: unit = blockIdx.x, bal_g = gridDim.x
    int bal_g;                      // units served by the launch's workgroups (G)
    // On a device split by fpr_reserve_comm_cus the core stream has EVERY unit and the launch one workgroup per unit: the ones that
    // land on a unit of the comm stream (bit fpr_cu_key() of this map, found by a probe launch) leave at once and keep it free.
    // Placement then cannot leave a working unit without a workgroup, whatever share the comm stream has (8, 16, ... units).
    const unsigned* reserved;
    const int* skip;                // fpr_diffusion3d_solve with pairs enqueued ahead of its exit test: return at once if *skip (nullptr = unconditional)
    int lane_off;                   // 1: lanes outside the cells the tile needs are switched off for the whole march (DIFF3_LANE_OFF)
#ifdef FPR_TUNE
    int dbg;                        // tuning harness only (tools/, -DFPR_TUNE): 1 = drop all stores, 2 = drop all loads of the z-loop
#endif
};

// ablation switches exist in the tuning harness only; the production library compiles them out
#ifdef FPR_TUNE
#define DIFF3_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define DIFF3_DBG(a, bit) false
#endif

// DPP wave shifts without an `old` operand: the edge lane receives 0 (no register copy needed)
__device__ __forceinline__ double diff3_lane_up1_z(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double diff3_lane_down1_z(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// DPP wave shifts whose edge lane (lane 0 for "up", lane 63 for "down") receives `edge` instead of a neighbour
__device__ __forceinline__ double diff3_lane_up1_edge(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double diff3_lane_down1_edge(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Buffer addressing: address = descriptor base (4 SGPRs, loop-invariant) + per-lane byte offset (1 VGPR,
// loop-invariant) + scalar byte offset (1 SGPR: plane and row).  No 64-bit address arithmetic per access.
// Every descriptor's num_records is the number of bytes from its base to the END OF THE ARRAY (capped below the
// sentinel), so the hardware range check bounds every access to the array whatever the guards in the code say; an
// access is dropped on purpose by giving it the sentinel offset OOR > num_records, in the VGPR or in the SGPR
// offset: gfx950 compares voffset + soffset against num_records for raw (stride 0) descriptors, measured by
// tools/oob_probe.hip (profiles/r2_oob_probe.txt) -- LLVM's documentation lists soffset as unchecked.
typedef double diff3_d2v __attribute__((ext_vector_type(2)));
typedef unsigned diff3_u4v __attribute__((ext_vector_type(4)));
typedef unsigned diff3_u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t diff3_rsrc(uintptr_t p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ DVec<2> diff3_bld2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    const diff3_d2v t = __builtin_bit_cast(diff3_d2v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    DVec<2> o;
    o.v[0] = t.x;
    o.v[1] = t.y;
    return o;
}
__device__ __forceinline__ double diff3_bld1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void diff3_bst2_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, double x, double y)
{
    diff3_d2v t;
    t.x = x;
    t.y = y;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(diff3_u4v, t), r, voff, soff, 2 /* nt */);
}
__device__ __forceinline__ void diff3_bst1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, double x)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(diff3_u2v, x), r, voff, soff, 0);
}

// Every global load and store inside the z-loop is issued UNCONDITIONALLY: rows, planes and lanes that must not
// be touched get an out-of-range buffer offset instead (the hardware range check drops the access without memory
// traffic).  With conditional memory instructions hipcc cannot count the operations that are younger than a
// prefetch and falls back to `s_waitcnt vmcnt(0)` -- draining the prefetched planes at every iteration.
// NW = waves per workgroup, stacked in y (block tile 128 x 4*NW rows): 4 (two workgroups per CU) or 8 (one).
// fixed-order block sum over NW waves; result in thread 0
template <int NW>
__device__ __forceinline__ double diff3_block_sum_waves(double v, double* red, int tid)
{
    v = diff3_wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[i];
    }
    return s;
}

// WRES = false: the residual of the second step is not written to memory (a solver loop that only needs its norm:
// 24 instead of 32 bytes per cell and launch, and two stores less per row)
// BAL = true: the grid is SHORT of the plain (tile, chunk) grid by a few units (device slots minus the compute units left
// to RCCL and the shell launches of a decomposed run); every workgroup serves its own unit and then a thin slice of one of
// the left-over units, so the launch still finishes in one balanced round and its tiles still march in lockstep.
template <bool NORM, int NW = 4, bool WRES = true, bool BAL = false, bool FMA = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void k_diff3_march2(Diff3Args2 a)
{
    // NW = 1: ONE wave per workgroup, a 128 x 4 tile whose first or last row is a y-boundary (or y-halo) row -- the
    // one-row shell next to a y-neighbour of a decomposed run.  Level 1 of the row next to the boundary needs level 0 of
    // rows inside the tile only, and the owned row is that one: the wave needs a single global halo row (the boundary row
    // of level 1, from B); the level-0 halo on the far side is never read into anything that is stored.  A 16-row block
    // (NW = 4) for one owned row did four times the work.
    constexpr int VX = 2, RY = 4, TXW = 128, SYB = NW * RY - 2;
    constexpr int NR = 3;                         // ring length = loop unroll: <= 256 registers, two workgroups per CU
    constexpr int SLOT = 4 * TXW;                 // doubles per wave slot: L0 first row, L0 last row, L1 first row, L1 last row
    constexpr unsigned OOR = 0x7fffffffu;         // offset beyond every descriptor's num_records
    __shared__ double red[2 * NW];
    // [parity][slot 0..5][row kind][TXW]; wave w owns slot w+1, slot 0 / 5 receive the global halo rows of
    // the bottom / top wave, so every wave reads "the slot below" and "the slot above" without a select
    __shared__ __attribute__((aligned(16))) double xrow[2 * (NW + 2) * SLOT];

    if (a.skip && *a.skip) return;   // a pair enqueued behind the iteration that ended the solver's loop (diffusion3d.hip)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: rows, halo sources and row masks stay scalar

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    double tot1 = 0.0, tot2 = 0.0;   // per-lane sums over the owned cells of all chunks of this workgroup (NORM)

    // unit of this workgroup and number of units the launch's workgroups serve
    int unit = blockIdx.x, nunits = gridDim.x;
    if (BAL && a.ticket) {
        // counters: [0, 8) units taken per XCD group, [8] workgroups that are through, [9] units taken, [10] workgroups that have
        // started and -- unless they sit on a unit of the comm stream -- made their claim
        __shared__ int tk;
        nunits = a.bal_g;
        if (tid == 0) {
            const int q = nunits >> 3, rem = nunits & 7;
            const unsigned key = fpr_cu_key(), xcc = key >> 8;
            const bool comm_unit = a.reserved && ((a.reserved[key >> 5] >> (key & 31)) & 1u);
            auto ld = [&](int i) { return __hip_atomic_load(a.ticket + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            if (comm_unit) {
                // A workgroup on a unit of the comm stream leaves as soon as every unit of work has an owner (a few microseconds
                // into the launch).  It takes one itself only if every workgroup has started and claimed and work is still left
                // (the dispatcher placed two workgroups on one unit one after the other), or after ~50 us without either.
                atomicAdd(a.ticket + 10, 1);
                for (int spin = 0; spin < 48; ++spin) {
                    if (ld(9) >= nunits || ld(10) >= (int)gridDim.x) break;
                    __builtin_amdgcn_s_sleep(32);
                }
            }
            int u = -1;
            if (!comm_unit || ld(9) < nunits)
                for (int t = 0; t < 8 && u < 0; ++t) {      // own XCD first: its units are x/y neighbours marching in lockstep on one L2
                    const int g = (int)((xcc + t) & 7), share = q + (g < rem ? 1 : 0);
                    if ((t > 0 || comm_unit) && ld(g) >= share) continue;   // (the own group's counter is not asked first: one round trip)
                    const int k = atomicAdd(a.ticket + g, 1);
                    if (k < share) u = g + 8 * k;           // the position blockIdx has in a grid of nunits workgroups
                }
            // (results not used: the wave does not wait for these two)
            if (u >= 0) __hip_atomic_fetch_add(a.ticket + 9, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!comm_unit) __hip_atomic_fetch_add(a.ticket + 10, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (u < 0) diff3_ticket_leave(a.ticket, (int)gridDim.x);   // (a workgroup with work does this at its end)
            tk = u;
        }
        __syncthreads();
        unit = __builtin_amdgcn_readfirstlane(tk);          // (block-uniform: keep it scalar)
        if (unit < 0) return;                               // no work left for this workgroup
    }
    constexpr int NITEM = BAL ? 2 : 1;
#pragma unroll 1
    for (int item = 0; item < NITEM; ++item) {
    int tx, by, k0, k1;
    int slice = -1;
    int bid = unit;
    if (BAL && item == 1) {
        const int j = bid / a.bal_sp;
        if (j >= a.bal_r) break;                           // block-uniform: no left-over slice for this workgroup
        slice = bid - j * a.bal_sp;
        bid = nunits + j;
    } else if (a.xcd_remap == 1) {
        const int nblk = nunits;
        const int q = nblk >> 3, rem = nblk & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        bid = xcd * q + (xcd < rem ? xcd : rem) + slot;
    }
    tx = bid % a.ntx;
    by = (bid / a.ntx) % a.nby;
    int tz = bid / (a.ntx * a.nby);
    if (!BAL && a.xcd_remap == 2) {
        // z-chunk ownership: XCD x (= blockIdx % 8, the hardware's round-robin) processes the chunks tz = x, x+8, ...
        // one after the other, so the workgroups resident on an XCD are x/y neighbours of ONE chunk (host: ntz % 8 == 0)
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int per = a.ntx * a.nby;
        tz = xcd + 8 * (l / per);
        const int rem = l % per;
        tx = rem % a.ntx;
        by = rem / a.ntx;
    }
    // z: owned planes [k0, k1)
    const bool zsecond = tz >= a.ntz_a;          // chunk of the second z-range (two thin boxes in one launch)
    const int zlo = zsecond ? a.zb_lo : a.lo[2], zhi = zsecond ? a.zb_hi : a.hi[2];
    k0 = zlo + (zsecond ? tz - a.ntz_a : tz) * a.zc;
    k1 = (k0 + a.zc < zhi) ? k0 + a.zc : zhi;
    if (BAL && item == 1) {
        k0 += slice * a.bal_q;
        k1 = (k0 + a.bal_q < k1) ? k0 + a.bal_q : k1;
        if (k1 <= k0) break;                               // block-uniform
        __syncthreads();                                   // the first unit's last LDS rows are still being read
    }

    // ---- x: owned output cells [ol, oh) (cut points between tiles are even); own cells [s, s+128) ----
    const int e0 = a.lo[0] & ~1;
    // cut points between x-tiles: e0 + t*sx, moved (xalign) to the middle of a 128-byte line (cell 16k+8): a seam then
    // costs one shared line instead of two
    auto cut = [&](int t) {
        const int c = e0 + t * a.sx;
        return a.xalign ? (c & ~15) + 8 : c;
    };
    const int olr = tx == 0 ? a.lo[0] : cut(tx), ohr = tx == a.ntx - 1 ? a.hi[0] : cut(tx + 1);
    const int ol = olr > a.lo[0] ? olr : a.lo[0];
    const int oh = ohr < a.hi[0] ? ohr : a.hi[0];
    int s = (ol - 1) & ~1;
    s = s < nx - TXW ? s : nx - TXW;                 // the x-boundary cell nx-1, if in the tile, belongs to lane 63
    const int ib = s + lane * VX;
    int ilast = (oh + 1) & ~1;                       // last pair that is needed (L0 at oh+1)
    ilast = ilast < nx - 2 ? ilast : nx - 2;
    int ifirst = (ol - 2) & ~1;                      // first pair that is needed (L0 at ol-2)
    ifirst = ifirst > 0 ? ifirst : 0;
    const int ibc = ib < ifirst ? ifirst : (ib < ilast ? ib : ilast);   // lanes outside re-read the nearest needed pair
    const unsigned voff = (unsigned)ibc * 8u;        // per-lane byte offset inside a row
    const bool xb_tile = (s == 0) || (s + TXW == nx);   // uniform: the tile holds an x-boundary cell
    const bool bndL = (ib == 0);                     // own cell v=0 is the x-boundary (lane 0 of tile 0)
    const bool bndR = (ib + 1 == nx - 1);            // own cell v=1 is the x-boundary (lane 63 of the last tile)
    // edge register: lane 0 / 63 fetch the L0 cell beyond the tile; the lane holding an x-boundary cell fetches that
    // cell's B value instead.  All other lanes issue the (unconditional) load with an out-of-range offset: nothing is
    // fetched -- re-reading an own cell looked free but missed the L2 often enough to add 20 % to the read traffic.
    int ie = (lane == 0) ? ib - 1 : ib + VX;
    ie = ie < ifirst ? ifirst : (ie > ilast + 1 ? ilast + 1 : ie);
    const unsigned eoff = ((lane == 0 || lane == 63) && !(bndL || bndR)) ? (unsigned)ie * 8u : OOR;   // from A
    const unsigned boff = bndL ? 0u : (bndR ? (unsigned)(nx - 1) * 8u : OOR);                     // from B

    // ---- y: owned rows [oly, ohy); block rows y1 .. y1+15 ----
    const int oly = a.lo[1] + by * SYB;
    const int ohy = (oly + SYB < a.hi[1]) ? oly + SYB : a.hi[1];
    const int y1 = (oly - 1 < ny - NW * RY) ? oly - 1 : ny - NW * RY;
    const int j0 = y1 + w * RY;
    const bool bb = (w == 0) && (y1 == 0);                    // own row 0 of wave 0 is the y-boundary
    const bool bt = (w == NW - 1) && (y1 + NW * RY - 1 == ny - 1);  // own last row of the top wave is the y-boundary
    const int jd = bb ? 0 : (j0 > 0 ? j0 - 1 : 0);
    const int ju = bt ? ny - 1 : (j0 + RY < ny - 1 ? j0 + RY : ny - 1);
    const bool hwave = (w == 0) || (w == NW - 1);             // waves that own a global halo row
    // (NW = 1: the wave is bottom and top wave at once and owns ONE halo row: the y-boundary row of level 1 on the side
    // of the tile that touches the boundary -- host: the tile touches one)
    const bool top1 = NW == 1 && bt && !bb;
    const double* Hsrc = (w == 0 && !top1) ? (bb ? a.B : a.A) : (bt ? a.B : a.A);
    const int hrow = (w == 0 && !top1) ? jd : ju;

    // ---- z: owned planes [k0, k1) (above); iterations m0 .. m1 ----
    const int m0 = k0 - 1, m1 = k1;

    bool cm[VX], rm[RY];
#pragma unroll
    for (int v = 0; v < VX; ++v) cm[v] = (ib + v >= ol) && (ib + v < oh);
#pragma unroll
    for (int r = 0; r < RY; ++r) rm[r] = (j0 + r >= oly) && (j0 + r < ohy);   // uniform
    const bool has_split = ((ol | oh) & 1) != 0;     // uniform: some lane owns only one cell of its pair
    // per-lane store offsets: 16-byte store if the lane owns its pair, 8-byte store of the one owned cell otherwise
    const unsigned sv4 = (cm[0] && cm[1]) ? (unsigned)ib * 8u : OOR;
    const unsigned sv2 = (cm[0] != cm[1]) ? (unsigned)(ib + (cm[1] ? 1 : 0)) * 8u : OOR;

    const Diff3Coef cf{a.dtau, a._dt, a._dx, a._dy, a._dz, a.D_dx, a.D_dy, a.D_dz};
    auto kcl = [&](int k) { return k < 0 ? 0 : (k > nz - 1 ? nz - 1 : k); };

#if DIFF3_LANE_OFF
    // (all waves of a workgroup share the tile's x-range, so every wave keeps the same lanes -- at least one -- and still arrives
    // at every barrier; a neighbour value that DPP would take from a switched-off lane only feeds level 1 at the cells ifirst /
    // ilast + 1, which nothing owned depends on)
    if (!a.lane_off || (ib >= ifirst && ib <= ilast)) {
#endif
    // ---- buffer descriptors: base = array + (pbA + plane shift) planes + first row; every access of iteration m
    //      uses the scalar offsets so + r * rs with so = (m + 2 - pbA) * ps (see the shifts below) ----
    const int ps = (int)(sz * 8), rs = (int)(sy * 8);   // plane / row stride in bytes (host: (zc + 8) * ps < 2^31)
    const int pbA = m0 - 1 > 0 ? m0 - 1 : 0;
    const long array_bytes = (long)sz * (long)nz * 8;
    // descriptor based at (plane pbA + pshift, row): num_records = bytes left in the array from there (a base before the
    // array's first byte -- pshift < 0 in the first chunk -- only makes the window longer, never past the array's end)
    auto rsrc_at = [&](const double* X, int pshift, int row) -> __amdgpu_buffer_rsrc_t {
        const long off = ((long)(pbA + pshift) * (long)sz + (long)row * (long)sy) * 8;
        long rem = array_bytes - off;
        rem = rem < 0 ? 0 : (rem > 0x7ffffff0L ? 0x7ffffff0L : rem);
        return diff3_rsrc((uintptr_t)X + (uintptr_t)off, (unsigned)rem);
    };
    const __amdgpu_buffer_rsrc_t rA = rsrc_at(a.A, 0, j0);       // L0 plane m+2
    const __amdgpu_buffer_rsrc_t rHt = rsrc_at(a.Ht, 0, j0);     // Ht plane m+2
    const __amdgpu_buffer_rsrc_t rEB = rsrc_at(a.B, -1, j0);     // B x-boundary cells, plane m+1
    const __amdgpu_buffer_rsrc_t rH = rsrc_at(Hsrc, -1, hrow);   // halo row, plane m+1
    const __amdgpu_buffer_rsrc_t rC = rsrc_at(a.C, -NR, j0);     // L2 plane m-1
    const __amdgpu_buffer_rsrc_t rD = rsrc_at(WRES ? a.dH : a.C, -NR, j0);   // (unused without WRES)

    DVec<VX> P[NR][RY];       // L0 planes m-1, m, m+1; plane p lives in slot (p - m0 + 1) % NR
    DVec<VX> HT[NR][RY];      // Ht planes m-1, m, m+1 (in flight), same slot rule
    DVec<VX> Q[NR][RY];       // L1 planes m-2, m-1, m, same slot rule
    DVec<VX> YH;              // global L0 halo row of plane m (bottom / top wave; the B boundary row if bb / bt)
    double ED[RY];            // L0 tile-edge cells of plane m (or the B boundary cell, see bndL / bndR)
    double acc1[VX], acc2[VX];
#pragma unroll
    for (int v = 0; v < VX; ++v) { acc1[v] = 0.0; acc2[v] = 0.0; }
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int v = 0; v < VX; ++v) Q[q][r].v[v] = 0.0;

    // soff = scalar offset of (plane, row 0) relative to the descriptor in use (OOR: nothing is fetched, zeros return)
    auto row_off = [&](int soff, int r) { return soff == (int)OOR ? (int)OOR : soff + r * rs; };
    auto load_rows = [&](DVec<VX>(&dst)[RY], __amdgpu_buffer_rsrc_t rsrc, int soff) {
#pragma unroll
        for (int r = 0; r < RY; ++r) dst[r] = diff3_bld2(rsrc, voff, row_off(soff, r));
    };
    // soff: plane offset for the descriptors based one plane earlier (rEB, rH); soffA: the same plane relative to rA
    auto load_halo = [&](DVec<VX>& yh, double (&e)[RY], int soff, int soffA) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const double ea = diff3_bld1(rA, eoff, row_off(soffA, r));   // 0 for the lane that holds an x-boundary cell
            const double eb = diff3_bld1(rEB, boff, row_off(soff, r));   // 0 for every other lane
            e[r] = __longlong_as_double(__double_as_longlong(ea) | __double_as_longlong(eb));
        }
        yh = diff3_bld2(rH, voff, hwave ? soff : (int)OOR);   // waves without a global halo row fetch nothing
    };

    // slots: plane m0-1 -> 0, m0 -> 1, m0+1 -> 2
    load_rows(P[0], rA, (kcl(m0 - 1) - pbA) * ps);
    load_rows(P[1], rA, (m0 - pbA) * ps);
    load_rows(P[2], rA, (m0 + 1 - pbA) * ps);
    load_rows(HT[1], rHt, (m0 - pbA) * ps);
    load_rows(HT[2], rHt, (m0 + 1 - pbA) * ps);
    load_halo(YH, ED, (m0 + 1 - pbA) * ps, (m0 - pbA) * ps);
    // scalar offset of iteration m: plane m+2 of rA / rHt = plane m+1 of rEA / rEB / rH = plane m-1 of rC / rD
    int so = (m0 + 2 - pbA) * ps;

    auto step = [&](auto Sc, auto Do2c, int m) {
        constexpr int S = decltype(Sc)::value;       // (m - m0) % NR
        constexpr bool DO2 = decltype(Do2c)::value;  // false in the two warm-up iterations (no L2 plane yet)
        DVec<VX>(&zmR)[RY] = P[S % NR];
        DVec<VX>(&cR)[RY] = P[(S + 1) % NR];
        DVec<VX>(&zpR)[RY] = P[(S + 2) % NR];
        DVec<VX>(&Qm)[RY] = Q[(S + NR - 1) % NR];
        DVec<VX>(&Qc)[RY] = Q[S % NR];
        DVec<VX>(&Qn)[RY] = Q[(S + 1) % NR];

        // ---- one LDS exchange for both levels: L0 rows of plane m, L1 rows of plane m-1 ----
        double* buf = xrow + (size_t)(m & 1) * ((NW + 2) * SLOT);
        {
            typedef double d2l __attribute__((ext_vector_type(2)));
            double* mine = buf + (size_t)(w + 1) * SLOT + lane * VX;
            d2l t;
            t.x = cR[0].v[0]; t.y = cR[0].v[1];           *reinterpret_cast<d2l*>(mine) = t;
            t.x = cR[RY - 1].v[0]; t.y = cR[RY - 1].v[1]; *reinterpret_cast<d2l*>(mine + TXW) = t;
            t.x = Qc[0].v[0]; t.y = Qc[0].v[1];           *reinterpret_cast<d2l*>(mine + 2 * TXW) = t;
            t.x = Qc[RY - 1].v[0]; t.y = Qc[RY - 1].v[1]; *reinterpret_cast<d2l*>(mine + 3 * TXW) = t;
            if (hwave) {   // bottom wave: slot 0 "last row"; top wave: slot NW+1 "first row"
                t.x = YH.v[0]; t.y = YH.v[1];
                *reinterpret_cast<d2l*>(buf + ((w == 0 && !top1) ? TXW : (NW + 1) * SLOT) + lane * VX) = t;
            }
        }
        diff3_lds_barrier();
        DVec<VX> yd0, yu0, yd1, yu1;
        {
            typedef double d2l __attribute__((ext_vector_type(2)));
            const double* od = buf + (size_t)w * SLOT + lane * VX;         // slot below: rows 1 (L0 last), 3 (L1 last)
            const double* ou = buf + (size_t)(w + 2) * SLOT + lane * VX;   // slot above: rows 0 (L0 first), 2 (L1 first)
            d2l t;
            t = *reinterpret_cast<const d2l*>(od + TXW);     yd0.v[0] = t.x; yd0.v[1] = t.y;
            t = *reinterpret_cast<const d2l*>(ou);           yu0.v[0] = t.x; yu0.v[1] = t.y;
            t = *reinterpret_cast<const d2l*>(od + 3 * TXW); yd1.v[0] = t.x; yd1.v[1] = t.y;
            t = *reinterpret_cast<const d2l*>(ou + 2 * TXW); yu1.v[0] = t.x; yu1.v[1] = t.y;
        }

        // ---- first step: L1 on plane m ----
        const bool zb = (m <= 0) || (m >= nz - 1);   // block-uniform: a z-boundary plane of L1 comes from B
        if (zb) {
            // rare (first / last chunk only): synchronous, so that no load of this branch is pending at the join
            const long offB = ((long)sz * m + (long)sy * j0) * 8;
            const long remB = array_bytes - offB;
            const __amdgpu_buffer_rsrc_t rB = diff3_rsrc((uintptr_t)a.B + (uintptr_t)offB, (unsigned)(remB > 0x7ffffff0L ? 0x7ffffff0L : remB));
            load_rows(Qn, rB, 0);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        } else {
            const bool own_plane = (m >= k0) && (m < k1);
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                // x-neighbours across lanes; lane 0 / lane 63 receive the edge register instead
                const double xl0 = diff3_lane_up1_edge(cR[r].v[VX - 1], ED[r]);
                const double xrL = diff3_lane_down1_edge(cR[r].v[0], ED[r]);
                double r1[VX];
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    const double xm = (v == 0) ? xl0 : cR[r].v[v == 0 ? 0 : v - 1];
                    const double xp = (v == VX - 1) ? xrL : cR[r].v[v == VX - 1 ? v : v + 1];
                    const double ym = (r == 0) ? yd0.v[v] : cR[r == 0 ? 0 : r - 1].v[v];
                    const double yp = (r == RY - 1) ? yu0.v[v] : cR[r == RY - 1 ? r : r + 1].v[v];
                    r1[v] = diff3_point<FMA>(cR[r].v[v], xm, xp, ym, yp, zmR[r].v[v], zpR[r].v[v],
                                        HT[(S + 1) % NR][r].v[v], cf, Qn[r].v[v]);
                }
                if constexpr (NORM) {
                    if (own_plane && rm[r]) {
#pragma unroll
                        for (int v = 0; v < VX; ++v) acc1[v] = __builtin_fma(r1[v], r1[v], acc1[v]);
                    }
                }
                if (xb_tile) {   // x-boundary own cells of L1 come from B (it arrived through the edge register)
                    asm volatile("" ::: "memory");   // keep this a (uniform) branch: hipcc would turn it into selects for all tiles
                    Qn[r].v[0] = bndL ? xl0 : Qn[r].v[0];
                    Qn[r].v[VX - 1] = bndR ? xrL : Qn[r].v[VX - 1];
                }
            }
            // y-boundary own rows of L1 come from B (carried in the halo-row register)
            if (bb) { asm volatile("" ::: "memory"); Qn[0] = YH; }
            if (bt) { asm volatile("" ::: "memory"); Qn[RY - 1] = YH; }
        }

        // L0 plane m-1 and the halo registers of plane m are dead: refill with L0 plane m+2 and the halos of plane m+1
        // (out of range, i.e. nothing, once the chunk ends)
        const int so1 = (m + 1 <= m1 && !DIFF3_DBG(a, 2)) ? so : (int)OOR;
        load_rows(P[S % NR], rA, (m + 2 <= nz - 1) ? so1 : (int)OOR);   // plane nz does not exist (only a z-boundary L1 plane would use it)
        load_halo(YH, ED, so1, so1 == (int)OOR ? (int)OOR : so1 - ps);

        // ---- second step: L2 on plane m-1 ----
        if constexpr (DO2) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const double fromL = diff3_lane_up1_z(Qc[r].v[VX - 1]);   // edge lanes: L2 there is never owned
                const double fromR = diff3_lane_down1_z(Qc[r].v[0]);
                double res[VX], h2[VX];
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    const double xm = (v == 0) ? fromL : Qc[r].v[v == 0 ? 0 : v - 1];
                    const double xp = (v == VX - 1) ? fromR : Qc[r].v[v == VX - 1 ? v : v + 1];
                    const double ym = (r == 0) ? yd1.v[v] : Qc[r == 0 ? 0 : r - 1].v[v];
                    const double yp = (r == RY - 1) ? yu1.v[v] : Qc[r == RY - 1 ? r : r + 1].v[v];
                    res[v] = diff3_point<FMA>(Qc[r].v[v], xm, xp, ym, yp, Qm[r].v[v], Qn[r].v[v], HT[S % NR][r].v[v], cf, h2[v]);
                }
                const int sor = (rm[r] && !DIFF3_DBG(a, 1)) ? so + r * rs : (int)OOR;   // rows the block does not own: dropped by the range check
                // lanes that own one cell of their pair (first / last owned cell of an odd-aligned range)
                double r2 = res[0], g2 = h2[0];
                if (has_split) { asm volatile("" ::: "memory"); r2 = cm[0] ? res[0] : res[1]; g2 = cm[0] ? h2[0] : h2[1]; }
                if constexpr (WRES) diff3_bst2_nt(rD, sv4, sor, res[0], res[1]);
                diff3_bst2_nt(rC, sv4, sor, h2[0], h2[1]);
                if constexpr (WRES) diff3_bst1(rD, sv2, sor, r2);
                diff3_bst1(rC, sv2, sor, g2);
                // Store-data hazard (gfx9 family, "VMEM store of more than 64 bits followed by a VALU write of its data
                // VGPRs: 1 wait state -- not needed when the store takes its offset from an SGPR"): hipcc relies on the
                // exemption, these stores do take an SGPR soffset, and on gfx950 the upper 8 bytes of their data were
                // nevertheless taken from a VALU result written 1-2 instructions later (2e-5 of the cells, odd cells
                // only, only with every CU busy).  DIFF3_STORE_NOP wait states between a row's stores and the next VALU
                // instruction; tools/hazard_soak.hip bisects the count (profiles/r2_hazard_soak.txt).  The barriers pin
                // the order 16-byte stores, 8-byte stores, wait states, next row.
#if DIFF3_STORE_NOP >= 0
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop %0" ::"n"(DIFF3_STORE_NOP));
                __builtin_amdgcn_sched_barrier(0);
#endif
                if constexpr (NORM) {
                    if (rm[r]) {
#pragma unroll
                        for (int v = 0; v < VX; ++v) acc2[v] = __builtin_fma(res[v], res[v], acc2[v]);
                    }
                }
            }
        }
        // Ht plane m-1 is dead: refill with plane m+2
        load_rows(HT[S % NR], rHt, (m + 2 <= m1 && !DIFF3_DBG(a, 2)) ? so : (int)OOR);
        so += ps;
    };

    using T = std::true_type;
    using F = std::false_type;
    // two warm-up iterations (L1 planes k0-1, k0), then the steady state in ring order 2, 0, 1
    step(std::integral_constant<int, 0>{}, F{}, m0);
    step(std::integral_constant<int, 1>{}, F{}, m0 + 1);
    int m = m0 + 2;
    for (; m + NR - 1 <= m1; m += NR) {
        step(std::integral_constant<int, 2>{}, T{}, m);
        step(std::integral_constant<int, 0>{}, T{}, m + 1);
        step(std::integral_constant<int, 1>{}, T{}, m + 2);
    }
    if (m <= m1) { step(std::integral_constant<int, 2>{}, T{}, m); ++m; }
    if (m <= m1) { step(std::integral_constant<int, 0>{}, T{}, m); ++m; }

    // lanes accumulate every cell of the owned rows / planes; cells the lane does not own are dropped here
    if constexpr (NORM) {
        tot1 += (cm[0] ? acc1[0] : 0.0) + (cm[1] ? acc1[1] : 0.0);
        tot2 += (cm[0] ? acc2[0] : 0.0) + (cm[1] ? acc2[1] : 0.0);
    }
#if DIFF3_LANE_OFF
    }
#endif
    }   // item

    if (BAL && a.ticket && tid == 0) diff3_ticket_leave(a.ticket, (int)gridDim.x);
    if constexpr (NORM) {
        const double sc2 = a.scale * a.scale;
        const double l1 = tot1 * sc2;
        const double l2 = tot2 * sc2;
        const double s1 = diff3_block_sum_waves<NW>(l1, red, tid);
        const double s2 = diff3_block_sum_waves<NW>(l2, red + NW, tid);
        if (tid == 0) { a.partials1[unit] = s1; a.partials2[unit] = s2; }
    }
}

// true if the fused two-step kernel can serve this problem
static inline bool diff3_can_fuse2(const double* Ht, const double* A, const double* B, const double* C, const double* dH,
                                   int nx, int ny, int nz)
{
    const uintptr_t al = (uintptr_t)Ht | (uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)dH;   // dH may be null
    return (nx % 2 == 0) && nx >= 128 && ny >= 16 && nz >= 3 && (al & 15) == 0;
}

// Launch on `stream`; *nparts = number of per-block partials written to each of partials1/partials2 (norm only).
// zc_opt: planes per z-chunk (0 = auto), nw_opt: waves per workgroup (0 = auto, 4 or 8), ncu: compute units of the
// device (for the chunking heuristic).
// reserve_cus > 0 (a halo exchange is in flight on the comm stream): the launch must leave that many compute units
// without a workgroup, so that RCCL's send / receive kernel runs BESIDE it instead of draining behind a grid that fills
// the device in one round -- the balanced form of the kernel (BAL) then serves the box with slots - reserve workgroups.
static inline hipError_t diff3_launch2(Diff3Args2 a, bool norm, int zc_opt, int xcd_opt, hipStream_t stream,
                                       int max_partials, int* nparts, int nw_opt = 0, int ncu = 256, int zb_lo = 0,
                                       int zb_hi = 0, int reserve_cus = 0, long* bal_info = nullptr, int* ticket = nullptr,
                                       const unsigned* reserved = nullptr)
{   // ticket (9 zeroed ints): the reserved form as ONE workgroup per device slot that take tickets (Diff3Args2::ticket)   // *bal_info (diagnostic): 0 = the plain grid was launched, else left-over units * 1000000 + slices per unit * 1000 + planes per slice
    const int wx = a.hi[0] - a.lo[0], wy = a.hi[1] - a.lo[1], wz = a.hi[2] - a.lo[2];
    const int wzb = zb_hi > zb_lo ? zb_hi - zb_lo : 0;   // second z-range (same x/y box), may be empty
    *nparts = 0;
    if (bal_info) *bal_info = 0;
    if (wx <= 0 || wy <= 0 || wz <= 0) return hipSuccess;
    if (!diff3_can_fuse2(a.Ht, a.A, a.B, a.C, a.dH, a.nx, a.ny, a.nz)) return hipErrorInvalidValue;
    // owned cells per x-tile: cut points at even cells e0 + t*sx; the tile's own cells [s, s+128) with
    // s = (ol-1)&~1 must contain ol-1 .. oh, hence sx <= 124
    const int span = a.hi[0] - (a.lo[0] & ~1);
    a.ntx = (span + 123) / 124;
    a.sx = (span + a.ntx - 1) / a.ntx;
    a.sx += a.sx & 1;
    a.xalign = (a.sx >= 32 && a.sx <= 108 && a.nx % 16 == 0 && ((((uintptr_t)a.A | (uintptr_t)a.Ht)) & 127) == 0) ? 1 : 0;
    // scalar byte offsets inside a chunk are 32-bit: (zc + 8) planes must stay below 2 GiB
    const long psb = (long)a.nx * a.ny * 8;
    if (psb * 12 >= (1L << 31)) return hipErrorInvalidValue;
    const int zc_max = (int)((1L << 31) / psb) - 8;
    // chunking: workgroups run in rounds of `slots` (what the device holds at once); a chunk of zc planes costs
    // zc + 6 plane-iterations (2 warm-up iterations, prologue / epilogue).  Pick the chunk count with the least
    // rounds x cost; ties go to more workgroups.
    auto plan = [&](int nw, int* zc_out) -> long {
        const int syb_ = nw * 4 - 2;
        const long tiles = (long)a.ntx * ((wy + syb_ - 1) / syb_);
        const long slots = (long)(ncu > 0 ? ncu : 256) * (nw == 8 ? 1 : (nw == 1 ? 6 : 2));   // nw = 1: 24.5 KB of LDS per workgroup
        long best = -1;
        int zb = 0;
        for (int ntz = 1; ntz <= wz; ++ntz) {
            const int z = (wz + ntz - 1) / ntz;
            if (z > zc_max) continue;
            if (z < 4 && ntz > 1) break;
            const long nb = tiles * ((wz + z - 1) / z + (wzb + z - 1) / z);
            const long cost = ((nb + slots - 1) / slots) * (z + 6);
            if (best < 0 || cost <= best) { best = cost; zb = z; }
        }
        if (zb <= 0) { zb = wz < zc_max ? wz : zc_max; best = zb + 6; }
        *zc_out = zb;
        return best;
    };
    // 8 waves per workgroup (32-row blocks, 30 owned: less redundant level-1 work and fewer re-read rows; one
    // workgroup per CU) when the grid is tall enough, else 4 (16-row blocks, two workgroups per CU).  On small grids the
    // 16-row blocks give twice as many (x, y) tiles and hence longer z-chunks for the same number of workgroups: they are
    // taken when the plan with them is strictly cheaper (128^3: 11 against 12 plane-iterations, measured 15.6 against 17.5 us
    // per iteration; 192^3: 17 = 17, measured equal; 256^3: 38 against 35, measured 59 against 56 us)
    int zc = zc_opt, zc8 = 0, zc4 = 0;
    const bool can8 = a.ny >= 32 && wy >= 24;
    // one-row shells next to a y-neighbour (rows 1 / ny-2, possibly both rows of a 2-row box at the boundary): one wave per
    // workgroup; needs the tile of 4 rows to touch the y-boundary and its far row to feed nothing that is owned
    const bool can1 = wzb == 0 && a.ny >= 8 && ((a.lo[1] == 1 && wy <= 1) || (a.hi[1] == a.ny - 1 && wy <= 1));
    if (nw_opt == 4 || nw_opt == 8) a.nw = (nw_opt == 8 && a.ny < 32) ? 4 : nw_opt;
    else if (nw_opt == 1 ? can1 : (can1 && nw_opt == 0)) a.nw = 1;
    else if (!can8) a.nw = 4;
    else a.nw = (zc_opt <= 0 && plan(4, &zc4) < plan(8, &zc8)) ? 4 : 8;
    const int syb = a.nw * 4 - 2;
    a.nby = (wy + syb - 1) / syb;
    const long tiles_xy = (long)a.ntx * a.nby;
    if (zc <= 0) plan(a.nw, &zc);
    if (zc > wz) zc = wz;
    if (zc > zc_max) zc = zc_max;
    a.zc = zc;
    a.ntz_a = (wz + zc - 1) / zc;
    a.zb_lo = zb_lo;
    a.zb_hi = zb_lo + wzb;
    a.ntz = a.ntz_a + (wzb + zc - 1) / zc;
    const long nblk = tiles_xy * a.ntz;
    if (nblk > 0x7fffffffL || (norm && nblk > max_partials)) return hipErrorInvalidValue;
#ifdef FPR_TUNE
    a.dbg = xcd_opt >> 4;
#endif
    xcd_opt &= 15;
    // block -> XCD mapping: 0 = auto, 1 = contiguous range of tiles per XCD, 2 = z-chunk ownership, 3 = hardware order.
    // With at most ~two rounds of workgroups the contiguous mapping puts y-neighbours (which share 4 of 34 rows) on
    // one L2 while they march in lockstep: measured -5..7 % time and -14 % read traffic at 512^3; with many rounds it
    // was slower than the hardware's round-robin.
    {
        const long slots = (long)(ncu > 0 ? ncu : 256) * (a.nw == 8 ? 1 : (a.nw == 1 ? 6 : 2));
        if (xcd_opt == 0) xcd_opt = (nblk >= 64 && nblk <= 2 * slots) ? 1 : 3;
    }
    a.xcd_remap = (xcd_opt == 1 && nblk >= 64) ? 1 : ((xcd_opt == 2 && a.ntz % 8 == 0) ? 2 : 0);
    const bool wres = a.dH != nullptr;
    a.bal_r = a.bal_sp = a.bal_q = 0;
    a.ticket = nullptr; a.bal_g = 0;
    a.reserved = nullptr;
    if (a.nw == 1) {
        if (wres) {
            if (norm) k_diff3_march2<true, 1, true><<<(int)nblk, 64, 0, stream>>>(a);
            else k_diff3_march2<false, 1, true><<<(int)nblk, 64, 0, stream>>>(a);
        } else {
            if (norm) k_diff3_march2<true, 1, false><<<(int)nblk, 64, 0, stream>>>(a);
            else k_diff3_march2<false, 1, false><<<(int)nblk, 64, 0, stream>>>(a);
        }
        *nparts = (int)nblk;
        return hipGetLastError();
    }
    if (reserve_cus != 0 && wzb == 0) {   // < 0 (tests, option diff3_bal_g): the reserved form on exactly -reserve_cus workgroups
        // The plain grid runs in rounds of `slots` workgroups.  A grid of several rounds frees units all the time and one
        // that leaves `reserve_cus` units idle anyway needs no change; a grid that fills the device in ONE round is cut to
        // G = slots - reserve workgroups: each serves its own (tile, chunk) unit, the r = nblk - G left-over units are
        // sliced thinly over all of them (sp slices of q planes per unit; a slice costs q + 6 plane-iterations).
        const int ncu_ = ncu > 0 ? ncu : 256, per_cu = a.nw == 8 ? 1 : 2;
        const long slots = (long)ncu_ * per_cu;
        const long G = reserve_cus < 0 ? -(long)reserve_cus : slots - (long)(reserve_cus < ncu_ / 2 ? reserve_cus : ncu_ / 2) * per_cu;
        const long r = nblk - G;
        if (r > 0 && r <= G && (reserve_cus < 0 || nblk <= slots) && (!norm || G <= max_partials)) {
            a.bal_r = (int)r;
            a.bal_sp = (int)(G / r);
            a.bal_q = (zc + a.bal_sp - 1) / a.bal_sp;
            if (a.xcd_remap == 2) a.xcd_remap = 0;
            a.bal_g = (int)G;
            long grid = G;
            if (ticket && G <= slots) {
                a.ticket = ticket;
                a.reserved = reserved;
                grid = slots;
            }
            if (a.nw == 8) {
                if (wres) {
                    if (norm) k_diff3_march2<true, 8, true, true><<<(int)grid, 512, 0, stream>>>(a);
                    else k_diff3_march2<false, 8, true, true><<<(int)grid, 512, 0, stream>>>(a);
                } else {
                    if (norm) k_diff3_march2<true, 8, false, true><<<(int)grid, 512, 0, stream>>>(a);
                    else k_diff3_march2<false, 8, false, true><<<(int)grid, 512, 0, stream>>>(a);
                }
            } else {
                if (wres) {
                    if (norm) k_diff3_march2<true, 4, true, true><<<(int)grid, 256, 0, stream>>>(a);
                    else k_diff3_march2<false, 4, true, true><<<(int)grid, 256, 0, stream>>>(a);
                } else {
                    if (norm) k_diff3_march2<true, 4, false, true><<<(int)grid, 256, 0, stream>>>(a);
                    else k_diff3_march2<false, 4, false, true><<<(int)grid, 256, 0, stream>>>(a);
                }
            }
            *nparts = (int)G;
            if (bal_info) *bal_info = r * 1000000 + (long)a.bal_sp * 1000 + a.bal_q;
            return hipGetLastError();
        }
    }
    if (a.fma) {      // opt-in contracted arithmetic: the plain grid only (the reserved / one-wave forms above stay exact)
        if (a.nw == 8) {
            if (wres) {
                if (norm) k_diff3_march2<true, 8, true, false, true><<<(int)nblk, 512, 0, stream>>>(a);
                else k_diff3_march2<false, 8, true, false, true><<<(int)nblk, 512, 0, stream>>>(a);
            } else {
                if (norm) k_diff3_march2<true, 8, false, false, true><<<(int)nblk, 512, 0, stream>>>(a);
                else k_diff3_march2<false, 8, false, false, true><<<(int)nblk, 512, 0, stream>>>(a);
            }
        } else {
            if (wres) {
                if (norm) k_diff3_march2<true, 4, true, false, true><<<(int)nblk, 256, 0, stream>>>(a);
                else k_diff3_march2<false, 4, true, false, true><<<(int)nblk, 256, 0, stream>>>(a);
            } else {
                if (norm) k_diff3_march2<true, 4, false, false, true><<<(int)nblk, 256, 0, stream>>>(a);
                else k_diff3_march2<false, 4, false, false, true><<<(int)nblk, 256, 0, stream>>>(a);
            }
        }
        *nparts = (int)nblk;
        return hipGetLastError();
    }
    if (a.nw == 8) {
        if (wres) {
            if (norm) k_diff3_march2<true, 8, true><<<(int)nblk, 512, 0, stream>>>(a);
            else k_diff3_march2<false, 8, true><<<(int)nblk, 512, 0, stream>>>(a);
        } else {
            if (norm) k_diff3_march2<true, 8, false><<<(int)nblk, 512, 0, stream>>>(a);
            else k_diff3_march2<false, 8, false><<<(int)nblk, 512, 0, stream>>>(a);
        }
    } else {
        if (wres) {
            if (norm) k_diff3_march2<true, 4, true><<<(int)nblk, 256, 0, stream>>>(a);
            else k_diff3_march2<false, 4, true><<<(int)nblk, 256, 0, stream>>>(a);
        } else {
            if (norm) k_diff3_march2<true, 4, false><<<(int)nblk, 256, 0, stream>>>(a);
            else k_diff3_march2<false, 4, false><<<(int)nblk, 256, 0, stream>>>(a);
        }
    }
    *nparts = (int)nblk;
    return hipGetLastError();
}
