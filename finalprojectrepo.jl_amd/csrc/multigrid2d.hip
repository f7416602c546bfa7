// multigrid2d.hip -- Part 2 of libfpr_hip.so: 2D geometric-multigrid V-cycle for (lap - c) u = f.
// Reference: scripts-part2/multigrid.jl, krylov.jl, part2_utils.jl (file:line cited per function).
//
// Data layout: Float64, column-major nx*ny (ix fastest).  All kernels use 64x4 thread blocks: a wave
// covers 64 consecutive x of one row (512 B coalesced per instruction; (2^k+1)-wide rows are only
// 8-byte aligned, so 16-byte accesses are not generally possible).
//
// Inside fpr_vcycle2d the reference's passes are fused without changing a single rounding:
//   * Jacobi sweep      : residual + update in one pass, ping-pong u <-> tmp   (multigrid.jl:245-258)
//   * residual+restrict : the residual is evaluated only at the injected points  (:128-129, :330-358)
//   * prolong+correct   : deterministic gather in the reference's accumulation order, subtracted
//                         from u in the same pass                              (:136-139, :403-472)
// The materialising entry points (fpr_residual2d, fpr_jacobi2d, fpr_restrict2d, fpr_prolongate2d)
// keep the reference's buffers observable for API parity.
#include <type_traits>

#include "fpr_internal.hpp"

#define BX 64
#define BY 4
static inline dim3 grid2(int nx, int ny) { return dim3((nx + BX - 1) / BX, (ny + BY - 1) / BY, 1); }
static const dim3 blk2(BX, BY, 1);

// residual at interior point (i,j): multigrid.jl:178-185
__device__ __forceinline__ double res_at(const double* __restrict__ u, const double* __restrict__ f, size_t id, int nx,
                                         double C, double _h2)
{
    return ((((u[id + 1] + u[id - 1]) + u[id + nx]) + u[id - nx]) - C * u[id]) * _h2 - f[id];
}

// ---- B1 -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_residual2d(const double* __restrict__ u, const double* __restrict__ f,
                                                     double* __restrict__ res, int nx, int ny, double C, double _h2)
{
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    if (i < 1 || j < 1 || i >= nx - 1 || j >= ny - 1) return;
    const size_t id = (size_t)i + (size_t)nx * j;
    res[id] = res_at(u, f, id, nx, C, _h2);
}

// u .+= fac .* res over the whole array (multigrid.jl:255)
__global__ __launch_bounds__(256) void k_axpy_inplace(double* __restrict__ u, const double* __restrict__ res, double fac,
                                                       size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) u[i] = u[i] + fac * res[i];
}

// ---- fused Jacobi sweep: uout = uin + fac*residual(uin) (interior), boundary copied --------------
// STATE: coarse-solver form, skipped entirely once state->done is set.
template <bool NORM, bool STATE>
__global__ __launch_bounds__(256) void k_sweep2d(const double* __restrict__ uin, const double* __restrict__ f,
                                                  double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                  double fac, double* __restrict__ partials,
                                                  const FprSolveState* __restrict__ state)
{
    __shared__ double red[16];
    if constexpr (STATE) {
        if (state->done) return;
    }
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    double acc = 0.0;
    if (i < nx && j < ny) {
        const size_t id = (size_t)i + (size_t)nx * j;
        const double uc = uin[id];
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const double r = ((((uin[id + 1] + uin[id - 1]) + uin[id + nx]) + uin[id - nx]) - C * uc) * _h2 - f[id];
            uout[id] = uc + fac * r;
            if constexpr (NORM) acc = r * r;
        } else {
            uout[id] = uc;
        }
    }
    if constexpr (NORM) {
        const double s = fpr_block_sum<256>(acc, red);
        if (threadIdx.x == 0 && threadIdx.y == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = s;
    }
}

// ---- S fused Jacobi sweeps per pass (temporal blocking in LDS) -------------------------------------
// A workgroup loads its TX x TY output tile plus a halo of S cells of u and f into LDS, applies up to S
// damped-Jacobi sweeps there (the valid region shrinks by one ring per sweep, except along domain
// boundaries, whose values are fixed) and writes the tile once: 2 sweeps cost one read of u and f and
// one write of u instead of two.  Arithmetic per point is that of k_sweep2d, so results are
// bit-identical.  NORM: 0 none, 1 sum(res^2) of the LAST executed sweep, 2 of EVERY sweep (coarse-solve
// exit test); partials[s * nblocks + block] for sweep s, counted on the workgroup's own tile only.
// STATE: skipped once state->done is set; nsw (<= S) = sweeps to execute.
template <int S, int TX, int TY, int NORM, bool STATE>
__global__ __launch_bounds__(256) void k_sweep2d_multi(const double* __restrict__ uin, const double* __restrict__ f,
                                                        double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                        double fac, int nsw, double* __restrict__ partials,
                                                        const FprSolveState* __restrict__ state)
{
    constexpr int P = TX + 2 * S;       // LDS pitch
    constexpr int RH = TY + 2 * S;      // LDS rows
    __shared__ double bufA[P * RH];
    __shared__ double bufB[P * RH];
    __shared__ double bufF[P * RH];
    __shared__ double red[16];
    if constexpr (STATE) {
        if (state->done) return;
    }
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;           // own tile origin
    const int rx0 = x0 - S < 0 ? 0 : x0 - S, ry0 = y0 - S < 0 ? 0 : y0 - S;
    const int rx1 = x0 + TX + S > nx ? nx : x0 + TX + S, ry1 = y0 + TY + S > ny ? ny : y0 + TY + S;
    const int RWr = rx1 - rx0, RHr = ry1 - ry0;                      // loaded region
    const bool exl = rx0 > 0, exh = rx1 < nx, eyl = ry0 > 0, eyh = ry1 < ny;  // "open" (shrinking) edges
    for (int idx = tid; idx < P * RH; idx += 256) {
        const int r = idx / P, cc = idx - r * P;
        if (r < RHr && cc < RWr) {
            const size_t g = (size_t)(rx0 + cc) + (size_t)nx * (size_t)(ry0 + r);
            bufA[idx] = uin[g];
            bufF[idx] = f[g];
        }
    }
    __syncthreads();
    double* cur = bufA;
    double* nxt = bufB;
    const int blk = blockIdx.x + gridDim.x * blockIdx.y;
    const int nblk = gridDim.x * gridDim.y;
    for (int sIt = 0; sIt < nsw; ++sIt) {
        const int clo = exl ? sIt + 1 : 0, chi = RWr - (exh ? sIt + 1 : 0);
        const int rlo = eyl ? sIt + 1 : 0, rhi = RHr - (eyh ? sIt + 1 : 0);
        double acc = 0.0;
        for (int idx = tid; idx < P * RH; idx += 256) {
            const int r = idx / P, cc = idx - r * P;
            if (r >= rlo && r < rhi && cc >= clo && cc < chi) {
                const int gi = rx0 + cc, gj = ry0 + r;
                const double uc = cur[idx];
                if (gi >= 1 && gj >= 1 && gi < nx - 1 && gj < ny - 1) {
                    const double rr = ((((cur[idx + 1] + cur[idx - 1]) + cur[idx + P]) + cur[idx - P]) - C * uc) * _h2 - bufF[idx];
                    nxt[idx] = uc + fac * rr;
                    if constexpr (NORM != 0) {
                        if (gi >= x0 && gi < x0 + TX && gj >= y0 && gj < y0 + TY) acc += rr * rr;
                    }
                } else {
                    nxt[idx] = uc;
                }
            }
        }
        if constexpr (NORM == 2) {
            const double sblk = fpr_block_sum<256>(acc, red);  // contains a barrier
            if (tid == 0) partials[(size_t)sIt * nblk + blk] = sblk;
        } else if constexpr (NORM == 1) {
            if (sIt == nsw - 1) {
                const double sblk = fpr_block_sum<256>(acc, red);
                if (tid == 0) partials[blk] = sblk;
            }
        }
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
    }
    // write the own tile
    for (int idx = tid; idx < TX * TY; idx += 256) {
        const int r = idx / TX, cc = idx - r * TX;
        const int gi = x0 + cc, gj = y0 + r;
        if (gi < nx && gj < ny) uout[(size_t)gi + (size_t)nx * gj] = cur[(gi - rx0) + P * (gj - ry0)];
    }
}

// Branch-free form of the bilinear gather (same value, same accumulation order as prolong_at below):
// v = (((0 + w00*c00) + w10*c10) + w01*c01) + w11*c11 with weights 1 | .5,.5 | .25 x4 by parity, a term
// being dropped (exact +0) when its coarse point is not an interior source.
__device__ __forceinline__ double prolong_bf(const double* __restrict__ cc, int i, int j, int nx, int ny, int nxc, int nyc)
{
    const bool in = i >= 1 && j >= 1 && i <= nx - 2 && j <= ny - 2;
    const int io = i & 1, jo = j & 1;
    const int icl = i >> 1, jcl = j >> 1;
    const int ich = (icl + 1 < nxc) ? icl + 1 : nxc - 1, jch = (jcl + 1 < nyc) ? jcl + 1 : nyc - 1;
    const double w = (io | jo) ? ((io & jo) ? 0.25 : 0.5) : 1.0;
    const bool sx0 = icl >= 1 && icl <= nxc - 2, sx1 = io && (icl + 1 <= nxc - 2);
    const bool sy0 = jcl >= 1 && jcl <= nyc - 2, sy1 = jo && (jcl + 1 <= nyc - 2);
    const double c00 = cc[(size_t)icl + (size_t)nxc * jcl], c10 = cc[(size_t)ich + (size_t)nxc * jcl];
    const double c01 = cc[(size_t)icl + (size_t)nxc * jch], c11 = cc[(size_t)ich + (size_t)nxc * jch];
    double v = 0.0;
    v = v + ((in && sx0 && sy0) ? w * c00 : 0.0);
    v = v + ((in && sx1 && sy0) ? w * c10 : 0.0);
    v = v + ((in && sx0 && sy1) ? w * c01 : 0.0);
    v = v + ((in && sx1 && sy1) ? w * c11 : 0.0);
    return v;
}

// ---- buffer addressing helpers (descriptor base + per-lane byte offset + scalar byte offset) ---------------------
// An offset of FPR_OOR is beyond every descriptor's num_records: the hardware drops the access (loads return 0).  The
// marching kernels use it instead of branches around loads / stores: with conditional memory instructions hipcc cannot
// count the operations younger than a prefetch and waits for more than it has to (finding 1).
constexpr unsigned FPR_OOR = 0x7fffffffu;
typedef unsigned fpr_u2v __attribute__((ext_vector_type(2)));
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
template <bool NORM, bool PROLONG, bool RESTRICT>
__global__ __launch_bounds__(256) void k_smooth2_march(const double* __restrict__ uin, const double* __restrict__ f,
                                                        double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                        double fac, int rows_per_chunk, int nstrips,
                                                        double* __restrict__ partials, const double* __restrict__ corr_c,
                                                        int apply_BCs, double* __restrict__ res_c_out,
                                                        double* __restrict__ corr_c_out, const int* __restrict__ skip)
{
    if (skip && *skip) return;   // a cycle enqueued ahead of the exit test that ended the loop (FprCycleCtl)
    __shared__ double red[16];
    constexpr int HX = RESTRICT ? 3 : 2;                     // feeder lanes on each side of a strip
    constexpr int SW = 64 - 2 * HX;                          // columns owned by a strip
    apply_BCs &= 255;                                        // (bit 8, non-temporal stores, is ignored by this kernel)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + w;
    const bool active = strip < nstrips;
    const int gi = strip * SW - HX + lane;                   // global column of this lane
    const bool col_ok = active && gi >= 0 && gi < nx;
    const int gic = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);  // clamped for loads
    const bool col_bnd = gi <= 0 || gi >= nx - 1;            // domain boundary column (or outside)
    const bool owner = col_ok && lane >= HX && lane < 64 - HX;
    const int y0 = blockIdx.y * rows_per_chunk;
    const int y1 = (y0 + rows_per_chunk < ny) ? y0 + rows_per_chunk : ny;  // output rows [y0, y1)
    const int rs = y0 - HX < 0 ? 0 : y0 - HX;
    double acc = 0.0;
    if (active) {
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        int gis = gic;  // Neumann rows of the prolongated correction (part2_utils.jl:35-39)
        if (PROLONG && apply_BCs) gis = (gic == 0) ? 1 : (gic == nx - 1 ? nx - 2 : gic);
        // PROLONG: the two coarse columns this lane interpolates from, cached for coarse rows pj and pj+1
        // (rows are visited in increasing order, so a coarse row is loaded once per two fine rows)
        const int p_io = gis & 1, p_icl = gis >> 1, p_ich = (p_icl + 1 < nxc) ? p_icl + 1 : nxc - 1;
        const bool p_sx0 = p_icl >= 1 && p_icl <= nxc - 2, p_sx1 = p_io && (p_icl + 1 <= nxc - 2);
        const bool p_inx = gis >= 1 && gis <= nx - 2;
        int pj = -2;
        double pc00 = 0.0, pc10 = 0.0, pc01 = 0.0, pc11 = 0.0;
        // rows are addressed relative to the first row of the chunk: (rows_per_chunk + 8) * nx * 8 < 2^31
        const int rowB = nx * 8;
        const __amdgpu_buffer_rsrc_t rUin = fpr_rsrc(uin + (size_t)nx * rs), rF = fpr_rsrc(f + (size_t)nx * rs);
        const __amdgpu_buffer_rsrc_t rUout = fpr_rsrc(uout + (size_t)nx * rs);
        const unsigned vld = (unsigned)gic * 8u;
        const unsigned vst = owner ? (unsigned)gi * 8u : FPR_OOR;   // lanes that own nothing store out of range
        auto ldu = [&](int r) {
            const int rc = r > ny - 1 ? ny - 1 : r;
            const double v = fpr_bld(rUin, vld, (rc - rs) * rowB);
            if constexpr (PROLONG) {
                const int jo = rc & 1, jcl = rc >> 1;
                if (jcl != pj) {
                    const int jch = (jcl + 1 < nyc) ? jcl + 1 : nyc - 1;
                    if (jcl == pj + 1) { pc00 = pc01; pc10 = pc11; }
                    else { pc00 = corr_c[(size_t)p_icl + (size_t)nxc * jcl]; pc10 = corr_c[(size_t)p_ich + (size_t)nxc * jcl]; }
                    pc01 = corr_c[(size_t)p_icl + (size_t)nxc * jch];
                    pc11 = corr_c[(size_t)p_ich + (size_t)nxc * jch];
                    pj = jcl;
                }
                // same value and accumulation order as prolong_bf
                const bool in = p_inx && rc >= 1 && rc <= ny - 2;
                const double wgt = (p_io | jo) ? ((p_io & jo) ? 0.25 : 0.5) : 1.0;
                const bool sy0 = jcl >= 1 && jcl <= nyc - 2, sy1 = jo && (jcl + 1 <= nyc - 2);
                double pv = 0.0;
                pv = pv + ((in && p_sx0 && sy0) ? wgt * pc00 : 0.0);
                pv = pv + ((in && p_sx1 && sy0) ? wgt * pc10 : 0.0);
                pv = pv + ((in && p_sx0 && sy1) ? wgt * pc01 : 0.0);
                pv = pv + ((in && p_sx1 && sy1) ? wgt * pc11 : 0.0);
                return v - pv;
            } else {
                return v;
            }
        };
        auto ldf = [&](int r) { const int rc = r > ny - 1 ? ny - 1 : r; return fpr_bld(rF, vld, (rc - rs) * rowB); };
        double a0 = 0.0, a1 = 0.0, a2 = ldu(rs);       // u   rows r-2, r-1, r
        double b0 = 0.0, b1 = 0.0, b2 = 0.0;           // u1  rows r-3, r-2, r-1
        double f0 = 0.0, f1 = 0.0, f2 = ldf(rs);       // f   rows r-2, r-1, r
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, fm = 0.0; // RESTRICT: u2 rows r-4, r-3, r-2 and f row r-3
        // software pipeline: rows r+1 .. r+PF are in flight (PF loads of u and of f per lane)
        constexpr int PF = 4;
        double pu[PF], pfv[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) { pu[q] = ldu(rs + 1 + q); pfv[q] = ldf(rs + 1 + q); }
        const int nxc_r = 1 + (nx - 1) / 2, nyc_r = 1 + (ny - 1) / 2;
        // RESTRICT: coarse arrays (whole-array descriptors: nxc * nyc * 8 < 2^31), even owned columns only
        const __amdgpu_buffer_rsrc_t rResC = fpr_rsrc(res_c_out), rCorC = fpr_rsrc(corr_c_out);
        const unsigned vstc = (RESTRICT && owner && !(gi & 1)) ? (unsigned)(gi >> 1) * 8u : FPR_OOR;
        // RESTRICT with apply_BCs: the coarse right-hand side gets its Neumann columns here (part2_utils.jl:35-39 as applied
        // at multigrid.jl:355-357: column 0 = column 1, column nxc-1 = column nxc-2) -- the lanes of coarse columns 1 and
        // nxc-2 store their value a second time, the lanes of columns 0 and nxc-1 do not store theirs
        const bool nbc = RESTRICT && apply_BCs != 0;
        const unsigned vstr = (nbc && (gi == 0 || gi == nx - 1)) ? FPR_OOR : vstc;
        const unsigned vstn = (nbc && owner && (gi == 2 || gi == nx - 3)) ? (gi == 2 ? 0u : (unsigned)(nxc_r - 1) * 8u) : FPR_OOR;
        // The ring slot is a compile-time constant (the row loop is unrolled by PF): a slot is consumed and
        // refilled in place, so no register of an in-flight load is ever copied (a copy would make hipcc wait
        // for that load) and PF rows stay in flight per lane.
        auto step = [&](auto Qc, int r) {
            constexpr int Q = decltype(Qc)::value;
            // An explicit register copy of the (completed) row ends the live range of the slot, so the refill below
            // can target the slot's own registers and nothing in flight has to be copied at the loop's back edge.
            double an, fn;
            asm volatile("v_mov_b64 %0, %1" : "=v"(an) : "v"(pu[Q]));
            asm volatile("v_mov_b64 %0, %1" : "=v"(fn) : "v"(pfv[Q]));
            pu[Q] = ldu(r + 1 + PF);                           // issue the loads of row r+1+PF
            pfv[Q] = ldf(r + 1 + PF);
            // ---- sweep 1 at row r-1 (needs u rows r-2, r-1, r) ----
            const int j1 = r - 1;
            double u1;
            {
                const double L = fpr_lane_up1z(a1), R = fpr_lane_down1z(a1);
                const double rr = ((((R + L) + a2) + a0) - C * a1) * _h2 - f1;
                const bool bnd = col_bnd || j1 <= 0 || j1 >= ny - 1;
                u1 = bnd ? a1 : a1 + fac * rr;
            }
            b0 = b1; b1 = b2; b2 = u1;                 // u1 rows r-3, r-2, r-1
            // ---- sweep 2 at row r-2 (needs u1 rows r-3, r-2, r-1) ----
            const int j2 = r - 2;
            {
                const double L = fpr_lane_up1z(b1), R = fpr_lane_down1z(b1);
                const double rr = ((((R + L) + b2) + b0) - C * b1) * _h2 - f0;
                const bool bnd = col_bnd || j2 <= 0 || j2 >= ny - 1;
                const double u2 = bnd ? b1 : b1 + fac * rr;
                const bool row_own = j2 >= y0 && j2 < y1;        // uniform
                fpr_bst(rUout, vst, row_own ? (j2 - rs) * rowB : (int)FPR_OOR, u2);   // unconditional (see FPR_OOR)
                if constexpr (NORM) {
                    if (owner && row_own && !bnd) acc += rr * rr;
                }
                if constexpr (RESTRICT) {
                    c0 = c1; c1 = c2; c2 = u2;         // u2 rows r-4, r-3, r-2
                }
            }
            if constexpr (RESTRICT) {
                // ---- residual of u2 at row r-3, injected at even (row, column) ----
                const int j3 = r - 3;
                const double L = fpr_lane_up1z(c1), R = fpr_lane_down1z(c1);
                const double rr = ((((R + L) + c2) + c0) - C * c1) * _h2 - fm;
                {
                    const int ic = gi >> 1, jc = j3 >> 1;
                    const bool cint = ic >= 1 && ic <= nxc_r - 2 && jc >= 1 && jc <= nyc_r - 2;
                    const bool row_inj = j3 >= y0 && j3 < y1 && !(j3 & 1);   // uniform
                    const int sc = row_inj ? jc * (nxc_r * 8) : (int)FPR_OOR;
                    fpr_bst(rResC, vstr, sc, cint ? rr : 0.0);
                    fpr_bst(rResC, vstn, sc, cint ? rr : 0.0);
                    fpr_bst(rCorC, vstc, sc, 0.0);
                }
                fm = f0;                               // becomes f row (r+1)-3
            }
            a0 = a1; a1 = a2; a2 = an;
            f0 = f1; f1 = f2; f2 = fn;
        };
        const int rend = y1 + (RESTRICT ? 2 : 1);
        int r = rs;
        static_assert(PF == 4, "the unrolled row loop below is written for PF = 4 (PF = 8 measured 12 % slower)");
        for (; r + PF - 1 <= rend; r += PF) {
            step(std::integral_constant<int, 0>{}, r);
            step(std::integral_constant<int, 1>{}, r + 1);
            step(std::integral_constant<int, 2>{}, r + 2);
            step(std::integral_constant<int, 3>{}, r + 3);
        }
        if (r <= rend) { step(std::integral_constant<int, 0>{}, r); ++r; }
        if (r <= rend) { step(std::integral_constant<int, 1>{}, r); ++r; }
        if (r <= rend) { step(std::integral_constant<int, 2>{}, r); ++r; }
    }
    if constexpr (NORM) {
        const double sblk = fpr_block_sum<256>(acc, red);
        if (threadIdx.x == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
    }
}

// ---- the seam between two V-cycles on the finest level: FOUR sweeps in one pass ------------------------------
// MGsolve_2DPoisson! (multigrid.jl:57-71) runs V-cycle after V-cycle on the same arrays: the post-smoothing pair of
// cycle k (:142-143, on the field corrected by the prolongated coarse solution, :136-139) is followed -- if the exit
// test :70 does not end the loop -- by the pre-smoothing pair of cycle k+1 (:124-125) and its residual + injection
// (:128-132).  k_seam_march does all of that in ONE pass over the finest grid: the same register-rolling march as
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
template <bool BCS>
__global__ __launch_bounds__(256) void k_seam_march(const double* __restrict__ uin, const double* __restrict__ f,
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
    const int y0 = blockIdx.y * rows_per_chunk;
    const int y1 = (y0 + rows_per_chunk < ny) ? y0 + rows_per_chunk : ny;  // output rows [y0, y1)
    const int rs = y0 - HX < 0 ? 0 : y0 - HX;
    double acc = 0.0;
    if (active) {
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        // the two coarse columns this lane interpolates from, cached for coarse rows pj and pj+1 (see k_smooth2_march)
        int gis = gic;  // Neumann rows of the prolongated correction (part2_utils.jl:35-39)
        if (BCS) gis = (gic == 0) ? 1 : (gic == nx - 1 ? nx - 2 : gic);
        const int p_io = gis & 1, p_icl = gis >> 1, p_ich = (p_icl + 1 < nxc) ? p_icl + 1 : nxc - 1;
        const bool p_sx0 = p_icl >= 1 && p_icl <= nxc - 2, p_sx1 = p_io && (p_icl + 1 <= nxc - 2);
        const bool p_inx = gis >= 1 && gis <= nx - 2;
        int pj = -2;
        double pc00 = 0.0, pc10 = 0.0, pc01 = 0.0, pc11 = 0.0;
        const int rowB = nx * 8;
        const __amdgpu_buffer_rsrc_t rUin = fpr_rsrc(uin + (size_t)nx * rs), rF = fpr_rsrc(f + (size_t)nx * rs);
        const __amdgpu_buffer_rsrc_t rUout = fpr_rsrc(uout + (size_t)nx * rs);
        const unsigned vld = (unsigned)gic * 8u;
        const unsigned vst = owner ? (unsigned)gi * 8u : FPR_OOR;   // lanes that own nothing store out of range
        auto ldu = [&](int r) {
            const int rc = r > ny - 1 ? ny - 1 : r;
            const double v = fpr_bld(rUin, vld, (rc - rs) * rowB);
            const int jo = rc & 1, jcl = rc >> 1;
            if (jcl != pj) {
                const int jch = (jcl + 1 < nyc) ? jcl + 1 : nyc - 1;
                if (jcl == pj + 1) { pc00 = pc01; pc10 = pc11; }
                else { pc00 = corr_c[(size_t)p_icl + (size_t)nxc * jcl]; pc10 = corr_c[(size_t)p_ich + (size_t)nxc * jcl]; }
                pc01 = corr_c[(size_t)p_icl + (size_t)nxc * jch];
                pc11 = corr_c[(size_t)p_ich + (size_t)nxc * jch];
                pj = jcl;
            }
            // same value and accumulation order as prolong_bf
            const bool in = p_inx && rc >= 1 && rc <= ny - 2;
            const double wgt = (p_io | jo) ? ((p_io & jo) ? 0.25 : 0.5) : 1.0;
            const bool sy0 = jcl >= 1 && jcl <= nyc - 2, sy1 = jo && (jcl + 1 <= nyc - 2);
            double pv = 0.0;
            pv = pv + ((in && p_sx0 && sy0) ? wgt * pc00 : 0.0);
            pv = pv + ((in && p_sx1 && sy0) ? wgt * pc10 : 0.0);
            pv = pv + ((in && p_sx0 && sy1) ? wgt * pc01 : 0.0);
            pv = pv + ((in && p_sx1 && sy1) ? wgt * pc11 : 0.0);
            return v - pv;
        };
        auto ldf = [&](int r) { const int rc = r > ny - 1 ? ny - 1 : r; return fpr_bld(rF, vld, (rc - rs) * rowB); };
        // 3-row windows with COMPILE-TIME slots: row j of every field lives in slot (j - rs) mod 3 (f: mod 6), and the row
        // loop is unrolled by 12 = lcm(3, 4, 6) so that the slot of every operand is a constant: no register moves to
        // shift fifteen window rows per step (the 2-sweep kernel shifts its windows; here that would be a fifth of the VALU work).
        // w[0] = corrected input, w[K] = field after sweep K.
        double w[5][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        double fw[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        w[0][0] = ldu(rs);
        fw[0] = ldf(rs);
        constexpr int PF = 4;
        double pu[PF], pfv[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) { pu[q] = ldu(rs + 1 + q); pfv[q] = ldf(rs + 1 + q); }
        const __amdgpu_buffer_rsrc_t rResC = fpr_rsrc(res_c_out), rCorC = fpr_rsrc(corr_c_out);
        const unsigned vstc = (owner && !(gi & 1)) ? (unsigned)(gi >> 1) * 8u : FPR_OOR;
        // BCS: Neumann columns of the coarse right-hand side (:355-357), see k_smooth2_march
        const unsigned vstr = (BCS && (gi == 0 || gi == nx - 1)) ? FPR_OOR : vstc;
        const unsigned vstn = (BCS && owner && (gi == 2 || gi == nx - 3)) ? (gi == 2 ? 0u : (unsigned)(nxc - 1) * 8u) : FPR_OOR;
        // one Jacobi sweep at row j of a field whose rows j-1, j, j+1 are (lo, mid, hi); rr = residual used by the update
        auto sweep = [&](double lo, double mid, double hi, double fv, int j, double& rr) {
            const double L = fpr_lane_up1z(mid), R = fpr_lane_down1z(mid);
            rr = ((((R + L) + hi) + lo) - C * mid) * _h2 - fv;
            const bool bnd = col_bnd || j <= 0 || j >= ny - 1;
            return bnd ? mid : mid + fac * rr;
        };
        auto step = [&](auto Tc, int r) {
            constexpr int T = decltype(Tc)::value;               // (r - rs) mod 12
            constexpr int Q = T % 4, M = T % 3, F = T % 6;
            constexpr int M1 = (M + 1) % 3, M2 = (M + 2) % 3;    // slots of rows r-2 (= r+1) and r-1
            auto fs = [](int k) { return (F - k + 6) % 6; };     // slot of f row r-k
            double an, fn;
            asm volatile("v_mov_b64 %0, %1" : "=v"(an) : "v"(pu[Q]));
            asm volatile("v_mov_b64 %0, %1" : "=v"(fn) : "v"(pfv[Q]));
            pu[Q] = ldu(r + 1 + PF);                           // issue the loads of row r+1+PF
            pfv[Q] = ldf(r + 1 + PF);
            double rr;
            // ---- cycle k, post-smoothing (:142-143): sweeps 1 and 2 at rows r-1, r-2 ----
            w[1][M2] = sweep(w[0][M1], w[0][M2], w[0][M], fw[fs(1)], r - 1, rr);
            const int j2 = r - 2;
            double u2 = sweep(w[1][M], w[1][M1], w[1][M2], fw[fs(2)], j2, rr);   // u at the end of cycle k (not stored)
            if constexpr (BCS) {
                // apply_boundary_conditions! between the cycles (multigrid.jl:60-62, part2_utils.jl:22-31): its Dirichlet
                // rows hold their values already (set before the first cycle, never changed by a sweep or a correction);
                // its Neumann columns copy their inner neighbour of THIS field
                const double fromR = fpr_lane_down1z(u2), fromL = fpr_lane_up1z(u2);
                u2 = (gi == 0) ? fromR : ((gi == nx - 1) ? fromL : u2);
            }
            w[2][M1] = u2;
            {
                const bool bnd = col_bnd || j2 <= 0 || j2 >= ny - 1;
                const bool row_own = j2 >= y0 && j2 < y1;      // uniform
                acc += (owner && row_own && !bnd) ? rr * rr : 0.0;  // r_rms of cycle k (:252); branch-free (+0.0 is exact)
            }
            // ---- cycle k+1, pre-smoothing (:124-125): sweeps 3 and 4 at rows r-3, r-4 ----
            w[3][M] = sweep(w[2][M2], w[2][M], w[2][M1], fw[fs(3)], r - 3, rr);
            const int j4 = r - 4;
            const double u4 = sweep(w[3][M1], w[3][M2], w[3][M], fw[fs(4)], j4, rr);
            {
                const bool row_own = j4 >= y0 && j4 < y1;      // uniform
                fpr_bst(rUout, vst, row_own ? (j4 - rs) * rowB : (int)FPR_OOR, u4);   // unconditional (see FPR_OOR)
            }
            w[4][M2] = u4;
            // ---- residual of the pre-smoothed field at row r-5, injected at even (row, column) (:128-132) ----
            {
                const int j5 = r - 5;
                const double mid = w[4][M1];
                const double L = fpr_lane_up1z(mid), R = fpr_lane_down1z(mid);
                const double rres = ((((R + L) + w[4][M2]) + w[4][M]) - C * mid) * _h2 - fw[fs(5)];
                const int ic = gi >> 1, jc = j5 >> 1;
                const bool cint = ic >= 1 && ic <= nxc - 2 && jc >= 1 && jc <= nyc - 2;
                const bool row_inj = j5 >= y0 && j5 < y1 && !(j5 & 1);   // uniform
                const int sc = row_inj ? jc * (nxc * 8) : (int)FPR_OOR;
                fpr_bst(rResC, vstr, sc, cint ? rres : 0.0);
                if constexpr (BCS) fpr_bst(rResC, vstn, sc, cint ? rres : 0.0);
                fpr_bst(rCorC, vstc, sc, 0.0);
            }
            w[0][M1] = an;             // row r+1 takes the slot of row r-2
            fw[(F + 1) % 6] = fn;      // row r+1 takes the slot of row r-5
        };
        const int rend = y1 + 4;
        int r = rs;
        static_assert(PF == 4, "the row loop below is unrolled by 12 = lcm(3 window slots, PF = 4, 6 rows of f)");
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
    }
    const double sblk = fpr_block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
}

// ---- S fused Jacobi sweeps, register patches (coarse solve on grids too large for one workgroup) ------
// The coarse-grid Jacobi solve of the "few levels" configurations (e.g. 257^2, 5140 sweeps per V-cycle)
// is latency bound: ~1 MFLOP per sweep.  A workgroup of 16x16 threads holds a 32x32 region in
// registers (2x2 points per thread, f too), exchanges patch edges through a double-buffered LDS image
// (one barrier per sweep), and applies up to S = 8 sweeps per launch; the inner 16x16 tile (S cells away
// from the region's edge) is exact and is the only part written back.  Per-sweep sums of res^2 over the
// own tile are kept in registers and reduced once at the end: partials[s * nblocks + block].
// STATE: the launch first replays the exit test of the PREVIOUS group (prev_nsw sweeps, partial sums in
// prev_partials) -- every workgroup evaluates it redundantly and identically, workgroup 0 records it in
// the solver state -- and does nothing once the criterion has been met.  This removes the separate
// check launch from the dependent chain (one launch per 8 sweeps instead of two).
template <int S, bool NORM, bool STATE>
__global__ __launch_bounds__(256) void k_jacobi_patch(const double* __restrict__ uin, const double* __restrict__ f,
                                                       double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                       double fac, int nsw, double* __restrict__ partials,
                                                       FprSolveState* __restrict__ state,
                                                       const double* __restrict__ prev_partials, int prev_nsw,
                                                       int prev_group, double Ntot)
{
    constexpr int P = 32;  // region = 32 x 32 points, own tile = inner 16 x 16
    constexpr int T = P - 2 * S;
    __shared__ __attribute__((aligned(16))) double img[2][P * P];
    __shared__ double red[4][S];
    __shared__ int stop_flag;
    const int tid = threadIdx.x;
    // the field loads are issued first so that they overlap the replay of the previous group's exit test
    const int tx = tid & 15, ty = tid >> 4;
    const int lx = 2 * tx, ly = 2 * ty;                      // patch origin inside the region
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T;      // own tile origin
    const int gx = x0 - S + lx, gy = y0 - S + ly;            // global coords of the patch origin
    double u[2][2], ff[2][2];
    bool inter[2][2], own[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int gi = gx + a, gj = gy + b;
            const bool in = gi >= 0 && gj >= 0 && gi < nx && gj < ny;
            const size_t g = in ? (size_t)gi + (size_t)nx * gj : 0;
            u[b][a] = in ? uin[g] : 0.0;
            ff[b][a] = in ? f[g] : 0.0;
            inter[b][a] = gi >= 1 && gj >= 1 && gi < nx - 1 && gj < ny - 1;     // updated points
            own[b][a] = in && gi >= x0 && gi < x0 + T && gj >= y0 && gj < y0 + T;
        }
    if constexpr (STATE) {
        // Replay of the previous group's exit test.  All loads (flag, threshold, partial sums) are issued together;
        // wave w sums the partials of sweeps w, w+4 (same order as k_jacobi_check_multi), lanes 0..prev_nsw-1 of
        // wave 0 then evaluate sqrt(sum/N) < thresh in parallel and a ballot finds the first sweep that met it.
        const int done0 = state->done;
        const double thresh = state->thresh;
        const int nblk = gridDim.x * gridDim.y;
        const int lane = tid & 63, wv = tid >> 6;
        double part_acc[2] = {0.0, 0.0};
        if (prev_nsw > 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sidx = wv + 4 * h;
                if (sidx < prev_nsw)
                    for (int i = lane; i < nblk; i += 64) part_acc[h] += prev_partials[(size_t)sidx * nblk + i];
            }
        }
        if (done0) return;
        if (prev_nsw > 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sidx = wv + 4 * h;
                const double a = fpr_wave_sum(part_acc[h]);
                if (lane == 0 && sidx < prev_nsw) red[0][sidx] = a;  // S >= prev_nsw
            }
            __syncthreads();
            if (wv == 0) {
                const double rms = (lane < prev_nsw) ? sqrt(red[0][lane < S ? lane : 0] / Ntot) : 0.0;
                const unsigned long long hit = __ballot(lane < prev_nsw && rms < thresh);
                const int conv = hit ? (int)__builtin_ctzll(hit) : -1;
                const int last = conv >= 0 ? conv : prev_nsw - 1;
                const double rms_last = __shfl(rms, last, 64);
                if (lane == 0) {
                    stop_flag = conv >= 0;
                    if (blockIdx.x == 0 && blockIdx.y == 0) {
                        state->iters += (conv >= 0) ? conv + 1 : prev_nsw;
                        state->last_rms = rms_last;
                        if (conv >= 0) {
                            state->redo = conv + 1;
                            state->group = prev_group;
                            state->done = 1;
                        }
                    }
                }
            }
            __syncthreads();
            if (stop_flag) return;
        }
    }
    double acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = 0.0;
    // clamped neighbour offsets (reads beyond the region return the thread's own edge value: such
    // garbage stays more than S cells away from the own tile)
    const int xl = lx > 0 ? lx - 1 : lx, xr = lx + 2 < P ? lx + 2 : lx + 1;
    const int yd = ly > 0 ? ly - 1 : ly, yu = ly + 2 < P ? ly + 2 : ly + 1;
    {
        double* w = img[0];
        *reinterpret_cast<double2*>(&w[lx + P * ly]) = make_double2(u[0][0], u[0][1]);
        *reinterpret_cast<double2*>(&w[lx + P * (ly + 1)]) = make_double2(u[1][0], u[1][1]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s < nsw) {
            const double* cur = img[s & 1];
            double* nxt = img[(s + 1) & 1];
            const double wl0 = cur[xl + P * ly], wl1 = cur[xl + P * (ly + 1)];
            const double er0 = cur[xr + P * ly], er1 = cur[xr + P * (ly + 1)];
            const double2 dn = *reinterpret_cast<const double2*>(&cur[lx + P * yd]);
            const double2 up = *reinterpret_cast<const double2*>(&cur[lx + P * yu]);
            // E, W, N, S of each patch point
            const double E[2][2] = {{u[0][1], er0}, {u[1][1], er1}};
            const double W[2][2] = {{wl0, u[0][0]}, {wl1, u[1][0]}};
            const double N[2][2] = {{u[1][0], u[1][1]}, {up.x, up.y}};
            const double Sx[2][2] = {{dn.x, dn.y}, {u[0][0], u[0][1]}};
            double un[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const double rr = ((((E[b][a] + W[b][a]) + N[b][a]) + Sx[b][a]) - C * u[b][a]) * _h2 - ff[b][a];
                    un[b][a] = inter[b][a] ? u[b][a] + fac * rr : u[b][a];
                    if constexpr (NORM) {
                        if (inter[b][a] && own[b][a]) acc[s] += rr * rr;
                    }
                }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) u[b][a] = un[b][a];
            *reinterpret_cast<double2*>(&nxt[lx + P * ly]) = make_double2(u[0][0], u[0][1]);
            *reinterpret_cast<double2*>(&nxt[lx + P * (ly + 1)]) = make_double2(u[1][0], u[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
            if (own[b][a]) uout[(size_t)(gx + a) + (size_t)nx * (gy + b)] = u[b][a];
    if constexpr (NORM) {
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double v = fpr_wave_sum(acc[s]);
            if (lane == 0) red[wv][s] = v;
        }
        __syncthreads();
        if (tid < S) {
            const int blk = blockIdx.x + gridDim.x * blockIdx.y, nblk = gridDim.x * gridDim.y;
            partials[(size_t)tid * nblk + blk] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        }
    }
}

// ---- k_smooth2_march, two columns per lane (opt-in; measured slower, see vcycle_level) -------------------------------------------------
// Same algorithm as k_smooth2_march with a strip of 128 columns per wave: a lane holds two adjacent
// columns, loads/stores them with one 16-byte access (the hardware accepts the 8-byte alignment that
// (2^k+1)-wide rows impose), needs a shuffle only for the outer neighbour of each pair, and halves the
// loop and address arithmetic per point.
struct __attribute__((aligned(8))) FprD2 { double x, y; };

template <bool NORM, bool PROLONG, bool RESTRICT>
__global__ __launch_bounds__(256) void k_smooth2_march2(const double* __restrict__ uin, const double* __restrict__ f,
                                                         double* __restrict__ uout, int nx, int ny, double C, double _h2,
                                                         double fac, int rows_per_chunk, int nstrips,
                                                         double* __restrict__ partials, const double* __restrict__ corr_c,
                                                         int apply_BCs, double* __restrict__ res_c_out,
                                                         double* __restrict__ corr_c_out, const int* __restrict__ skip)
{
    if (skip && *skip) return;   // a cycle enqueued ahead of the exit test that ended the loop (FprCycleCtl)
    __shared__ double red[16];
    // Feeder COLUMNS: HXL on the left, HXR on the right.  On rows whose start is only 8-byte aligned (odd rows of
    // an odd-width grid) the lane <-> column mapping is shifted by one column so that every 16-byte access stays
    // 16-byte aligned; the wave then loses its last column on those rows, hence one more feeder on the right.
    // All widths are even so that strips start on even columns.
    constexpr int HX = RESTRICT ? 4 : 2;   // left feeders (3 needed with RESTRICT, rounded up to an even number)
    constexpr int HXR = RESTRICT ? 4 : 4;  // right feeders: needed (2 or 3) + 1 lost column, rounded up to even
    constexpr int SW = 128 - HX - HXR;     // columns owned by a strip
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + w;
    const bool active = strip < nstrips;
    const int g0 = strip * SW - HX + 2 * lane;   // global column of element 0 (element 1 = g0 + 1)
    const int y0 = blockIdx.y * rows_per_chunk;
    const int y1 = (y0 + rows_per_chunk < ny) ? y0 + rows_per_chunk : ny;
    const int rs = y0 - HX < 0 ? 0 : y0 - HX;
    double acc = 0.0;
    if (active) {
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        int gi[2], gic[2], gis[2];
        bool colbnd[2], owner[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            gi[e] = g0 + e;
            gic[e] = gi[e] < 0 ? 0 : (gi[e] > nx - 1 ? nx - 1 : gi[e]);
            colbnd[e] = gi[e] <= 0 || gi[e] >= nx - 1;
            const int q = 2 * lane + e;
            owner[e] = gi[e] >= 0 && gi[e] < nx && q >= HX && q < 128 - HXR;
            gis[e] = gic[e];
            if (PROLONG && apply_BCs) gis[e] = (gic[e] == 0) ? 1 : (gic[e] == nx - 1 ? nx - 2 : gic[e]);
        }
        const bool vec_ok = g0 >= 0 && g0 + 1 < nx;   // both columns exist: one 16-byte access (unshifted rows)
        const bool vec_ok_s = g0 >= 1 && g0 < nx;     // columns g0-1, g0 exist (shifted rows)
        const bool odd_pitch = (nx & 1) != 0;
        // element 1 of the previous lane is owned / exists (needed for the shifted store)
        const bool owner_prev = (2 * lane - 1 >= HX) && (2 * lane - 1 < 128 - HXR) && g0 - 1 >= 0 && g0 - 1 < nx;
        // PROLONG: per element, the two coarse columns it interpolates from, cached for coarse rows pj, pj+1
        int p_icl[2], p_ich[2], p_io[2];
        bool p_sx0[2], p_sx1[2], p_inx[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            p_io[e] = gis[e] & 1;
            p_icl[e] = gis[e] >> 1;
            p_ich[e] = (p_icl[e] + 1 < nxc) ? p_icl[e] + 1 : nxc - 1;
            p_sx0[e] = p_icl[e] >= 1 && p_icl[e] <= nxc - 2;
            p_sx1[e] = p_io[e] && (p_icl[e] + 1 <= nxc - 2);
            p_inx[e] = gis[e] >= 1 && gis[e] <= nx - 2;
        }
        int pj = -2;
        double pc00[2] = {0.0, 0.0}, pc10[2] = {0.0, 0.0}, pc01[2] = {0.0, 0.0}, pc11[2] = {0.0, 0.0};
        auto ld2 = [&](const double* __restrict__ p, int r, double& v0, double& v1) {
            const int rc = r > ny - 1 ? ny - 1 : r;
            const size_t row = (size_t)nx * rc;
            if (odd_pitch && (rc & 1)) {
                // shifted row: this lane fetches columns (g0-1, g0), 16-byte aligned; element 1 (column g0+1) is the
                // first element of the next lane's pair
                double s0, s1;
                if (vec_ok_s) {
                    const double2 t = *reinterpret_cast<const double2*>(p + row + (g0 - 1));
                    s0 = t.x; s1 = t.y;
                } else {
                    const int c0 = g0 - 1 < 0 ? 0 : (g0 - 1 > nx - 1 ? nx - 1 : g0 - 1);
                    s0 = p[row + c0]; s1 = p[row + gic[0]];
                }
                v0 = s1;
                v1 = fpr_lane_down1(s0);
            } else {
                if (vec_ok) {
                    const double2 t = *reinterpret_cast<const double2*>(p + row + g0);
                    v0 = t.x; v1 = t.y;
                } else {
                    v0 = p[row + gic[0]]; v1 = p[row + gic[1]];
                }
            }
        };
        auto ldu = [&](int r, double& v0, double& v1) {
            ld2(uin, r, v0, v1);
            if constexpr (PROLONG) {
                const int rc = r > ny - 1 ? ny - 1 : r;
                const int jo = rc & 1, jcl = rc >> 1;
                if (jcl != pj) {
                    const int jch = (jcl + 1 < nyc) ? jcl + 1 : nyc - 1;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (jcl == pj + 1) { pc00[e] = pc01[e]; pc10[e] = pc11[e]; }
                        else { pc00[e] = corr_c[(size_t)p_icl[e] + (size_t)nxc * jcl]; pc10[e] = corr_c[(size_t)p_ich[e] + (size_t)nxc * jcl]; }
                        pc01[e] = corr_c[(size_t)p_icl[e] + (size_t)nxc * jch];
                        pc11[e] = corr_c[(size_t)p_ich[e] + (size_t)nxc * jch];
                    }
                    pj = jcl;
                }
                const bool sy0 = jcl >= 1 && jcl <= nyc - 2, sy1 = jo && (jcl + 1 <= nyc - 2);
                const bool iny = rc >= 1 && rc <= ny - 2;
                double pv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {  // same value and accumulation order as prolong_bf
                    const bool in = p_inx[e] && iny;
                    const double wgt = (p_io[e] | jo) ? ((p_io[e] & jo) ? 0.25 : 0.5) : 1.0;
                    double t = 0.0;
                    t = t + ((in && p_sx0[e] && sy0) ? wgt * pc00[e] : 0.0);
                    t = t + ((in && p_sx1[e] && sy0) ? wgt * pc10[e] : 0.0);
                    t = t + ((in && p_sx0[e] && sy1) ? wgt * pc01[e] : 0.0);
                    t = t + ((in && p_sx1[e] && sy1) ? wgt * pc11[e] : 0.0);
                    pv[e] = t;
                }
                v0 = v0 - pv[0];
                v1 = v1 - pv[1];
            }
        };
        double a0[2] = {0, 0}, a1[2] = {0, 0}, a2[2], an[2];   // u  rows r-2, r-1, r, r+1
        double b0[2] = {0, 0}, b1[2] = {0, 0}, b2[2] = {0, 0}; // u1 rows r-3, r-2, r-1
        double c0[2] = {0, 0}, c1[2] = {0, 0}, c2[2] = {0, 0}; // u2 rows r-4, r-3, r-2 (RESTRICT)
        double f0[2] = {0, 0}, f1[2] = {0, 0}, f2[2], fn[2], fm[2] = {0, 0};
        ldu(rs, a2[0], a2[1]);
        ld2(f, rs, f2[0], f2[1]);
        // prefetch ring with compile-time slots (see k_smooth2_march): rows r+1 .. r+PF in flight
        constexpr int PF = 4;
        double pu[PF][2], pfv[PF][2];
#pragma unroll
        for (int q = 0; q < PF; ++q) { ldu(rs + 1 + q, pu[q][0], pu[q][1]); ld2(f, rs + 1 + q, pfv[q][0], pfv[q][1]); }
        auto step = [&](auto Qc, int r) {
            constexpr int Q = decltype(Qc)::value;
            an[0] = pu[Q][0]; an[1] = pu[Q][1];
            fn[0] = pfv[Q][0]; fn[1] = pfv[Q][1];
            ldu(r + 1 + PF, pu[Q][0], pu[Q][1]);
            ld2(f, r + 1 + PF, pfv[Q][0], pfv[Q][1]);
            // ---- sweep 1 at row r-1 ----
            const int j1 = r - 1;
            double u1[2];
            {
                const double Lo = fpr_lane_up1(a1[1]), Ro = fpr_lane_down1(a1[0]);
                const bool rowb = j1 <= 0 || j1 >= ny - 1;
                const double rr0 = ((((a1[1] + Lo) + a2[0]) + a0[0]) - C * a1[0]) * _h2 - f1[0];
                const double rr1 = ((((Ro + a1[0]) + a2[1]) + a0[1]) - C * a1[1]) * _h2 - f1[1];
                u1[0] = (rowb || colbnd[0]) ? a1[0] : a1[0] + fac * rr0;
                u1[1] = (rowb || colbnd[1]) ? a1[1] : a1[1] + fac * rr1;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) { b0[e] = b1[e]; b1[e] = b2[e]; b2[e] = u1[e]; }
            // ---- sweep 2 at row r-2 ----
            const int j2 = r - 2;
            {
                const double Lo = fpr_lane_up1(b1[1]), Ro = fpr_lane_down1(b1[0]);
                const bool rowb = j2 <= 0 || j2 >= ny - 1;
                const double rr0 = ((((b1[1] + Lo) + b2[0]) + b0[0]) - C * b1[0]) * _h2 - f0[0];
                const double rr1 = ((((Ro + b1[0]) + b2[1]) + b0[1]) - C * b1[1]) * _h2 - f0[1];
                const bool bn0 = rowb || colbnd[0], bn1 = rowb || colbnd[1];
                const double u20 = bn0 ? b1[0] : b1[0] + fac * rr0;
                const double u21 = bn1 ? b1[1] : b1[1] + fac * rr1;
                const bool shifted_row = odd_pitch && (j2 & 1) && j2 >= 0;
                const double prev21 = fpr_lane_up1(u21);  // column g0-1 (element 1 of the previous lane)
                if (j2 >= y0 && j2 < y1) {
                    const size_t o = (size_t)nx * j2;
                    if (shifted_row) {
                        if (owner_prev && owner[0]) {
                            *reinterpret_cast<double2*>(uout + o + (g0 - 1)) = make_double2(prev21, u20);
                        } else {
                            if (owner_prev) uout[o + g0 - 1] = prev21;
                            if (owner[0]) uout[o + gi[0]] = u20;
                        }
                        // (this lane's element 1, column g0+1, is stored by the next lane as its `prev21`)
                    } else if (owner[0] && owner[1]) {
                        *reinterpret_cast<double2*>(uout + o + g0) = make_double2(u20, u21);
                    } else {
                        if (owner[0]) uout[o + gi[0]] = u20;
                        if (owner[1]) uout[o + gi[1]] = u21;
                    }
                    if constexpr (NORM) {
                        if (owner[0] && !bn0) acc += rr0 * rr0;
                        if (owner[1] && !bn1) acc += rr1 * rr1;
                    }
                }
                if constexpr (RESTRICT) {
                    c0[0] = c1[0]; c1[0] = c2[0]; c2[0] = u20;
                    c0[1] = c1[1]; c1[1] = c2[1]; c2[1] = u21;
                }
            }
            if constexpr (RESTRICT) {
                // ---- residual of u2 at row r-3, injected at even (row, column): one of the lane's two columns ----
                const int j3 = r - 3;
                const double Lo = fpr_lane_up1(c1[1]), Ro = fpr_lane_down1(c1[0]);
                const double rr0 = ((((c1[1] + Lo) + c2[0]) + c0[0]) - C * c1[0]) * _h2 - fm[0];
                const double rr1 = ((((Ro + c1[0]) + c2[1]) + c0[1]) - C * c1[1]) * _h2 - fm[1];
                if (j3 >= y0 && j3 < y1 && !(j3 & 1)) {
                    const int e = (gi[0] & 1) ? 1 : 0;  // the even column
                    const double rr = e ? rr1 : rr0;
                    if (e ? owner[1] : owner[0]) {
                        const int ic = (e ? gi[1] : gi[0]) >> 1, jc = j3 >> 1;
                        const bool cint = ic >= 1 && ic <= nxc - 2 && jc >= 1 && jc <= nyc - 2;
                        const size_t cid = (size_t)ic + (size_t)nxc * jc;
                        res_c_out[cid] = cint ? rr : 0.0;
                        corr_c_out[cid] = 0.0;
                    }
                }
                fm[0] = f0[0]; fm[1] = f0[1];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                a0[e] = a1[e]; a1[e] = a2[e]; a2[e] = an[e];
                f0[e] = f1[e]; f1[e] = f2[e]; f2[e] = fn[e];
            }
        };
        const int rend = y1 + (RESTRICT ? 2 : 1);
        int r = rs;
        for (; r + PF - 1 <= rend; r += PF) {
            step(std::integral_constant<int, 0>{}, r);
            step(std::integral_constant<int, 1>{}, r + 1);
            step(std::integral_constant<int, 2>{}, r + 2);
            step(std::integral_constant<int, 3>{}, r + 3);
        }
        if (r <= rend) { step(std::integral_constant<int, 0>{}, r); ++r; }
        if (r <= rend) { step(std::integral_constant<int, 1>{}, r); ++r; }
        if (r <= rend) { step(std::integral_constant<int, 2>{}, r); ++r; }
    }
    if constexpr (NORM) {
        const double sblk = fpr_block_sum<256>(acc, red);
        if (threadIdx.x == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
    }
}

// ---- residual + injection (+ Neumann rows) into the coarse rhs ------------------------------------
// one thread per COARSE point; (nx, ny) = fine dims.  multigrid.jl:128-129, 330-358
__global__ __launch_bounds__(256) void k_restrict_residual2d(const double* __restrict__ u, const double* __restrict__ f,
                                                              double* __restrict__ res_c, int nx, int ny, double C,
                                                              double _h2, int apply_BCs, double* __restrict__ corr_c)
{
    const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
    const int ic = blockIdx.x * BX + threadIdx.x, jc = blockIdx.y * BY + threadIdx.y;
    if (ic >= nxc || jc >= nyc) return;
    double v = 0.0;
    // interior coarse points take the fine residual at (2ic, 2jc); Neumann copies row 1 / nxc-2
    int is = ic;
    if (apply_BCs) is = (ic == 0) ? 1 : (ic == nxc - 1 ? nxc - 2 : ic);
    if (is >= 1 && is <= nxc - 2 && jc >= 1 && jc <= nyc - 2) {
        const size_t id = (size_t)(2 * is) + (size_t)nx * (size_t)(2 * jc);
        v = res_at(u, f, id, nx, C, _h2);
    }
    res_c[(size_t)ic + (size_t)nxc * jc] = v;
    if (corr_c) corr_c[(size_t)ic + (size_t)nxc * jc] = 0.0;  // `corr_c .= 0.` (multigrid.jl:132) in the same pass
}

// ---- bilinear prolongation as a gather -------------------------------------------------------------
// value the reference's sequential scatter (iy outer, ix inner; multigrid.jl:427-444) leaves at fine (i,j)
__device__ __forceinline__ double prolong_at(const double* __restrict__ cc, int i, int j, int nx, int ny, int nxc)
{
    // sources are interior coarse points only: fine even index in [2, n-3]
    if (i < 1 || j < 1 || i > nx - 2 || j > ny - 2) return 0.0;
    const int io = i & 1, jo = j & 1;
    const int icl = i >> 1, jcl = j >> 1;  // lower coarse neighbour
    const int nyc = 1 + (ny - 1) / 2;
    auto src = [&](int ic, int jc) -> bool { return ic >= 1 && ic <= nxc - 2 && jc >= 1 && jc <= nyc - 2; };
    auto cv = [&](int ic, int jc) -> double { return cc[(size_t)ic + (size_t)nxc * jc]; };
    double v = 0.0;
    if (!io && !jo) {
        if (src(icl, jcl)) v = v + cv(icl, jcl);
    } else if (io && !jo) {  // between two coarse points in x: lower-x source is visited first
        if (src(icl, jcl)) v = v + 0.5 * cv(icl, jcl);
        if (src(icl + 1, jcl)) v = v + 0.5 * cv(icl + 1, jcl);
    } else if (!io && jo) {
        if (src(icl, jcl)) v = v + 0.5 * cv(icl, jcl);
        if (src(icl, jcl + 1)) v = v + 0.5 * cv(icl, jcl + 1);
    } else {  // order: (ic,jc), (ic+1,jc), (ic,jc+1), (ic+1,jc+1)
        if (src(icl, jcl)) v = v + 0.25 * cv(icl, jcl);
        if (src(icl + 1, jcl)) v = v + 0.25 * cv(icl + 1, jcl);
        if (src(icl, jcl + 1)) v = v + 0.25 * cv(icl, jcl + 1);
        if (src(icl + 1, jcl + 1)) v = v + 0.25 * cv(icl + 1, jcl + 1);
    }
    return v;
}

// CORRECT = false: fine = P(coarse) (prolongate_wrapper!)   CORRECT = true: fine -= P(coarse) (:136-139)
template <bool CORRECT>
__global__ __launch_bounds__(256) void k_prolong2d(const double* __restrict__ cc, double* __restrict__ fine, int nx, int ny,
                                                    int apply_BCs)
{
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    if (i >= nx || j >= ny) return;
    const int nxc = 1 + (nx - 1) / 2;
    int is = i;
    if (apply_BCs) is = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i);  // Neumann rows (part2_utils.jl:35-39)
    const double p = prolong_bf(cc, is, j, nx, ny, nxc, 1 + (ny - 1) / 2);
    const size_t id = (size_t)i + (size_t)nx * j;
    if constexpr (CORRECT) fine[id] = fine[id] - p;
    else fine[id] = p;
}

// injection only (restrict_wrapper!, multigrid.jl:344-358)
__global__ __launch_bounds__(256) void k_restrict2d(const double* __restrict__ fine, double* __restrict__ coarse, int nx,
                                                     int ny, int apply_BCs)
{
    const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
    const int ic = blockIdx.x * BX + threadIdx.x, jc = blockIdx.y * BY + threadIdx.y;
    if (ic >= nxc || jc >= nyc) return;
    int is = ic;
    if (apply_BCs) is = (ic == 0) ? 1 : (ic == nxc - 1 ? nxc - 2 : ic);
    double v = 0.0;
    if (is >= 1 && is <= nxc - 2 && jc >= 1 && jc <= nyc - 2) v = fine[(size_t)(2 * is) + (size_t)nx * (size_t)(2 * jc)];
    coarse[(size_t)ic + (size_t)nxc * jc] = v;
}

// ---- B5: (lap - c) T, krylov.jl:7-13 -------------------------------------------------------------
__device__ __forceinline__ double lap_at(const double* __restrict__ T, size_t id, int nx, double hx2, double hy2, double c)
{
    const double t = T[id];
    return (((T[id + 1] - 2 * t) + T[id - 1]) / hx2 + ((T[id + nx] - 2 * t) + T[id - nx]) / hy2) - c * t;
}

__global__ __launch_bounds__(256) void k_laplace2d(const double* __restrict__ T, double* __restrict__ out, int nx, int ny,
                                                    double hx2, double hy2, double c)
{
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    if (i < 1 || j < 1 || i >= nx - 1 || j >= ny - 1) return;
    const size_t id = (size_t)i + (size_t)nx * j;
    out[id] = lap_at(T, id, nx, hx2, hy2, c);
}

// ---- B6: boundary conditions, part2_utils.jl:22-39 ------------------------------------------------
__global__ __launch_bounds__(256) void k_bc_dirichlet(double* __restrict__ T, int nx, int ny, const int* __restrict__ skip)
{
    if (skip && *skip) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx) return;
    T[i] = 1.0;
    T[(size_t)i + (size_t)nx * (ny - 1)] = 0.0;
}
__global__ __launch_bounds__(256) void k_bc_neumann(double* __restrict__ T, int nx, int ny, const int* __restrict__ skip)
{
    if (skip && *skip) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ny) return;
    T[(size_t)nx * j] = T[(size_t)nx * j + 1];
    T[(size_t)nx * j + nx - 1] = T[(size_t)nx * j + nx - 2];
}

// ---- coarse-solver state kernels -------------------------------------------------------------------
// thresh = tol * sqrt(sumsq/N)   (multigrid.jl:150)  /  tol * sqrt(sumsq)   (krylov.jl:57-58)
__global__ void k_state_init(FprSolveState* st, const double* sumsq, double tol, double N, int cg)
{
    st->done = 0;
    st->iters = 0;
    st->redo = 0;
    st->group = -1;
    st->last_rms = 0.0;
    st->thresh = cg ? tol * sqrt(sumsq[0]) : tol * sqrt(sumsq[0] / N);
    st->rho = sumsq[0];  // CG: rho = sum(r.*r) with r = b (krylov.jl:64)
    st->rho2[0] = sumsq[0];
    st->rho2[1] = 0.0;
    st->rho_old = 0.0;
    st->alpha = 0.0;
    st->beta = 0.0;
    st->pq = 0.0;
}

// after a Jacobi sweep: r_rms = sqrt(sum/N); stop when r_rms < thresh (multigrid.jl:152-155)
__global__ __launch_bounds__(256) void k_jacobi_check(FprSolveState* st, const double* __restrict__ partials, int nparts,
                                                       double N)
{
    __shared__ double red[16];
    if (st->done) return;
    const double s = fpr_sum_partials_256(partials, nparts, red);
    if (threadIdx.x == 0) {
        const double rms = sqrt(s / N);
        st->iters += 1;
        st->last_rms = rms;
        if (rms < st->thresh) st->done = 1;
    }
}

// after a group of nsw fused sweeps: replay the per-sweep exit test in order (multigrid.jl:152-155)
__global__ __launch_bounds__(256) void k_jacobi_check_multi(FprSolveState* st, const double* __restrict__ partials, int nblk,
                                                             int nsw, double N, int group)
{
    __shared__ double sums[16];
    if (st->done) return;
    // wave w sums the partials of sweeps w, w+4, ... in a fixed order; then thread 0 replays the tests
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int s = wv; s < nsw; s += 4) {
        double a = 0.0;
        for (int i = lane; i < nblk; i += 64) a += partials[(size_t)s * nblk + i];
        a = fpr_wave_sum(a);
        if (lane == 0) sums[s] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int it = st->iters;
        for (int s = 0; s < nsw; ++s) {
            const double rms = sqrt(sums[s] / N);
            it += 1;
            st->last_rms = rms;
            if (rms < st->thresh) {
                st->done = 1;
                st->redo = s + 1;
                st->group = group;
                break;
            }
        }
        st->iters = it;
    }
}

// ---- CG kernels (krylov.jl:55-91) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cg_init(const double* __restrict__ b, double* __restrict__ r, double* __restrict__ p,
                                                  double* __restrict__ ph, double* __restrict__ x, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = b[i];
        r[i] = v; p[i] = v; ph[i] = v; x[i] = 0.0;
    }
}

// p_hat = A p on the interior (boundary of p_hat keeps b's values), partial sums of p .* p_hat over
// the WHOLE array (krylov.jl:68-69)
__global__ __launch_bounds__(256) void k_cg_matvec_dot(const double* __restrict__ p, double* __restrict__ ph, int nx, int ny,
                                                        double hx2, double hy2, double c, double* __restrict__ partials,
                                                        const FprSolveState* __restrict__ st)
{
    __shared__ double red[16];
    if (st->done) return;
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    double acc = 0.0;
    if (i < nx && j < ny) {
        const size_t id = (size_t)i + (size_t)nx * j;
        double q;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            q = lap_at(p, id, nx, hx2, hy2, c);
            ph[id] = q;
        } else {
            q = ph[id];
        }
        acc = p[id] * q;
    }
    const double s = fpr_block_sum<256>(acc, red);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[blockIdx.x + gridDim.x * blockIdx.y] = s;
}

__global__ __launch_bounds__(256) void k_cg_alpha(FprSolveState* st, const double* __restrict__ partials, int nparts)
{
    __shared__ double red[16];
    if (st->done) return;
    const double s = fpr_sum_partials_256(partials, nparts, red);
    if (threadIdx.x == 0) {
        st->pq = s;
        st->alpha = st->rho / s;  // krylov.jl:69
    }
}

// x .+= alpha p ; r .-= alpha p_hat ; partial sums of r.^2   (krylov.jl:70-72)
__global__ __launch_bounds__(256) void k_cg_update(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                    const double* __restrict__ ph, size_t n, double* __restrict__ partials,
                                                    const FprSolveState* __restrict__ st)
{
    __shared__ double red[16];
    if (st->done) return;
    const double alpha = st->alpha;
    const size_t stride = (size_t)gridDim.x * 256;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        x[i] = x[i] + alpha * p[i];
        const double rn = r[i] - alpha * ph[i];
        r[i] = rn;
        acc += rn * rn;
    }
    const double s = fpr_block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cg_check(FprSolveState* st, const double* __restrict__ partials, int nparts, double N)
{
    __shared__ double red[16];
    if (st->done) return;
    const double s = fpr_sum_partials_256(partials, nparts, red);
    if (threadIdx.x == 0) {
        const double normr = sqrt(s);
        st->iters += 1;
        st->last_rms = sqrt(s / N);  // krylov.jl:90
        if (normr < st->thresh) {
            st->done = 1;  // krylov.jl:76-81
        } else {
            st->rho_old = st->rho;
            st->rho = s;                      // krylov.jl:83
            st->beta = st->rho / st->rho_old; // krylov.jl:84
        }
    }
}

// ---- fused CG iteration: 3 dependent launches instead of 5 ---------------------------------------------
// k_cg_matvec_dot -> k_cg_update_f (every workgroup derives alpha from the dot partials) ->
// k_cg_p_f (every workgroup derives ||r||, the exit test and beta from the r.r partials).
// rho is double-buffered by iteration parity so that workgroup 0 can publish the new value while the
// others still read the old one.  `it` = 0-based iteration index.
__global__ __launch_bounds__(256) void k_cg_update_f(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                      const double* __restrict__ ph, size_t n, const double* __restrict__ pq_partials,
                                                      int npq, double* __restrict__ partials, FprSolveState* __restrict__ st, int it)
{
    __shared__ double red[16];
    __shared__ double s_alpha;
    if (st->done) return;
    // first element of this thread's grid-stride sequence: loaded before alpha is known (the loads overlap the reduction
    // of the dot-product partials; coarse grids have at most one element per thread)
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    double x0 = 0.0, p0 = 0.0, r0 = 0.0, q0 = 0.0;
    if (i0 < n) { x0 = x[i0]; p0 = p[i0]; r0 = r[i0]; q0 = ph[i0]; }
    const double pq = fpr_sum_partials_256(pq_partials, npq, red);
    if (threadIdx.x == 0) {
        const double alpha = st->rho2[it & 1] / pq;  // krylov.jl:69
        s_alpha = alpha;
        if (blockIdx.x == 0) { st->pq = pq; st->alpha = alpha; }
    }
    __syncthreads();
    const double alpha = s_alpha;
    double acc = 0.0;
    if (i0 < n) {
        x[i0] = x0 + alpha * p0;
        const double rn = r0 - alpha * q0;
        r[i0] = rn;
        acc += rn * rn;
    }
    for (size_t i = i0 + stride; i < n; i += stride) {
        x[i] = x[i] + alpha * p[i];
        const double rn = r[i] - alpha * ph[i];
        r[i] = rn;
        acc += rn * rn;
    }
    __syncthreads();
    const double sblk = fpr_block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = sblk;
}

__global__ __launch_bounds__(256) void k_cg_p_f(double* __restrict__ p, const double* __restrict__ r, size_t n,
                                                 const double* __restrict__ rr_partials, int nrr, FprSolveState* __restrict__ st,
                                                 int it, double N)
{
    __shared__ double red[16];
    __shared__ double s_beta;
    __shared__ int s_conv;
    if (st->done) return;
    const double rr = fpr_sum_partials_256(rr_partials, nrr, red);
    if (threadIdx.x == 0) {
        const double normr = sqrt(rr);
        const int conv = normr < st->thresh;          // krylov.jl:76
        const double rho_old = st->rho2[it & 1];
        const double beta = rr / rho_old;             // krylov.jl:83-84
        s_conv = conv;
        s_beta = beta;
        if (blockIdx.x == 0) {
            st->iters = it + 1;
            st->last_rms = sqrt(rr / N);              // krylov.jl:90
            if (conv) st->done = 1;
            else { st->rho2[(it + 1) & 1] = rr; st->rho_old = rho_old; st->rho = rr; st->beta = beta; }
        }
    }
    __syncthreads();
    if (s_conv) return;
    const double beta = s_beta;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = r[i] + beta * p[i];  // krylov.jl:85
}

// ---- CG iteration in TWO dependent launches ---------------------------------------------------------------
// k_cg_pmv_f (this kernel) -> k_cg_update_f.  The direction update p = r + beta p of iteration it-1 (krylov.jl:85)
// moves into the matvec of iteration it: every workgroup derives ||r||, the exit test and beta from the r.r partials
// of the previous iteration (as k_cg_p_f does), forms the new p on its 32 x 8 tile plus a one-point ring (the ring is
// recomputed, not communicated: p is double-buffered, nobody reads what a neighbour is writing), stores the tile,
// applies the operator from the LDS image and reduces p .* p_hat.  Same operations on the same operands as the
// three-launch form: x, r, p, the iteration count and the returned residual are bit-identical.
constexpr int CGX = BX, CGY = BY;   // the tiles of k_cg_matvec_dot: identical dot-product partials, identical alpha
__global__ __launch_bounds__(256) void k_cg_pmv_f(const double* __restrict__ p_old, double* __restrict__ p_new,
                                                   const double* __restrict__ r, double* __restrict__ ph, int nx, int ny,
                                                   double hx2, double hy2, double c, double* __restrict__ pq_partials,
                                                   const double* __restrict__ rr_partials, int nrr,
                                                   FprSolveState* __restrict__ st, int it, double N)
{
    __shared__ double red[16];
    __shared__ double s_beta;
    __shared__ int s_conv;
    __shared__ double tile[CGY + 2][CGX + 2];
    if (st->done) return;
    const int tid = threadIdx.x + CGX * threadIdx.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int i0 = blockIdx.x * CGX, j0 = blockIdx.y * CGY;
    // operands first: their loads do not depend on beta and overlap the reduction of the r.r partials below
    const int i = i0 + tx, j = j0 + ty;
    const bool own = i < nx && j < ny;
    const size_t id = (size_t)(own ? i : 0) + (size_t)nx * (own ? j : 0);
    const double p_own = p_old[id], r_own = it > 0 ? r[id] : 0.0;
    constexpr int NRING = 2 * (CGX + 2) + 2 * CGY;   // bottom row, top row, left column, right column
    int hx = 0, hy = 0;
    if (tid < CGX + 2) { hx = tid; hy = 0; }
    else if (tid < 2 * (CGX + 2)) { hx = tid - (CGX + 2); hy = CGY + 1; }
    else if (tid < 2 * (CGX + 2) + CGY) { hx = 0; hy = tid - 2 * (CGX + 2) + 1; }
    else if (tid < NRING) { hx = CGX + 1; hy = tid - 2 * (CGX + 2) - CGY + 1; }
    const int ri = i0 + hx - 1, rj = j0 + hy - 1;
    const bool ring = tid < NRING && ri >= 0 && rj >= 0 && ri < nx && rj < ny;
    const size_t rid = (size_t)(ring ? ri : 0) + (size_t)nx * (ring ? rj : 0);
    const double p_ring = p_old[rid], r_ring = it > 0 ? r[rid] : 0.0;
    double beta = 0.0;
    if (it > 0) {   // exit test and beta of iteration it-1 (krylov.jl:73-84)
        double sacc = 0.0;   // fpr_sum_partials_256 for a 64 x 4 block: strided accumulation by linear thread id
        for (int q = tid; q < nrr; q += 256) sacc += rr_partials[q];
        const double rr = fpr_block_sum<256>(sacc, red);
        if (tid == 0) {
            const double normr = sqrt(rr);
            const int conv = normr < st->thresh;              // krylov.jl:76
            const double rho_old = st->rho2[(it - 1) & 1];
            const double b = rr / rho_old;                    // krylov.jl:83-84
            s_conv = conv;
            s_beta = b;
            if (blockIdx.x == 0 && blockIdx.y == 0) {
                st->iters = it;
                st->last_rms = sqrt(rr / N);                  // krylov.jl:90
                if (conv) st->done = 1;
                else { st->rho2[it & 1] = rr; st->rho_old = rho_old; st->rho = rr; st->beta = b; }
            }
        }
        __syncthreads();
        if (s_conv) return;
        beta = s_beta;
    }
    // krylov.jl:85 (p = r = b before the first iteration)
    tile[ty + 1][tx + 1] = own ? (it > 0 ? r_own + beta * p_own : p_own) : 0.0;
    if (tid < NRING) tile[hy][hx] = ring ? (it > 0 ? r_ring + beta * p_ring : p_ring) : 0.0;
    __syncthreads();
    double acc = 0.0;
    if (own) {
        const double t = tile[ty + 1][tx + 1];
        p_new[id] = t;
        double q;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            q = (((tile[ty + 1][tx + 2] - 2 * t) + tile[ty + 1][tx]) / hx2 + ((tile[ty + 2][tx + 1] - 2 * t) + tile[ty][tx + 1]) / hy2) - c * t;   // lap_at
            ph[id] = q;
        } else {
            q = ph[id];   // boundary of p_hat keeps b's values (krylov.jl:61, 68)
        }
        acc = t * q;
    }
    const double sblk = fpr_block_sum<256>(acc, red);
    if (tid == 0) pq_partials[blockIdx.x + gridDim.x * blockIdx.y] = sblk;
}

// exit test of the LAST enqueued iteration (its successor's k_cg_pmv_f would have made it): one workgroup
__global__ __launch_bounds__(256) void k_cg_tail_f(const double* __restrict__ rr_partials, int nrr, FprSolveState* __restrict__ st,
                                                    int it, double N)
{
    __shared__ double red[16];
    if (st->done) return;
    const double rr = fpr_sum_partials_256(rr_partials, nrr, red);
    if (threadIdx.x == 0) {
        const double normr = sqrt(rr);
        const double rho_old = st->rho2[(it - 1) & 1];
        st->iters = it;
        st->last_rms = sqrt(rr / N);
        if (normr < st->thresh) st->done = 1;
        else { st->rho2[it & 1] = rr; st->rho_old = rho_old; st->rho = rr; st->beta = rr / rho_old; }
    }
}

// p .= r + beta p  (krylov.jl:85)
__global__ __launch_bounds__(256) void k_cg_p(double* __restrict__ p, const double* __restrict__ r, size_t n,
                                               const FprSolveState* __restrict__ st)
{
    if (st->done) return;
    const double beta = st->beta;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = r[i] + beta * p[i];
}

// ================================================================================================
// k_mg_small: a whole sub-hierarchy of the V-cycle in ONE workgroup, resident in LDS.
//
// Coarse multigrid levels are launch-latency bound (a 65^2 level costs ~8 dependent launches of ~2 us
// each for ~0.1 us of work, and the 5x5 Jacobi solve another ~130).  Once a level and everything below
// it fit in the CU's 160 KiB LDS (3 arrays per level: u, rhs, ping-pong partner; sum over levels
// <= 20 000 doubles, i.e. up to 65x65 or 129x33), one 1024-thread workgroup executes
// Vcycle_2DPoisson! (multigrid.jl:91-170) for that level and all coarser ones with __syncthreads()
// between the passes: pre-smoothing, residual+injection, the Jacobi coarse solve with its early exit
// (:147-159), prolongation+correction, post-smoothing.  Same pointwise arithmetic as the per-level
// kernels (bit-identical fields); norms are block-tree sums.
// ================================================================================================
struct MgSmallArgs {
    double* u;          // top level of the sub-hierarchy, global memory, in/out
    const double* rhs;  // its right-hand side
    int nx, ny, nlev;   // nlev = levels including the coarsest one
    double h, c, tol;
    int css, apply_BCs, want_norm;
    double* out_sumsq;  // want_norm: sum(res.^2) of the last post-smoothing sweep of the top level
    FprSolveState* state;
    const int* skip;    // cycles enqueued ahead: return at once if *skip (null = unconditional)
    int row_solve;      // 1: coarsest grids with <= 16 interior points are solved inside one DPP row (option mg_small_row)
    long long* prof;    // diagnostic (option mg_small_prof): wall_clock64 stamps (100 MHz) of thread 0 at the section borders
};

constexpr int MGS_NT = 1024;
constexpr int MGS_RED = 32;  // doubles reserved for reductions / broadcasts

__device__ __forceinline__ double mgs_block_sum(double v, double* red)
{
    // all MGS_NT threads call; returns the total in every thread
    v = fpr_wave_sum(v);
    const int tid = threadIdx.x;
    __syncthreads();  // protect red[] from the previous use
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        double s = red[0];
#pragma unroll
        for (int w = 1; w < MGS_NT / 64; ++w) s += red[w];
        red[MGS_NT / 64] = s;
    }
    __syncthreads();
    return red[MGS_NT / 64];
}

// shifts inside a 16-lane DPP row (zero where no lane is the source); n is uniform, 1..15
template <int CTRL>
__device__ __forceinline__ double mgs_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N> __device__ __forceinline__ double mgs_row_ror(double v) { return mgs_dpp<0x120 + N>(v); }
// idx -> row j = idx / nx without an integer division: floor((idx + 0.5) * (1/nx)) in float is exact while
// nx*ny*2.4e-7 < 0.5 (the rounding error of the product stays below the distance 0.5/nx of (idx + 0.5)/nx from an
// integer); the LDS arena holds 20000 doubles, so N < 2^15 here.
__device__ __forceinline__ int mgs_row(int idx, float rnx) { return (int)(((float)idx + 0.5f) * rnx); }

// uout = uin + fac*res(uin) on the interior, boundary copied; returns this thread's sum of res^2
__device__ __forceinline__ double mgs_sweep(const double* uin, const double* f, double* uout, int nx, int ny, double C,
                                            double _h2, double fac)
{
    double acc = 0.0;
    const int N = nx * ny;
    const float rnx = 1.0f / (float)nx;
    for (int idx = threadIdx.x; idx < N; idx += MGS_NT) {
        const int j = mgs_row(idx, rnx), i = idx - j * nx;
        const double uc = uin[idx];
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const double r = ((((uin[idx + 1] + uin[idx - 1]) + uin[idx + nx]) + uin[idx - nx]) - C * uc) * _h2 - f[idx];
            uout[idx] = uc + fac * r;
            acc += r * r;
        } else {
            uout[idx] = uc;
        }
    }
    return acc;
}

__global__ __launch_bounds__(MGS_NT) void k_mg_small(MgSmallArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (a.skip && *a.skip) return;
    double* red = sm;
    double* arena = sm + MGS_RED;
    const int tid = threadIdx.x;
    int pslot = 0;
    auto stamp = [&]() { if (a.prof && tid == 0) a.prof[pslot++] = wall_clock64(); };
    stamp();   // 0: start

    // level d: dims ((nx-1)>>d)+1, arrays u|f|t at arena + off(d)
    auto lnx = [&](int d) { return ((a.nx - 1) >> d) + 1; };
    auto lny = [&](int d) { return ((a.ny - 1) >> d) + 1; };
    auto off = [&](int d) {
        int o = 0;
        for (int e = 0; e < d; ++e) o += 3 * lnx(e) * lny(e);
        return o;
    };
    auto hlev = [&](int d) {
        double h = a.h;
        for (int e = 0; e < d; ++e) h = h * 2;  // the recursion passes h*2 (multigrid.jl:133)
        return h;
    };

    {   // load the top level
        const int N = a.nx * a.ny;
        double* U = arena;
        double* F = arena + N;
        for (int idx = tid; idx < N; idx += MGS_NT) {
            U[idx] = a.u[idx];
            F[idx] = a.rhs[idx];
        }
    }
    __syncthreads();
    stamp();   // 1: top level loaded

    const int c_lev = a.nlev - 1;
    // ---- down sweep ----
    for (int d = 0; d < c_lev; ++d) {
        const int nx = lnx(d), ny = lny(d), N = nx * ny;
        double* U = arena + off(d);
        double* F = U + N;
        double* T = F + N;
        const double h = hlev(d);
        const double C = 4.0 + a.c * (h * h), _h2 = 1 / (h * h);
        const double fac = (4.0 / 5.0) * ((h * h) / (4.0 + a.c * (h * h)));
        mgs_sweep(U, F, T, nx, ny, C, _h2, fac);  // :124
        __syncthreads();
        mgs_sweep(T, F, U, nx, ny, C, _h2, fac);  // :125
        __syncthreads();
        // residual + injection + Neumann rows into the next level's rhs; next level's u = 0 (:128-132)
        const int nxc = lnx(d + 1), nyc = lny(d + 1), Nc = nxc * nyc;
        double* Uc = arena + off(d + 1);
        double* Fc = Uc + Nc;
        const float rnxc = 1.0f / (float)nxc;
        for (int idx = tid; idx < Nc; idx += MGS_NT) {
            const int jc = mgs_row(idx, rnxc), ic = idx - jc * nxc;
            int is = ic;
            if (a.apply_BCs) is = (ic == 0) ? 1 : (ic == nxc - 1 ? nxc - 2 : ic);
            double v = 0.0;
            if (is >= 1 && is <= nxc - 2 && jc >= 1 && jc <= nyc - 2) {
                const int id = 2 * is + nx * (2 * jc);
                v = ((((U[id + 1] + U[id - 1]) + U[id + nx]) + U[id - nx]) - C * U[id]) * _h2 - F[id];
            }
            Fc[idx] = v;
            Uc[idx] = 0.0;
        }
        __syncthreads();
        stamp();   // 2 .. 1+c_lev: level d went down
    }

    // ---- coarsest level: Jacobi with early exit (:147-159) ----
    double* ucur;  // where the coarse solution ends up
    {
        const int nx = lnx(c_lev), ny = lny(c_lev), N = nx * ny;
        double* U = arena + off(c_lev);
        double* F = U + N;
        double* T = F + N;
        const double h = hlev(c_lev);
        const double C = 4.0 + a.c * (h * h), _h2 = 1 / (h * h);
        const double fac = (4.0 / 5.0) * ((h * h) / (4.0 + a.c * (h * h)));
        double acc = 0.0;
        for (int idx = tid; idx < N; idx += MGS_NT) acc += F[idx] * F[idx];
        const double tol_rhs = a.tol * sqrt(mgs_block_sum(acc, red) / (double)N);  // :150
        const int iters = 20 * a.css;
        double res_rms = 0.0;
        int it = 0;
        double* pin = U;
        double* pout = T;
        const int nxi = nx - 2, nyi = ny - 2, ni = nxi * nyi;
        if (N <= 64 && ni >= 1 && ni <= 16 && a.row_solve) {
            // coarsest grid with at most 16 INTERIOR points (5x5 -> 3x3, the default coarse_solve_size): the interior lives in
            // the first lanes of wave 0, one point per lane (lane = (i-1) + nxi*(j-1)), all of it inside one 16-lane DPP row:
            // neighbours by row shifts (a boundary neighbour is a per-lane constant), the norm by four rotate-and-add steps
            // (lane 0's order of summation; uniform through readfirstlane).  sqrt and the division of :157 are only evaluated
            // when the exit test can possibly hold: sum > N * tol_rhs^2 * (1 + 1e-10) implies sqrt(sum/N) > tol_rhs.
            if (tid < 64) {
                const int lane = tid;
                const bool in = lane < ni;
                const int jj = in ? lane / nxi : 0, ii = in ? lane - jj * nxi : 0;   // interior coordinates, once
                const int g = (ii + 1) + nx * (jj + 1);
                const double fv = in ? F[g] : 0.0;
                double uu = in ? U[g] : 0.0;
                // boundary neighbours are constants (Dirichlet copy); interior ones come from the row shifts
                const bool iE = in && ii + 1 < nxi, iW = in && ii > 0, iN = in && jj + 1 < nyi, iS = in && jj > 0;
                const double cE = (in && !iE) ? U[g + 1] : 0.0, cW = (in && !iW) ? U[g - 1] : 0.0;
                const double cN = (in && !iN) ? U[g + nx] : 0.0, cS = (in && !iS) ? U[g - nx] : 0.0;
                const double hi_thr = ((double)N * (tol_rhs * tol_rhs)) * (1.0 + 1e-10);
                double sq_last = 0.0;
                bool have_rms = false;
                // the DPP control is an immediate: the loop is instantiated per interior width (dispatch ONCE, outside it)
                auto run = [&](auto NXIc) {
                    constexpr int NXI = decltype(NXIc)::value;
                    // The exit test of sweep k is evaluated while sweep k+1 is already in flight (its update is dropped if the
                    // test holds): the dependent chain of a sweep is then its stencil alone, not stencil + reduction + test.
                    double sq_vec = 0.0;     // per-lane total of sweep k-1 (every lane of row 0 holds a full sum)
                    bool done = false;
                    auto test = [&](int ksweep) {   // :157-158 for sweep `ksweep`, whose sum sits in sq_vec
                        const double sq = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sq_vec)),
                                                           __builtin_amdgcn_readfirstlane(__double2loint(sq_vec)));
                        it = ksweep;
                        sq_last = sq;
                        have_rms = false;
                        if (sq > hi_thr) return false;      // cannot have converged
                        res_rms = sqrt(sq / (double)N);     // :157
                        have_rms = true;
                        return res_rms < tol_rhs;
                    };
                    for (int k = 1; k <= iters; ++k) {
                        const double sE = mgs_dpp<0x101>(uu), sW = mgs_dpp<0x111>(uu);                 // lanes i+1, i-1
                        const double sN = mgs_dpp<0x100 + NXI>(uu), sS = mgs_dpp<0x110 + NXI>(uu);     // lanes i+nxi, i-nxi
                        const double E = iE ? sE : cE, W = iW ? sW : cW, Nn = iN ? sN : cN, Ss = iS ? sS : cS;
                        const double r = ((((E + W) + Nn) + Ss) - C * uu) * _h2 - fv;
                        const double uu_new = in ? uu + fac * r : uu;
                        double sq = in ? r * r : 0.0;
                        sq += mgs_row_ror<8>(sq);
                        sq += mgs_row_ror<4>(sq);
                        sq += mgs_row_ror<2>(sq);
                        sq += mgs_row_ror<1>(sq);
                        if (k > 1 && test(k - 1)) { done = true; break; }   // uu is still the field after sweep k-1
                        uu = uu_new;
                        sq_vec = sq;
                    }
                    if (!done) test(iters);
                };
#define FPR_ROW_CASE(n) case n: run(std::integral_constant<int, n>{}); break;
                switch (nxi) {
                    FPR_ROW_CASE(1) FPR_ROW_CASE(2) FPR_ROW_CASE(3) FPR_ROW_CASE(4) FPR_ROW_CASE(5) FPR_ROW_CASE(6) FPR_ROW_CASE(7)
                    FPR_ROW_CASE(8) FPR_ROW_CASE(9) FPR_ROW_CASE(10) FPR_ROW_CASE(11) FPR_ROW_CASE(12) FPR_ROW_CASE(13)
                    FPR_ROW_CASE(14) FPR_ROW_CASE(15)
                default: run(std::integral_constant<int, 15>{}); break;   // nxi = 16: one row of points, iN = iS = false everywhere
                }
#undef FPR_ROW_CASE
                if (!have_rms) res_rms = sqrt(sq_last / (double)N);
                if (in) U[g] = uu;
                if (tid == 0) { red[MGS_RED - 1] = res_rms; red[MGS_RED - 2] = (double)it; }
            }
            __syncthreads();
            res_rms = red[MGS_RED - 1];
            it = (int)red[MGS_RED - 2];
        } else
        if (N <= 64) {
            // tiny coarsest grid (5x5, 9x5, ...): one point per lane of wave 0, all in registers --
            // neighbours by wavefront shuffles, norm by a butterfly (every lane gets the same bits), no barrier
            if (tid < 64) {
                const int lane = tid;
                const bool in = lane < N;
                const int j = lane / nx, i = lane - j * nx;
                const bool inter = in && i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1;
                double uu = in ? U[lane] : 0.0;
                const double fv = in ? F[lane] : 0.0;
                for (int k = 1; k <= iters; ++k) {
                    const double E = __shfl(uu, lane + 1, 64), W = __shfl(uu, lane - 1, 64);
                    const double Nn = __shfl(uu, lane + nx, 64), Ss = __shfl(uu, lane - nx, 64);
                    const double r = ((((E + W) + Nn) + Ss) - C * uu) * _h2 - fv;
                    double sq = inter ? r * r : 0.0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
                    res_rms = sqrt(sq / (double)N);
                    if (inter) uu = uu + fac * r;
                    it = k;
                    if (res_rms < tol_rhs) break;
                }
                if (in) U[lane] = uu;
                if (tid == 0) { red[MGS_RED - 1] = res_rms; red[MGS_RED - 2] = (double)it; }
            }
            __syncthreads();
            res_rms = red[MGS_RED - 1];
            it = (int)red[MGS_RED - 2];
        } else
        for (int i = 1; i <= iters; ++i) {
            const double s = mgs_block_sum(mgs_sweep(pin, F, pout, nx, ny, C, _h2, fac), red);  // syncs inside
            res_rms = sqrt(s / (double)N);
            double* t = pin; pin = pout; pout = t;
            it = i;
            if (res_rms < tol_rhs) break;  // uniform: every thread holds the same value
        }
        ucur = pin;
        if (tid == 0) {
            a.state->acc_iters += it;
            a.state->last_rms = res_rms;
        }
        __syncthreads();
        stamp();   // coarsest level solved
    }

    // ---- up sweep ----
    for (int d = c_lev - 1; d >= 0; --d) {
        const int nx = lnx(d), ny = lny(d), N = nx * ny;
        double* U = arena + off(d);
        double* F = U + N;
        double* T = F + N;
        const double h = hlev(d);
        const double C = 4.0 + a.c * (h * h), _h2 = 1 / (h * h);
        const double fac = (4.0 / 5.0) * ((h * h) / (4.0 + a.c * (h * h)));
        const int nxc = lnx(d + 1);
        const double* Uc = (d + 1 == c_lev) ? ucur : arena + off(d + 1);
        const float rnx = 1.0f / (float)nx;
        for (int idx = tid; idx < N; idx += MGS_NT) {  // prolongation + correction (:136-139)
            const int j = mgs_row(idx, rnx), i = idx - j * nx;
            int is = i;
            if (a.apply_BCs) is = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i);
            U[idx] = U[idx] - prolong_bf(Uc, is, j, nx, ny, nxc, lny(d + 1));   // branch-free form of prolong_at: same value
        }
        __syncthreads();
        mgs_sweep(U, F, T, nx, ny, C, _h2, fac);  // :142
        __syncthreads();
        const double acc = mgs_sweep(T, F, U, nx, ny, C, _h2, fac);  // :143
        if (d == 0 && a.want_norm) {
            const double s = mgs_block_sum(acc, red);
            if (tid == 0) a.out_sumsq[0] = s;
        }
        __syncthreads();
        stamp();   // level d came up
    }

    {   // store the top level's solution
        const int N = a.nx * a.ny;
        const double* U = (c_lev == 0) ? ucur : arena;
        for (int idx = tid; idx < N; idx += MGS_NT) a.u[idx] = U[idx];
    }
    stamp();   // stored
}

// ================================================================================================
// host side
// ================================================================================================
static inline int flat_grid(size_t n)
{
    size_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

static int check_dims(fpr_ctx* ctx, int nx, int ny)
{
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3, "grid must be at least 3x3");
    return FPR_OK;
}

static int read_state(fpr_ctx* ctx)
{
    FPR_HIP(ctx, hipMemcpyAsync(ctx->state_h, ctx->state, sizeof(FprSolveState), hipMemcpyDeviceToHost, ctx->stream[0]));
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
    return FPR_OK;
}

static int read_scalar(fpr_ctx* ctx, const double* dev, double* out_host)
{
    FPR_HIP(ctx, hipMemcpyAsync(ctx->host_scalars, dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream[0]));
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
    *out_host = ctx->host_scalars[0];
    return FPR_OK;
}

extern "C" int fpr_residual2d(fpr_ctx* ctx, const double* u, const double* f, double h, double c, double* res, int nx, int ny)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, u && f && res, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    const double C = 4.0 + c * (h * h), _h2 = 1 / (h * h);
    k_residual2d<<<grid2(nx, ny), blk2, 0, ctx->stream[0]>>>(u, f, res, nx, ny, C, _h2);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_jacobi2d(fpr_ctx* ctx, double* u, const double* f, double h, double c, double* res, int nx, int ny,
                            double alpha, double* rms_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, u && f && res, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    const size_t N = (size_t)nx * ny;
    const double C = 4.0 + c * (h * h), _h2 = 1 / (h * h);
    const double fac = alpha * ((h * h) / (4.0 + c * (h * h)));
    k_residual2d<<<grid2(nx, ny), blk2, 0, ctx->stream[0]>>>(u, f, res, nx, ny, C, _h2);
    FPR_CHECK_LAUNCH(ctx);
    if (rms_host) {
        // whole array, including whatever the caller keeps on res' boundary (multigrid.jl:252)
        if (int rc = fprx_sumsq_scaled_dev(ctx, res, N, 1.0, ctx->scalars + 1, 0)) return rc;
    }
    k_axpy_inplace<<<flat_grid(N), 256, 0, ctx->stream[0]>>>(u, res, fac, N);
    FPR_CHECK_LAUNCH(ctx);
    if (rms_host) {
        double s;
        if (int rc = read_scalar(ctx, ctx->scalars + 1, &s)) return rc;
        *rms_host = sqrt(s / (double)N);
    }
    return FPR_OK;
}

extern "C" int fpr_restrict2d(fpr_ctx* ctx, const double* fine, double* coarse, int nx, int ny, int apply_BCs)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, fine && coarse, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    FPR_REQUIRE(ctx, (nx & 1) && (ny & 1), "fine dims must be odd");
    k_restrict2d<<<grid2(1 + (nx - 1) / 2, 1 + (ny - 1) / 2), blk2, 0, ctx->stream[0]>>>(fine, coarse, nx, ny, apply_BCs);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_prolongate2d(fpr_ctx* ctx, const double* coarse, double* fine, int nx, int ny, int apply_BCs)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, fine && coarse, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    FPR_REQUIRE(ctx, (nx & 1) && (ny & 1), "fine dims must be odd");
    k_prolong2d<false><<<grid2(nx, ny), blk2, 0, ctx->stream[0]>>>(coarse, fine, nx, ny, apply_BCs);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_laplace_apply2d(fpr_ctx* ctx, const double* T, double hx, double hy, double c, double* dT2, int nx, int ny)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, T && dT2, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    k_laplace2d<<<grid2(nx, ny), blk2, 0, ctx->stream[0]>>>(T, dT2, nx, ny, hx * hx, hy * hy, c);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_bc_dirichlet2d(fpr_ctx* ctx, double* T, int nx, int ny)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, T && nx >= 1 && ny >= 1, "bad array");
    k_bc_dirichlet<<<(nx + 255) / 256, 256, 0, ctx->stream[0]>>>(T, nx, ny, ctx->cyc_skip);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_bc_neumann2d(fpr_ctx* ctx, double* T, int nx, int ny)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, T && nx >= 2 && ny >= 1, "bad array");
    k_bc_neumann<<<(ny + 255) / 256, 256, 0, ctx->stream[0]>>>(T, nx, ny, ctx->cyc_skip);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// apply_boundary_conditions! (part2_utils.jl:22-31) in one launch: Dirichlet rows, then Neumann columns.  The corner points
// end up with the Dirichlet value of their row (their inner neighbour lies on that row), so every point has one writer.
__global__ __launch_bounds__(256) void k_bc2d(double* __restrict__ T, int nx, int ny, const int* __restrict__ skip)
{
    if (skip && *skip) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < nx) {
        T[t] = 1.0;
        T[(size_t)t + (size_t)nx * (ny - 1)] = 0.0;
    }
    if (t >= 1 && t < ny - 1) {
        T[(size_t)nx * t] = T[(size_t)nx * t + 1];
        T[(size_t)nx * t + nx - 1] = T[(size_t)nx * t + nx - 2];
    }
}

extern "C" int fpr_bc2d(fpr_ctx* ctx, double* T, int nx, int ny)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, T && nx >= 2 && ny >= 1, "bad array");
    if (nx < 3 || ny < 2) {   // degenerate shapes: the two reference operations one after the other
        if (int rc = fpr_bc_dirichlet2d(ctx, T, nx, ny)) return rc;
        return fpr_bc_neumann2d(ctx, T, nx, ny);
    }
    const int m = nx > ny ? nx : ny;
    k_bc2d<<<(m + 255) / 256, 256, 0, ctx->stream[0]>>>(T, nx, ny, ctx->cyc_skip);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// ---- CG ---------------------------------------------------------------------------------------------
struct CgWork { double *r, *p, *ph, *x, *p2; size_t n; };

static int cg_work(fpr_ctx* ctx, size_t n, CgWork* w)
{
    // four work vectors (krylov.jl:59-62) + the second p of the two-launch iteration, kept by the context
    if (ctx->cg_cap < 5 * n) {
        if (ctx->cg_buf) {
            FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
            FPR_HIP(ctx, hipFree(ctx->cg_buf));
            ctx->cg_buf = nullptr;
            ctx->cg_cap = 0;
        }
        FPR_HIP(ctx, hipMalloc(&ctx->cg_buf, 5 * n * sizeof(double)));
        ctx->cg_cap = 5 * n;
    }
    double* b = ctx->cg_buf;
    w->r = b; w->p = b + n; w->ph = b + 2 * n; w->x = b + 3 * n; w->p2 = b + 4 * n; w->n = n;
    return FPR_OK;
}

// runs cg! on the compute stream; leaves iters / last_rms in ctx->state_h (synchronises)
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
// Same operations per point as k_cg_pmv_f / k_cg_update_f; the dot products are summed per workgroup and then over the 16
// workgroups, i.e. in another order than the 64x4-tile partials of the other forms: results agree to rounding, not bit for
// bit (cg_fused = 3; forms 0-2 remain bit-identical among themselves).
constexpr int CGP_NB = 16, CGP_NBX = 4, CGP_NT = 1024, CGP_PPT = 5, CGP_RPT = 1;   // workgroups, threads, tile / ring points per thread
// (256 threads x 17 points: 9.0 us per iteration against 6.5 -- the two divisions per point of lap_at then weigh 2.8 us)
struct CgpArgs {
    const double* b;
    double* x_out;         // solution (whole array written)
    double* r_glob;        // N doubles: tile-edge values of r are exchanged through it
    double* part;          // 4 x CGP_NB slots (8 bytes every 128: partial sum = arrival flag), all CGP_EMPTY before the launch
    unsigned* ctr;         // [1] abort flag
    FprSolveState* st;
    int nx, ny, Nmax;
    double hx2, hy2, c, tol, N;
    long long* prof;       // diagnostic (option cg_prof = device address of 8 int64): ticks of workgroup 0 per section, summed
};

__device__ __forceinline__ double cgp_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cgp_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid barrier and all-reduce in ONE round trip: workgroup b stores its partial sum into its own slot of the set that belongs
// to this barrier; lanes 0..15 of wave 0 of every workgroup each watch one slot until it no longer holds the EMPTY pattern (a
// NaN payload no sum produces) -- the value itself is the arrival flag.  FOUR sets rotate: when a workgroup publishes for
// barrier g it first empties its slot of set (g+2) mod 4, last used at barrier g-2 (everybody has read that one: they all
// published g-1 since).  That slot is polled next at barrier g+2; between the emptying and that poll lie the owner's
// publications g and g+1, and the barriers alternate between RELEASE (publishes the workgroup's earlier stores -- the tile-edge
// values of r and the emptying -- acquired by the pollers) and relaxed (nothing but the sum travels), so one of the two orders
// the emptying before the poll.  Returns the 16 partials summed in workgroup order in every thread; *ok = false if the wait
// timed out (abort raised for everybody).
constexpr unsigned long long CGP_EMPTY = 0x7ff8dead0badf00dull;
constexpr int CGP_SLOT_STRIDE = 16;   // 8-byte words between two slots
// sum over the 64 lanes of a wave by DPP row shifts (an inclusive scan inside each 16-lane row, then the four row totals):
// ~25 instructions where the shuffle tree of fpr_wave_sum takes 12 LDS-crossbar round trips.  Result in every lane.
__device__ __forceinline__ double cgp_wave_sum(double v)
{
    v += mgs_dpp<0x111>(v);
    v += mgs_dpp<0x112>(v);
    v += mgs_dpp<0x114>(v);
    v += mgs_dpp<0x118>(v);   // lane 15 of every row: the row's total
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return ((r0 + r1) + r2) + r3;
}

template <bool RELEASE>
__device__ __forceinline__ double cgp_allsum(unsigned long long* slots, unsigned* abort_flag, double v_thread, unsigned& gen, double* red,
                                             double* gpart, int* s_abort, bool* ok)
{
    ++gen;
    // (a slot per 128-byte line: the pollers of different slots do not queue at one memory channel)
    unsigned long long* set = slots + (gen & 3u) * (CGP_NB * CGP_SLOT_STRIDE);
    unsigned long long* nxt = slots + ((gen + 2u) & 3u) * (CGP_NB * CGP_SLOT_STRIDE);
    const double v_wave = cgp_wave_sum(v_thread);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v_wave;
    __syncthreads();                       // wave totals in LDS; the workgroup's earlier stores precede the publication below
    if (threadIdx.x == 0) {
        double v_blk = red[0];
#pragma unroll
        for (int w = 1; w < CGP_NT / 64; ++w) v_blk += red[w];
        __hip_atomic_store(&nxt[blockIdx.x * CGP_SLOT_STRIDE], CGP_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long bits = (unsigned long long)__double_as_longlong(v_blk);
        if (bits == CGP_EMPTY) bits ^= 1ull;   // (a sum that happens to be this very NaN stays a NaN)
        if (RELEASE) __hip_atomic_store(&set[blockIdx.x * CGP_SLOT_STRIDE], bits, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(&set[blockIdx.x * CGP_SLOT_STRIDE], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < 64) {                // wave 0: lane w < 16 waits for workgroup w
        const int w = threadIdx.x;
        int ab = 0;
        if (w < CGP_NB) {
            unsigned spins = 0;
            unsigned long long bits;
            while (true) {
                bits = RELEASE ? __hip_atomic_load(&set[w * CGP_SLOT_STRIDE], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                               : __hip_atomic_load(&set[w * CGP_SLOT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bits != CGP_EMPTY) break;
                if ((++spins & 0x3ff) == 0) {
                    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
                    if (spins > (1u << 22)) {   // seconds, not minutes
                        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ab = 1;
                        break;
                    }
                }
            }
            gpart[w] = __longlong_as_double((long long)bits);
        }
        ab = __any(ab);
        if (threadIdx.x == 0) *s_abort = ab;
    }
    __syncthreads();
    *ok = *s_abort == 0;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < CGP_NB; ++w) s += gpart[w];
    return s;
}

__global__ void k_cgp_slots_init(unsigned long long* slots)
{
    if (threadIdx.x < 4 * CGP_NB) slots[threadIdx.x * CGP_SLOT_STRIDE] = CGP_EMPTY;
}

__global__ __launch_bounds__(CGP_NT) void k_cg_persistent(CgpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[CGP_NT / 64];
    __shared__ double gpart[CGP_NB];
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
        double acc0 = 0.0;
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k) acc0 += r[k] * r[k];
        rho = cgp_allsum<false>(slots, &a.ctr[1], acc0, gen, red, gpart, &s_abort, &ok);
        alive = ok;
        rr = rho;
    }
    const double thresh = a.tol * sqrt(rho);
    long long tsec[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool prof = a.prof != nullptr && blockIdx.x == 0 && tid == 0;
    auto lap = [&](int k) { if (prof) { const long long t = wall_clock64(); tsec[k] += t - tprev; tprev = t; } };
    if (prof) tprev = wall_clock64();
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
            for (int m = 0; m < CGP_RPT; ++m)   // acquired at barrier 2 by wave 0 (cache invalidate), workgroup barrier since
                if (rg[m] >= 0) p_ring[m] = a.r_glob[rg[m]] + beta * p_ring[m];
        }
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k)
            if (gi[k] >= 0) P[li[k]] = p[k];
#pragma unroll
        for (int m = 0; m < CGP_RPT; ++m)
            if (rl[m] >= 0) P[rl[m]] = p_ring[m];
        __syncthreads();
        lap(0);   // beta, new p, ring loads, LDS image
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k) {
            if (inter[k]) {
                const double t = p[k];
                const int l = li[k];
                q[k] = (((P[l + 1] - 2 * t) + P[l - 1]) / a.hx2 + ((P[l + lw] - 2 * t) + P[l - lw]) / a.hy2) - a.c * t;   // lap_at
            }
            if (gi[k] >= 0) acc += p[k] * q[k];
        }
        lap(1);   // operator
        const double pq = cgp_allsum<false>(slots, &a.ctr[1], acc, gen, red, gpart, &s_abort, &ok);   // barrier 1 of the iteration
        if (!ok) { alive = false; break; }
        lap(3);   // barrier 1
        // ---- alpha, x and r (krylov.jl:69-72), r.r; the tile-edge values of r go to the neighbours ----
        const double alpha = rho / pq;
        double acc2 = 0.0;
#pragma unroll
        for (int k = 0; k < CGP_PPT; ++k) {
            if (gi[k] >= 0) {
                x[k] = x[k] + alpha * p[k];
                const double rn = r[k] - alpha * q[k];
                r[k] = rn;
                acc2 += rn * rn;
                if (edge[k]) a.r_glob[gi[k]] = rn;   // plain store: published by the RELEASE of barrier 2 (an atomic store each would be
                                                     // issued behind an s_waitcnt of its own: five serial round trips)
            }
        }
        lap(4);   // update
        rr = cgp_allsum<true>(slots, &a.ctr[1], acc2, gen, red, gpart, &s_abort, &ok);        // barrier 2 (publishes the r edges)
        if (!ok) { alive = false; break; }
        lap(5);   // barrier 2
    }
    if (prof)
        for (int k = 0; k < 6; ++k) a.prof[k] += tsec[k];
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

static int cg_solve(fpr_ctx* ctx, double* x_in, const double* b, double hx, double hy, double c, double tol, int Nmax,
                    int nx, int ny)
{
    const size_t N = (size_t)nx * ny;
    CgWork w;
    if (int rc = cg_work(ctx, N, &w)) return rc;
    hipStream_t s = ctx->stream[0];
    const int fg = flat_grid(N);
    const dim3 g2 = grid2(nx, ny);
    const int np2 = (int)(g2.x * g2.y);
    if (np2 > FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
    k_cg_init<<<fg, 256, 0, s>>>(b, w.r, w.p, w.ph, w.x, N);
    FPR_CHECK_LAUNCH(ctx);
    if (int rc = fprx_dot_dev(ctx, b, b, N, ctx->scalars + 2)) return rc;
    k_state_init<<<1, 1, 0, s>>>(ctx->state, ctx->scalars + 2, tol, (double)N, 1);
    FPR_CHECK_LAUNCH(ctx);
    const int chunk = 64;
    // cg_fused: 3 (default where it applies) = one persistent launch for the whole solve, 2 = two dependent launches per
    // iteration, 1 = three, 0 = five (one per operation)
    long fused = fpr_opt(ctx, "cg_fused", 3);
    if (fused == 3) {
        const int twm = (nx + CGP_NBX - 1) / CGP_NBX, thm = (ny + CGP_NBX - 1) / CGP_NBX;
        const size_t lds = (size_t)(twm + 2) * (thm + 2) * sizeof(double);
        if (Nmax > 0 && (long)twm * thm <= (long)CGP_NT * CGP_PPT && 2L * (twm + 2) + 2L * thm <= (long)CGP_NT * CGP_RPT && lds <= 64 * 1024) {
            CgpArgs a;
            a.b = b; a.x_out = x_in; a.r_glob = w.r; a.part = ctx->partials;
            a.ctr = (unsigned*)(ctx->scalars + 40);   // two words of the scalar block: barrier counter, abort flag
            a.st = ctx->state;
            a.nx = nx; a.ny = ny; a.Nmax = Nmax;
            a.hx2 = hx * hx; a.hy2 = hy * hy; a.c = c; a.tol = tol; a.N = (double)N;
            a.prof = (long long*)(uintptr_t)fpr_opt(ctx, "cg_prof", 0);
            FPR_HIP(ctx, hipMemsetAsync(a.ctr, 0, 2 * sizeof(unsigned), s));
            k_cgp_slots_init<<<1, 64, 0, s>>>(reinterpret_cast<unsigned long long*>(a.part));
            // An ordinary launch: 16 workgroups of 1024 threads are resident together on any device this library runs on
            // (one per CU, 256 CUs), every wait is bounded, and a cooperative launch moves the process onto the runtime's
            // cooperative queue -- after it, kernels of two streams no longer overlap (measured: the side-by-side T / W solves
            // of the NS step 1.61 -> 1.96 ms, tools/exp_ns_only.py).
            k_cg_persistent<<<dim3(CGP_NB), dim3(CGP_NT), lds, s>>>(a);
            FPR_CHECK_LAUNCH(ctx);
            if (int rc = read_state(ctx)) return rc;
            if (ctx->state_h->done < 0) return fpr_fail(ctx, FPR_ERR_HIP, "cg!: a grid barrier of the persistent kernel timed out");
            return FPR_OK;   // x_in holds the solution (krylov.jl:88)
        }
        fused = 2;
    }
    const dim3 gcg((nx + CGX - 1) / CGX, (ny + CGY - 1) / CGY);
    const int npcg = (int)(gcg.x * gcg.y);
    if (npcg > FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
    int done_iters = 0;
    ctx->state_h->done = 0; ctx->state_h->iters = 0; ctx->state_h->last_rms = 0.0;
    while (done_iters < Nmax) {
        const int m = (Nmax - done_iters < chunk) ? Nmax - done_iters : chunk;
        for (int i = 0; i < m; ++i) {
            const int it = done_iters + i;
            if (fused == 2) {
                const double* pin = (it & 1) ? w.p2 : w.p;
                double* pout = (it & 1) ? w.p : w.p2;
                k_cg_pmv_f<<<gcg, blk2, 0, s>>>(pin, pout, w.r, w.ph, nx, ny, hx * hx, hy * hy, c, ctx->partials, ctx->partials2, fg,
                                               ctx->state, it, (double)N);
                k_cg_update_f<<<fg, 256, 0, s>>>(w.x, w.r, pout, w.ph, N, ctx->partials, npcg, ctx->partials2, ctx->state, it);
                continue;
            }
            k_cg_matvec_dot<<<g2, blk2, 0, s>>>(w.p, w.ph, nx, ny, hx * hx, hy * hy, c, ctx->partials, ctx->state);
            if (fused) {
                k_cg_update_f<<<fg, 256, 0, s>>>(w.x, w.r, w.p, w.ph, N, ctx->partials, np2, ctx->partials2, ctx->state, it);
                k_cg_p_f<<<fg, 256, 0, s>>>(w.p, w.r, N, ctx->partials2, fg, ctx->state, it, (double)N);
            } else {
                k_cg_alpha<<<1, 256, 0, s>>>(ctx->state, ctx->partials, np2);
                k_cg_update<<<fg, 256, 0, s>>>(w.x, w.r, w.p, w.ph, N, ctx->partials2, ctx->state);
                k_cg_check<<<1, 256, 0, s>>>(ctx->state, ctx->partials2, fg, (double)N);
                k_cg_p<<<fg, 256, 0, s>>>(w.p, w.r, N, ctx->state);
            }
        }
        done_iters += m;
        if (fused == 2) k_cg_tail_f<<<1, 256, 0, s>>>(ctx->partials2, fg, ctx->state, done_iters, (double)N);
        FPR_CHECK_LAUNCH(ctx);
        if (int rc = read_state(ctx)) return rc;
        if (ctx->state_h->done) break;
    }
    if (Nmax <= 0) {  // loop body never runs: r = b, x = 0 (krylov.jl:66,88,90)
        if (int rc = read_scalar(ctx, ctx->scalars + 2, &ctx->state_h->last_rms)) return rc;
        ctx->state_h->last_rms = sqrt(ctx->state_h->last_rms / (double)N);
    }
    FPR_HIP(ctx, hipMemcpyAsync(x_in, w.x, N * sizeof(double), hipMemcpyDeviceToDevice, s));  // krylov.jl:88
    return FPR_OK;
}

extern "C" int fpr_cg2d(fpr_ctx* ctx, double* x_in, const double* b, double hx, double hy, double c, double tol, int Nmax,
                        int nx, int ny, double* rms_host, int* iters_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, x_in && b, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    if (int rc = cg_solve(ctx, x_in, b, hx, hy, c, tol, Nmax, nx, ny)) return rc;
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
    if (rms_host) *rms_host = ctx->state_h->last_rms;
    if (iters_host) *iters_host = ctx->state_h->iters;
    return FPR_OK;
}

// ---- V-cycle ----------------------------------------------------------------------------------------
static int get_arena(fpr_ctx* ctx, int nx, int ny, int css, std::vector<FprLevel>** out)
{
    auto key = std::make_pair(nx, ny);
    auto it = ctx->arenas.find(key);
    if (it == ctx->arenas.end()) {
        std::vector<FprLevel> v;
        int lx = nx, ly = ny;
        // allocate down to 3x3-ish so any coarse_solve_size can reuse the arena
        while (true) {
            FprLevel L;
            L.nx = lx; L.ny = ly;
            FPR_HIP(ctx, hipMalloc(&L.tmp, (size_t)lx * ly * sizeof(double)));
            const bool can_coarsen = ((lx - 1) % 2 == 0) && ((ly - 1) % 2 == 0) && lx >= 5 && ly >= 5;
            if (can_coarsen) {
                const size_t nc = (size_t)(1 + (lx - 1) / 2) * (size_t)(1 + (ly - 1) / 2);
                FPR_HIP(ctx, hipMalloc(&L.res_c, nc * sizeof(double)));
                FPR_HIP(ctx, hipMalloc(&L.corr_c, nc * sizeof(double)));
            }
            v.push_back(L);
            if (!can_coarsen) break;
            lx = 1 + (lx - 1) / 2;
            ly = 1 + (ly - 1) / 2;
        }
        it = ctx->arenas.emplace(key, std::move(v)).first;
    }
    (void)css;
    *out = &it->second;
    return FPR_OK;
}

// ================================================================================================
// k_mid_down / k_mid_up: the three levels right above the LDS-resident sub-hierarchy in TWO launches
// ================================================================================================
// Below the finest levels a V-cycle is bound by launches, not by bytes: rocprofv3 shows ~4.5 us from dispatch to completion
// for a kernel that does nothing, and the marching passes of a 513^2, 257^2 or 129^2 level take 7-11 us each for 1-3 us
// of work.  The pre-smoothing passes of three consecutive levels A > B > C (each: two sweeps from the ZERO initial guess
// every level below the top starts from, multigrid.jl:132, + residual + injection) become ONE launch, and so do their three
// post-smoothing passes (prolongation + correction + two sweeps each): a workgroup owns a tile of the coarsest output and
// recomputes, in LDS, the halo it needs on the finer levels (pre: 53^2 points of A for 32^2 owned; post: 36^2 for 32^2),
// so workgroups never communicate.  Same point arithmetic and the same prolongation order as the per-level kernels: all
// arrays a later pass reads (tmp = the pre-smoothed field and res_c = the right-hand side of every level, the solution of
// level A) are bit-identical; the solutions of B and C only ever exist in LDS.
struct MidReg { int x0, x1, y0, y1; };   // inclusive index ranges at one level
__device__ __forceinline__ MidReg mid_clip(int x0, int x1, int y0, int y1, int nx, int ny)
{
    MidReg r;
    r.x0 = x0 < 0 ? 0 : x0; r.x1 = x1 > nx - 1 ? nx - 1 : x1;
    r.y0 = y0 < 0 ? 0 : y0; r.y1 = y1 > ny - 1 ? ny - 1 : y1;
    return r;
}
__device__ __forceinline__ MidReg mid_grow(MidReg r, int k, int nx, int ny) { return mid_clip(r.x0 - k, r.x1 + k, r.y0 - k, r.y1 + k, nx, ny); }
__device__ __forceinline__ int mid_w(const MidReg& r) { return r.x1 - r.x0 + 1; }
__device__ __forceinline__ int mid_h(const MidReg& r) { return r.y1 - r.y0 + 1; }
__device__ __forceinline__ int mid_n(const MidReg& r) { return mid_w(r) * mid_h(r); }
__device__ __forceinline__ int mid_at(const MidReg& r, int i, int j) { return (i - r.x0) + mid_w(r) * (j - r.y0); }

struct MidLevel {
    const double* f;     // right-hand side of the level (down: level A only is read, B and C are produced)
    double* tmp;         // pre-smoothed field (down writes, up reads)
    double* fout;        // down: right-hand side of the NEXT coarser level (res_c of this level)
    int nx, ny;
    double C, _h2, fac;
};
struct MidArgs {
    MidLevel L[3];       // A, B, C
    int nxD, nyD;        // the level below C (top of the LDS-resident sub-hierarchy)
    double* uD;          // down: its zero initial guess is written; up: its solution is read
    double* uA;          // up: solution of level A
    int apply_BCs;
    const int* skip;
};

constexpr int MID_TD = 4;     // k_mid_down: tile of the level-D right-hand side per workgroup
constexpr int MID_TA = 32;    // k_mid_up: tile of the level-A solution per workgroup
constexpr int MID_NT_DOWN = 1024, MID_NT_UP = 512;

// out(p) = in(p) + fac*res(in)(p) on region `ro`, boundary points copied; `in` lives on region `ri` (ro grown by one,
// clipped), f on region `rf`
__device__ __forceinline__ void mid_sweep(const double* in, const MidReg& ri, const double* f, const MidReg& rf, double* out,
                                          const MidReg& ro, int nx, int ny, double C, double _h2, double fac, int nt)
{
    const int w = mid_w(ro), n = mid_n(ro);
    const float rw = 1.0f / (float)w;
    const int wi = mid_w(ri);
    for (int idx = threadIdx.x; idx < n; idx += nt) {
        const int jj = mgs_row(idx, rw), ii = idx - jj * w;
        const int i = ro.x0 + ii, j = ro.y0 + jj;
        const int q = mid_at(ri, i, j);
        const double uc = in[q];
        double v = uc;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const double r = ((((in[q + 1] + in[q - 1]) + in[q + wi]) + in[q - wi]) - C * uc) * _h2 - f[mid_at(rf, i, j)];
            v = uc + fac * r;
        }
        out[idx] = v;
    }
}

// prolong_bf with the coarse field in an LDS region (indices clamped into it; a clamped value is never used)
__device__ __forceinline__ double mid_prolong(const double* cc, const MidReg& rc, int i, int j, int nx, int ny, int nxc, int nyc)
{
    const bool in = i >= 1 && j >= 1 && i <= nx - 2 && j <= ny - 2;
    const int io = i & 1, jo = j & 1;
    int icl = i >> 1, jcl = j >> 1;
    int ich = (icl + 1 < nxc) ? icl + 1 : nxc - 1, jch = (jcl + 1 < nyc) ? jcl + 1 : nyc - 1;
    const double w = (io | jo) ? ((io & jo) ? 0.25 : 0.5) : 1.0;
    const bool sx0 = icl >= 1 && icl <= nxc - 2, sx1 = io && (icl + 1 <= nxc - 2);
    const bool sy0 = jcl >= 1 && jcl <= nyc - 2, sy1 = jo && (jcl + 1 <= nyc - 2);
    icl = icl < rc.x0 ? rc.x0 : (icl > rc.x1 ? rc.x1 : icl); ich = ich < rc.x0 ? rc.x0 : (ich > rc.x1 ? rc.x1 : ich);
    jcl = jcl < rc.y0 ? rc.y0 : (jcl > rc.y1 ? rc.y1 : jcl); jch = jch < rc.y0 ? rc.y0 : (jch > rc.y1 ? rc.y1 : jch);
    const double c00 = cc[mid_at(rc, icl, jcl)], c10 = cc[mid_at(rc, ich, jcl)];
    const double c01 = cc[mid_at(rc, icl, jch)], c11 = cc[mid_at(rc, ich, jch)];
    double v = 0.0;
    v = v + ((in && sx0 && sy0) ? w * c00 : 0.0);
    v = v + ((in && sx1 && sy0) ? w * c10 : 0.0);
    v = v + ((in && sx0 && sy1) ? w * c01 : 0.0);
    v = v + ((in && sx1 && sy1) ? w * c11 : 0.0);
    return v;
}

__global__ __launch_bounds__(MID_NT_DOWN) void k_mid_down(MidArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (a.skip && *a.skip) return;
    constexpr int NT = MID_NT_DOWN;
    const int tid = threadIdx.x;
    const int ntx = (a.nxD - 1) / MID_TD > 0 ? (a.nxD - 1) / MID_TD : 1, nty = (a.nyD - 1) / MID_TD > 0 ? (a.nyD - 1) / MID_TD : 1;
    const int bx = blockIdx.x, by = blockIdx.y;
    // owned tile of the level-D right-hand side (the last tile of a dimension takes the remainder)
    MidReg own;
    own.x0 = bx * MID_TD; own.x1 = (bx == ntx - 1) ? a.nxD - 1 : own.x0 + MID_TD - 1;
    own.y0 = by * MID_TD; own.y1 = (by == nty - 1) ? a.nyD - 1 : own.y0 + MID_TD - 1;
    // regions, coarse to fine: rt[l] = where the pre-smoothed field of level l is needed, rf[l] = rt[l] grown by one
    // (first sweep / right-hand side); the residual at coarse point c reads the fine field at 2c-1 .. 2c+1
    MidReg rt[3], rf[3], rn = own;   // rn = region of the next coarser right-hand side
    for (int l = 2; l >= 0; --l) {
        rt[l] = mid_clip(2 * rn.x0 - 1, 2 * rn.x1 + 1, 2 * rn.y0 - 1, 2 * rn.y1 + 1, a.L[l].nx, a.L[l].ny);
        rf[l] = mid_grow(rt[l], 1, a.L[l].nx, a.L[l].ny);
        rn = rf[l];
    }
    // LDS: F | U1 (on rf) | U2 (on rt) of the current level, then the next level's F behind them
    double* F = sm;
    {   // right-hand side of level A from memory
        const MidLevel& L = a.L[0];
        const int w = mid_w(rf[0]), n = mid_n(rf[0]);
        const float rw = 1.0f / (float)w;
        for (int idx = tid; idx < n; idx += NT) {
            const int jj = mgs_row(idx, rw), ii = idx - jj * w;
            F[idx] = L.f[(size_t)(rf[0].x0 + ii) + (size_t)L.nx * (rf[0].y0 + jj)];
        }
    }
    __syncthreads();
    int scale = 8;   // level-l index = scale * level-D index
    for (int l = 0; l < 3; ++l, scale >>= 1) {
        const MidLevel& L = a.L[l];
        const int nf = mid_n(rf[l]), ntm = mid_n(rt[l]);
        double* U1 = F + nf;
        double* U2 = U1 + nf;
        double* Fn = U2 + ntm;   // next level's right-hand side
        {   // first sweep from the zero initial guess (:124 with u = 0; the literal arithmetic on zeros, kept bit for bit)
            const int w = mid_w(rf[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < nf; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = rf[l].x0 + ii, j = rf[l].y0 + jj;
                double v = 0.0;
                if (i >= 1 && j >= 1 && i < L.nx - 1 && j < L.ny - 1) {
                    const double r = ((((0.0 + 0.0) + 0.0) + 0.0) - L.C * 0.0) * L._h2 - F[idx];
                    v = 0.0 + L.fac * r;
                }
                U1[idx] = v;
            }
        }
        __syncthreads();
        mid_sweep(U1, rf[l], F, rf[l], U2, rt[l], L.nx, L.ny, L.C, L._h2, L.fac, NT);   // :125
        __syncthreads();
        {   // the owned part of the pre-smoothed field goes to memory (the post-smoothing pass reads it)
            MidReg o;
            o.x0 = scale * own.x0; o.x1 = (bx == ntx - 1) ? L.nx - 1 : scale * (own.x1 + 1) - 1;
            o.y0 = scale * own.y0; o.y1 = (by == nty - 1) ? L.ny - 1 : scale * (own.y1 + 1) - 1;
            const int w = mid_w(o), n = mid_n(o);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = o.x0 + ii, j = o.y0 + jj;
                L.tmp[(size_t)i + (size_t)L.nx * j] = U2[mid_at(rt[l], i, j)];
            }
        }
        {   // residual at the injected points = right-hand side of the next level (:128-131; Neumann columns :355-357)
            const MidReg rc = (l < 2) ? rf[l + 1] : own;
            const int nxc = 1 + (L.nx - 1) / 2, nyc = 1 + (L.ny - 1) / 2;
            const int half = scale >> 1;
            MidReg o;   // owned part of that right-hand side
            o.x0 = half * own.x0; o.x1 = (bx == ntx - 1) ? nxc - 1 : half * (own.x1 + 1) - 1;
            o.y0 = half * own.y0; o.y1 = (by == nty - 1) ? nyc - 1 : half * (own.y1 + 1) - 1;
            const int w = mid_w(rc), n = mid_n(rc), wt = mid_w(rt[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int ic = rc.x0 + ii, jc = rc.y0 + jj;
                int is = ic;
                if (a.apply_BCs) is = (ic == 0) ? 1 : (ic == nxc - 1 ? nxc - 2 : ic);
                double v = 0.0;
                if (is >= 1 && is <= nxc - 2 && jc >= 1 && jc <= nyc - 2) {
                    const int q = mid_at(rt[l], 2 * is, 2 * jc);
                    v = ((((U2[q + 1] + U2[q - 1]) + U2[q + wt]) + U2[q - wt]) - L.C * U2[q]) * L._h2 - F[mid_at(rf[l], 2 * is, 2 * jc)];
                }
                if (l < 2) Fn[idx] = v;
                if (ic >= o.x0 && ic <= o.x1 && jc >= o.y0 && jc <= o.y1) {
                    L.fout[(size_t)ic + (size_t)nxc * jc] = v;
                    if (l == 2) a.uD[(size_t)ic + (size_t)nxc * jc] = 0.0;   // zero initial guess of level D (:132)
                }
            }
        }
        __syncthreads();
        F = Fn;
    }
}

__global__ __launch_bounds__(MID_NT_UP) void k_mid_up(MidArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (a.skip && *a.skip) return;
    constexpr int NT = MID_NT_UP;
    const int tid = threadIdx.x;
    const MidLevel& LA = a.L[0];
    const int ntx = (LA.nx - 1) / MID_TA > 0 ? (LA.nx - 1) / MID_TA : 1, nty = (LA.ny - 1) / MID_TA > 0 ? (LA.ny - 1) / MID_TA : 1;
    const int bx = blockIdx.x, by = blockIdx.y;
    MidReg own;   // owned tile of the level-A solution
    own.x0 = bx * MID_TA; own.x1 = (bx == ntx - 1) ? LA.nx - 1 : own.x0 + MID_TA - 1;
    own.y0 = by * MID_TA; own.y1 = (by == nty - 1) ? LA.ny - 1 : own.y0 + MID_TA - 1;
    // regions, fine to coarse: r2[l] = where the solution of level l is needed, r1 = r2 grown by one (after the first
    // sweep), rc = r2 grown by two (the corrected pre-smoothed field); the prolongation onto rc reads coarse points i>>1, (i>>1)+1
    MidReg r2[3], r1[3], rc[3], rD;
    r2[0] = own;
    for (int l = 0; l < 3; ++l) {
        r1[l] = mid_grow(r2[l], 1, a.L[l].nx, a.L[l].ny);
        rc[l] = mid_grow(r2[l], 2, a.L[l].nx, a.L[l].ny);
        const int nxc = 1 + (a.L[l].nx - 1) / 2, nyc = 1 + (a.L[l].ny - 1) / 2;
        const MidReg rn = mid_clip(rc[l].x0 >> 1, (rc[l].x1 + 1) >> 1, rc[l].y0 >> 1, (rc[l].y1 + 1) >> 1, nxc, nyc);
        if (l < 2) r2[l + 1] = rn; else rD = rn;
    }
    // LDS layout: per level X (on rc) | F (on r1) | U1 (on r1) | U2 (on r2; level A writes to memory instead), then D's field
    double* X[3]; double* Fv[3]; double* U1[3]; double* U2[3];
    double* p = sm;
    for (int l = 0; l < 3; ++l) {
        X[l] = p; p += mid_n(rc[l]);
        Fv[l] = p; p += mid_n(r1[l]);
        U1[l] = p; p += mid_n(r1[l]);
        U2[l] = p; p += (l == 0) ? 0 : mid_n(r2[l]);
    }
    double* UD = p;
    // every load the launch needs is issued up front: the pre-smoothed fields, the right-hand sides, the solution of level D
    for (int l = 0; l < 3; ++l) {
        const MidLevel& L = a.L[l];
        {
            const int w = mid_w(rc[l]), n = mid_n(rc[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                X[l][idx] = L.tmp[(size_t)(rc[l].x0 + ii) + (size_t)L.nx * (rc[l].y0 + jj)];
            }
        }
        {
            const int w = mid_w(r1[l]), n = mid_n(r1[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                Fv[l][idx] = L.f[(size_t)(r1[l].x0 + ii) + (size_t)L.nx * (r1[l].y0 + jj)];
            }
        }
    }
    {
        const int w = mid_w(rD), n = mid_n(rD);
        const float rw = 1.0f / (float)w;
        for (int idx = tid; idx < n; idx += NT) {
            const int jj = mgs_row(idx, rw), ii = idx - jj * w;
            UD[idx] = a.uD[(size_t)(rD.x0 + ii) + (size_t)a.nxD * (rD.y0 + jj)];
        }
    }
    __syncthreads();
    const double* cc = UD;
    MidReg rcc = rD;
    for (int l = 2; l >= 0; --l) {
        const MidLevel& L = a.L[l];
        const int nxc = 1 + (L.nx - 1) / 2, nyc = 1 + (L.ny - 1) / 2;
        {   // prolongation + correction (:136-139) of the pre-smoothed field, in place
            const int w = mid_w(rc[l]), n = mid_n(rc[l]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = rc[l].x0 + ii, j = rc[l].y0 + jj;
                int is = i;
                if (a.apply_BCs) is = (i == 0) ? 1 : (i == L.nx - 1 ? L.nx - 2 : i);
                X[l][idx] = X[l][idx] - mid_prolong(cc, rcc, is, j, L.nx, L.ny, nxc, nyc);
            }
        }
        __syncthreads();
        mid_sweep(X[l], rc[l], Fv[l], r1[l], U1[l], r1[l], L.nx, L.ny, L.C, L._h2, L.fac, NT);   // :142
        __syncthreads();
        if (l > 0) {
            mid_sweep(U1[l], r1[l], Fv[l], r1[l], U2[l], r2[l], L.nx, L.ny, L.C, L._h2, L.fac, NT);   // :143
            __syncthreads();
            cc = U2[l];
            rcc = r2[l];
        } else {   // level A: the second sweep writes the owned tile of the solution to memory
            const int w = mid_w(own), n = mid_n(own), wi = mid_w(r1[0]);
            const float rw = 1.0f / (float)w;
            for (int idx = tid; idx < n; idx += NT) {
                const int jj = mgs_row(idx, rw), ii = idx - jj * w;
                const int i = own.x0 + ii, j = own.y0 + jj;
                const int q = mid_at(r1[0], i, j);
                const double uc = U1[0][q];
                double v = uc;
                if (i >= 1 && j >= 1 && i < L.nx - 1 && j < L.ny - 1) {
                    const double r = ((((U1[0][q + 1] + U1[0][q - 1]) + U1[0][q + wi]) + U1[0][q - wi]) - L.C * uc) * L._h2 - Fv[0][q];
                    v = uc + L.fac * r;
                }
                a.uA[(size_t)i + (size_t)L.nx * j] = v;
            }
        }
    }
}

// Does the sub-hierarchy below an (nx, ny) level fit k_mg_small's LDS arena?  nlev = its levels, tot = doubles needed.
static bool mgs_plan(int nx, int ny, int css, int* nlev_out, size_t* tot_out)
{
    int nlev = 1, lx = nx, ly = ny;
    size_t tot = 3 * (size_t)lx * ly;
    while ((lx < ly ? lx : ly) > css) {
        if ((lx - 1) % 2 || (ly - 1) % 2) return false;
        const int m = (lx < ly ? lx : ly) - 1;
        if (m & (m - 1)) return false;
        lx = 1 + (lx - 1) / 2;
        ly = 1 + (ly - 1) / 2;
        tot += 3 * (size_t)lx * ly;
        ++nlev;
        if (nlev > 16) return false;
    }
    *nlev_out = nlev;
    *tot_out = tot;
    return tot + MGS_RED <= 20000;
}

// True if every launch of a V-cycle on this hierarchy honours FprCycleCtl::stop (the marching passes with fused
// restriction / prolongation down to an LDS-resident sub-hierarchy, Jacobi coarse solver): fpr_mgsolve2d may then
// enqueue cycles ahead of the host's convergence check.  Anything else (CG, the per-operation kernels, levels the
// march does not take, shapes that raise the reference's errors) runs the plain loop.
static bool vcycle_streams(fpr_ctx* ctx, int nx, int ny, int css, int solver)
{
    if (solver != FPR_COARSE_JACOBI || !fpr_opt(ctx, "mg_small", 1) || fpr_opt(ctx, "mg_multi", 1) != 1) return false;
    if (!fpr_opt(ctx, "mg_fuse_restrict", 1) || !fpr_opt(ctx, "mg_fuse_prolong", 1)) return false;
    int lx = nx, ly = ny;
    for (int depth = 0; depth < 32; ++depth) {
        if ((lx - 1) % 2 || (ly - 1) % 2) return false;
        const int m = (lx < ly ? lx : ly) - 1;
        if (m <= 0 || (m & (m - 1))) return false;
        int nlev;
        size_t tot;
        if (mgs_plan(lx, ly, css, &nlev, &tot)) return depth > 0;
        if (!((lx < ly ? lx : ly) > css && lx >= 64 && ly >= 16)) return false;
        lx = 1 + (lx - 1) / 2;
        ly = 1 + (ly - 1) / 2;
    }
    return false;
}

// One level of Vcycle_2DPoisson! (multigrid.jl:91-170).  want_norm: top level only -- the r_rms of the
// last post-smoothing sweep is left in ctx->scalars[0] (as sum of squares) for the caller.
static int vcycle_level(fpr_ctx* ctx, std::vector<FprLevel>& A, size_t d, double* u, const double* rhs, double h, double c,
                        double tol, int css, int solver, int apply_BCs, bool top, double* rms_out_host, bool* rms_is_host)
{
    FprLevel& L = A[d];
    const int nx = L.nx, ny = L.ny;
    hipStream_t s = ctx->stream[0];
    const int* skp = ctx->cyc_skip;
    if ((nx - 1) != 2 * ((nx - 1) / 2) || (ny - 1) != 2 * ((ny - 1) / 2))
        return fpr_fail(ctx, FPR_ERR_NOT_POW2, "ERROR:not a power of 2 (nx=%d, ny=%d)", nx, ny);  // multigrid.jl:95-97
    {
        const int m = (nx < ny ? nx : ny) - 1;
        if (m <= 0 || (m & (m - 1)) != 0)
            return fpr_fail(ctx, FPR_ERR_NOT_POW2, "min(nx,ny)-1 = %d is not a power of 2 (multigrid.jl:103)", m);
    }
    const size_t N = (size_t)nx * ny;
    const double C = 4.0 + c * (h * h), _h2 = 1 / (h * h);
    const double fac = (4.0 / 5.0) * ((h * h) / (4.0 + c * (h * h)));
    const dim3 g = grid2(nx, ny);
    const int np = (int)(g.x * g.y);
    if (np > FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");

    // ---- LDS-resident sub-hierarchy: this level and everything below it in one workgroup ----
    if (solver == FPR_COARSE_JACOBI && fpr_opt(ctx, "mg_small", 1)) {
        int nlev = 1;
        size_t tot = 0;
        const bool ok = mgs_plan(nx, ny, css, &nlev, &tot);
        if (ok) {
            static bool attr_set = false;
            if (!attr_set) {
                FPR_HIP(ctx, hipFuncSetAttribute((const void*)k_mg_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set = true;
            }
            MgSmallArgs a;
            a.u = u; a.rhs = rhs; a.nx = nx; a.ny = ny; a.nlev = nlev;
            a.h = h; a.c = c; a.tol = tol; a.css = css; a.apply_BCs = apply_BCs;
            a.want_norm = (top && nlev > 1) ? 1 : 0;
            a.out_sumsq = ctx->scalars;
            a.state = ctx->state;
            a.skip = ctx->cyc_skip;
            a.row_solve = fpr_opt(ctx, "mg_small_row", 1) != 0;
            a.prof = (long long*)(uintptr_t)fpr_opt(ctx, "mg_small_prof", 0);   // tools/exp_mg_small_prof.py: device address of 32 int64, or 0
            k_mg_small<<<1, MGS_NT, (tot + MGS_RED) * sizeof(double), s>>>(a);
            FPR_CHECK_LAUNCH(ctx);
            if (top) {
                *rms_is_host = false;
                ctx->top_is_coarsest = (nlev == 1);
            }
            ctx->used_small = true;
            return FPR_OK;
        }
    }

    // ---- three launch-bound levels in two launches (k_mid_down / k_mid_up) around the LDS-resident sub-hierarchy ----
    if (!top && solver == FPR_COARSE_JACOBI && fpr_opt(ctx, "mg_mid", 1) && fpr_opt(ctx, "mg_small", 1) &&
        fpr_opt(ctx, "mg_multi", 1) == 1 && fpr_opt(ctx, "mg_fuse_restrict", 1) && fpr_opt(ctx, "mg_fuse_prolong", 1) &&
        fpr_opt(ctx, "mg_vx", 1) != 2 && d + 3 < A.size()) {
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {   // levels d .. d+2 are levels the march would take, and they coarsen
            const FprLevel& Lk = A[d + k];
            const int m = (Lk.nx < Lk.ny ? Lk.nx : Lk.ny) - 1;
            ok = Lk.res_c && Lk.corr_c && Lk.tmp && (Lk.nx - 1) % 2 == 0 && (Lk.ny - 1) % 2 == 0 && m > 0 && (m & (m - 1)) == 0 &&
                 (Lk.nx < Lk.ny ? Lk.nx : Lk.ny) > css && Lk.nx >= 64 && Lk.ny >= 16 && Lk.nx <= 1025 && Lk.ny <= 1025;
        }
        int nlevD = 0;
        size_t totD = 0;
        ok = ok && mgs_plan(A[d + 3].nx, A[d + 3].ny, css, &nlevD, &totD);
        size_t lds_down = 0, lds_up = 0;
        if (ok) {
            const int nxD = A[d + 3].nx, nyD = A[d + 3].ny;
            auto own_max = [](int n, int t) { return (n - 1) / t > 0 ? t + 1 + (n - 1) % t : n; };   // widest tile (the last one)
            {   // k_mid_down: F | U1 on rf, U2 on rt per level, the level-D tile
                long wx = own_max(nxD, MID_TD), wy = own_max(nyD, MID_TD);
                lds_down = (size_t)(wx * wy);
                for (int k = 2; k >= 0; --k) {
                    long tx = 2 * wx + 1, ty = 2 * wy + 1;
                    long fx = tx + 2, fy = ty + 2;
                    if (tx > A[d + k].nx) tx = A[d + k].nx;
                    if (ty > A[d + k].ny) ty = A[d + k].ny;
                    if (fx > A[d + k].nx) fx = A[d + k].nx;
                    if (fy > A[d + k].ny) fy = A[d + k].ny;
                    lds_down += (size_t)(2 * fx * fy + tx * ty);
                    wx = fx; wy = fy;
                }
            }
            {   // k_mid_up: X on rc, F and U1 on r1, U2 on r2 per level, the level-D field
                long wx = own_max(A[d].nx, MID_TA), wy = own_max(A[d].ny, MID_TA);
                for (int k = 0; k < 3; ++k) {
                    if (wx > A[d + k].nx) wx = A[d + k].nx;
                    if (wy > A[d + k].ny) wy = A[d + k].ny;
                    lds_up += (size_t)((wx + 4) * (wy + 4) + 2 * (wx + 2) * (wy + 2) + wx * wy);
                    wx = (wx + 4 + 1) / 2 + 1; wy = (wy + 4 + 1) / 2 + 1;
                }
                lds_up += (size_t)(wx * wy);
            }
            ok = lds_down <= 20000 && lds_up <= 20000;
        }
        if (ok) {
            static bool attr_set = false;
            if (!attr_set) {
                FPR_HIP(ctx, hipFuncSetAttribute((const void*)k_mid_down, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                FPR_HIP(ctx, hipFuncSetAttribute((const void*)k_mid_up, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set = true;
            }
            MidArgs a;
            double hk = h;
            for (int k = 0; k < 3; ++k) {
                FprLevel& Lk = A[d + k];
                a.L[k].f = (k == 0) ? rhs : A[d + k - 1].res_c;
                a.L[k].tmp = Lk.tmp;
                a.L[k].fout = Lk.res_c;
                a.L[k].nx = Lk.nx; a.L[k].ny = Lk.ny;
                a.L[k].C = 4.0 + c * (hk * hk);
                a.L[k]._h2 = 1 / (hk * hk);
                a.L[k].fac = (4.0 / 5.0) * ((hk * hk) / (4.0 + c * (hk * hk)));
                hk = hk * 2;   // the recursion passes h*2 (multigrid.jl:133)
            }
            a.nxD = A[d + 3].nx; a.nyD = A[d + 3].ny;
            a.uD = A[d + 2].corr_c;
            a.uA = u;
            a.apply_BCs = apply_BCs;
            a.skip = skp;
            const dim3 gd((a.nxD - 1) / MID_TD > 0 ? (a.nxD - 1) / MID_TD : 1, (a.nyD - 1) / MID_TD > 0 ? (a.nyD - 1) / MID_TD : 1);
            const dim3 gu((nx - 1) / MID_TA > 0 ? (nx - 1) / MID_TA : 1, (ny - 1) / MID_TA > 0 ? (ny - 1) / MID_TA : 1);
            k_mid_down<<<gd, MID_NT_DOWN, lds_down * sizeof(double), s>>>(a);   // :124-132 of levels d, d+1, d+2
            FPR_CHECK_LAUNCH(ctx);
            double dummy; bool dh;
            if (int rc = vcycle_level(ctx, A, d + 3, A[d + 2].corr_c, A[d + 2].res_c, hk, c, tol, css, solver, apply_BCs, false, &dummy, &dh))
                return rc;  // :133 (k_mg_small)
            k_mid_up<<<gu, MID_NT_UP, lds_up * sizeof(double), s>>>(a);       // :136-143 of levels d+2, d+1, d
            FPR_CHECK_LAUNCH(ctx);
            return FPR_OK;
        }
    }

    if ((nx < ny ? nx : ny) > css) {  // multigrid.jl:121
        if (d + 1 >= A.size() || !L.res_c) return fpr_fail(ctx, FPR_ERR_INVALID, "level arena exhausted");
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        if (fpr_opt(ctx, "mg_multi", 1) == 1 && nx >= 64 && ny >= 16) {
            // temporal blocking, register-rolling march: each smoothing pair is ONE pass (u -> tmp, tmp -> u)
            const bool fuse_r = fpr_opt(ctx, "mg_fuse_restrict", 1) != 0;
            const int ntf = fpr_opt(ctx, "mg_nt", 0) ? 256 : 0;
            // two columns per lane (128-column strips, aligned 16-byte accesses with a per-row lane shift): bit-identical but
            // measured SLOWER on MI355X (4.4 vs 2.8 ms per 4097^2 solve; 114-204 VGPRs against 36-60), so it is opt-in (mg_vx = 2)
            const bool al16 = ((((uintptr_t)u | (uintptr_t)rhs | (uintptr_t)L.tmp) & 15) == 0);
            const bool vx2 = fpr_opt(ctx, "mg_vx", 1) == 2 && nx >= 128 && al16;
            const int sw = vx2 ? 122 : 60, sw_r = vx2 ? 120 : 58;
            const int nstrips = (nx + sw - 1) / sw;
            const int nstrips_r = (nx + sw_r - 1) / sw_r;  // strips of the restricting pre-smoothing pass
            int rpc = (int)fpr_opt(ctx, "mg_rows_per_chunk", 0);
            if (rpc <= 0) {  // enough chunks for >= ~16 waves per CU, chunks of at least 16 rows
                const long target = fpr_opt(ctx, "mg_wave_target", 4096);
                rpc = 64;
                while (rpc > 16 && (long)nstrips * ((ny + rpc - 1) / rpc) < target) rpc >>= 1;
            }
            const dim3 gm((nstrips + 3) / 4, (ny + rpc - 1) / rpc);
            const int npm = (int)(gm.x * gm.y);
            const bool fuse_p = fpr_opt(ctx, "mg_fuse_prolong", 1) != 0;
            if (fuse_r) {  // pre-smoothing pair + residual + injection + zero coarse guess in ONE pass (:124-132)
                const dim3 gr((nstrips_r + 3) / 4, (ny + rpc - 1) / rpc);
                const bool timed = top && fpr_ktimer_begin(ctx, FPR_KT_MG_PRE, s);
                { if (vx2) k_smooth2_march2<false, false, true><<<gr, 256, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, rpc, nstrips_r, nullptr, nullptr, ntf, L.res_c, L.corr_c, skp); else k_smooth2_march<false, false, true><<<gr, 256, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, rpc, nstrips_r, nullptr, nullptr, apply_BCs | ntf, L.res_c, L.corr_c, skp); }
                fpr_ktimer_end(ctx, timed, s);
                if (apply_BCs && vx2) k_bc_neumann<<<(nyc + 255) / 256, 256, 0, s>>>(L.res_c, nxc, nyc, skp);  // :355-357 (k_smooth2_march does it itself)
            } else {
                { if (vx2) k_smooth2_march2<false, false, false><<<gm, 256, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, nullptr, ntf, nullptr, nullptr, skp); else k_smooth2_march<false, false, false><<<gm, 256, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, nullptr, ntf, nullptr, nullptr, skp); }  // :124-125
                k_restrict_residual2d<<<grid2(nxc, nyc), blk2, 0, s>>>(L.tmp, rhs, L.res_c, nx, ny, C, _h2, apply_BCs, L.corr_c);  // :128-132
            }
            FPR_CHECK_LAUNCH(ctx);
            double dummy; bool dh;
            if (int rc = vcycle_level(ctx, A, d + 1, L.corr_c, L.res_c, h * 2, c, tol, css, solver, apply_BCs, false, &dummy, &dh))
                return rc;  // :133
            if (!fuse_p) k_prolong2d<true><<<g, blk2, 0, s>>>(L.corr_c, L.tmp, nx, ny, apply_BCs);  // :136-139
            // post-smoothing pair (:142-143); with fuse_p the correction u - P(corr_c) is applied while loading
            if (top) {
                const bool timed = fpr_ktimer_begin(ctx, FPR_KT_MG_POST, s);
                if (fuse_p) { if (vx2) k_smooth2_march2<true, true, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, ctx->partials, L.corr_c, apply_BCs | ntf, nullptr, nullptr, skp); else k_smooth2_march<true, true, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, ctx->partials, L.corr_c, apply_BCs | ntf, nullptr, nullptr, skp); }
                else { if (vx2) k_smooth2_march2<true, false, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, ctx->partials, nullptr, ntf, nullptr, nullptr, skp); else k_smooth2_march<true, false, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, ctx->partials, nullptr, ntf, nullptr, nullptr, skp); }
                fpr_ktimer_end(ctx, timed, s);
                FPR_CHECK_LAUNCH(ctx);
                if (skp) { if (int rc = fprx_cycle_finish(ctx, ctx->partials, npm, ctx->scalars, (double)nx * (double)ny, ctx->cyc_slot)) return rc; }
                else if (int rc = fprx_finish_sum(ctx, ctx->partials, npm, ctx->scalars, false, 0)) return rc;
                *rms_is_host = false;
            } else {
                if (fuse_p) { if (vx2) k_smooth2_march2<false, true, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, L.corr_c, apply_BCs | ntf, nullptr, nullptr, skp); else k_smooth2_march<false, true, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, L.corr_c, apply_BCs | ntf, nullptr, nullptr, skp); }
                else { if (vx2) k_smooth2_march2<false, false, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, nullptr, ntf, nullptr, nullptr, skp); else k_smooth2_march<false, false, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, nullptr, ntf, nullptr, nullptr, skp); }
                FPR_CHECK_LAUNCH(ctx);
            }
            return FPR_OK;
        }
        if (fpr_opt(ctx, "mg_multi", 1) == 2 && nx >= 64 && ny >= 32) {
            // temporal blocking, LDS tile variant (kept for A/B comparison; slower than the march on gfx950)
            constexpr int TX = 64, TY = 32;
            const dim3 gm((nx + TX - 1) / TX, (ny + TY - 1) / TY);
            const int npm = (int)(gm.x * gm.y);
            k_sweep2d_multi<2, TX, TY, 0, false><<<gm, 256, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, 2, nullptr, nullptr);  // :124-125
            k_restrict_residual2d<<<grid2(nxc, nyc), blk2, 0, s>>>(L.tmp, rhs, L.res_c, nx, ny, C, _h2, apply_BCs, L.corr_c);  // :128-132
            FPR_CHECK_LAUNCH(ctx);
            double dummy; bool dh;
            if (int rc = vcycle_level(ctx, A, d + 1, L.corr_c, L.res_c, h * 2, c, tol, css, solver, apply_BCs, false, &dummy, &dh))
                return rc;  // :133
            k_prolong2d<true><<<g, blk2, 0, s>>>(L.corr_c, L.tmp, nx, ny, apply_BCs);  // :136-139
            if (top) {
                k_sweep2d_multi<2, TX, TY, 1, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, 2, ctx->partials, nullptr);  // :142-143
                FPR_CHECK_LAUNCH(ctx);
                if (int rc = fprx_finish_sum(ctx, ctx->partials, npm, ctx->scalars, false, 0)) return rc;
                *rms_is_host = false;
            } else {
                k_sweep2d_multi<2, TX, TY, 0, false><<<gm, 256, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, 2, nullptr, nullptr);
                FPR_CHECK_LAUNCH(ctx);
            }
            return FPR_OK;
        }
        // two pre-smoothing sweeps (:124-125)
        k_sweep2d<false, false><<<g, blk2, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, nullptr, nullptr);
        k_sweep2d<false, false><<<g, blk2, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, nullptr, nullptr);
        // residual + restriction (:128-129), coarse correction starts from zero (:132)
        k_restrict_residual2d<<<grid2(nxc, nyc), blk2, 0, s>>>(u, rhs, L.res_c, nx, ny, C, _h2, apply_BCs, L.corr_c);
        FPR_CHECK_LAUNCH(ctx);
        double dummy; bool dh;
        if (int rc = vcycle_level(ctx, A, d + 1, L.corr_c, L.res_c, h * 2, c, tol, css, solver, apply_BCs, false, &dummy, &dh))
            return rc;  // :133
        // prolongation + correction (:136-139)
        k_prolong2d<true><<<g, blk2, 0, s>>>(L.corr_c, u, nx, ny, apply_BCs);
        // two post-smoothing sweeps (:142-143); only the very last one of the top level needs its norm
        k_sweep2d<false, false><<<g, blk2, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, nullptr, nullptr);
        if (top) {
            k_sweep2d<true, false><<<g, blk2, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, ctx->partials, nullptr);
            FPR_CHECK_LAUNCH(ctx);
            if (int rc = fprx_finish_sum(ctx, ctx->partials, np, ctx->scalars, false, 0)) return rc;
            *rms_is_host = false;
        } else {
            k_sweep2d<false, false><<<g, blk2, 0, s>>>(L.tmp, rhs, u, nx, ny, C, _h2, fac, nullptr, nullptr);
            FPR_CHECK_LAUNCH(ctx);
        }
        return FPR_OK;
    }

    // ---- coarsest level ----
    const int iters = 20 * css;  // :149, :161
    if (solver == FPR_COARSE_JACOBI) {
        if (int rc = fprx_sumsq_scaled_dev(ctx, rhs, N, 1.0, ctx->scalars + 3, 0)) return rc;  // :150
        k_state_init<<<1, 1, 0, s>>>(ctx->state, ctx->scalars + 3, tol, (double)N, 0);
        FPR_CHECK_LAUNCH(ctx);
        ctx->state_h->done = 0; ctx->state_h->iters = 0; ctx->state_h->last_rms = 0.0;
        if (fpr_opt(ctx, "mg_multi", 1) && nx >= 32 && ny >= 32) {
            // groups of S fused sweeps per launch; the exit test is replayed per sweep on the device
            constexpr int S = 8, TX = 16, TY = 16;
            const bool patch = fpr_opt(ctx, "mg_patch", 1) != 0;
            const dim3 gm((nx + TX - 1) / TX, (ny + TY - 1) / TY);
            const int nblk = (int)(gm.x * gm.y);
            if ((size_t)nblk * S > (size_t)FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
            int Sg = (int)fpr_opt(ctx, "mg_group_sweeps", S);  // sweeps per launch (<= S; tuning/diagnostic knob)
            if (Sg < 1 || Sg > S) Sg = S;
            const int groups = (iters + Sg - 1) / Sg;
            double* a = u;
            double* b = L.tmp;
            int gdone = 0;
            int chunk_groups = 8;
            if ((size_t)nblk * S * 2 > (size_t)FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
            while (gdone < groups) {
                // poll the solver state after 8 groups, then after ever longer chunks (up to 64 groups = 512 sweeps)
                const int gend = (gdone + chunk_groups < groups) ? gdone + chunk_groups : groups;
                if (chunk_groups < 64) chunk_groups *= 2;
                for (int gi = gdone; gi < gend; ++gi) {
                    const int nsw = (iters - gi * Sg < Sg) ? iters - gi * Sg : Sg;
                    if (patch) {
                        // group gi writes partial slot gi&1 and first replays the exit test of group gi-1
                        double* slot = ctx->partials + (size_t)(gi & 1) * S * nblk;
                        const double* pslot = ctx->partials + (size_t)((gi + 1) & 1) * S * nblk;
                        const bool pending = gi > gdone;  // the previous group of THIS chunk is still unchecked
                        k_jacobi_patch<S, true, true><<<gm, 256, 0, s>>>(a, rhs, b, nx, ny, C, _h2, fac, nsw, slot, ctx->state,
                                                                          pslot, pending ? Sg : 0, gi - 1, (double)N);
                        if (gi == gend - 1)  // last group of the chunk: stand-alone check before the host polls
                            k_jacobi_check_multi<<<1, 256, 0, s>>>(ctx->state, slot, nblk, nsw, (double)N, gi);
                    } else {
                        k_sweep2d_multi<S, TX, TY, 2, true><<<gm, 256, 0, s>>>(a, rhs, b, nx, ny, C, _h2, fac, nsw, ctx->partials, ctx->state);
                        k_jacobi_check_multi<<<1, 256, 0, s>>>(ctx->state, ctx->partials, nblk, nsw, (double)N, gi);
                    }
                    double* t = a; a = b; b = t;
                }
                FPR_CHECK_LAUNCH(ctx);
                gdone = gend;
                if (int rc = read_state(ctx)) return rc;
                if (ctx->state_h->done) break;
            }
            double* result;
            if (ctx->state_h->done) {
                const int gs = ctx->state_h->group, redo = ctx->state_h->redo;
                double* in = (gs & 1) ? L.tmp : u;
                double* out = (gs & 1) ? u : L.tmp;
                const int nsw_g = (iters - gs * Sg < Sg) ? iters - gs * Sg : Sg;
                if (redo < nsw_g) {  // the exit fell inside the group: recompute exactly `redo` sweeps from its input
                    if (patch) k_jacobi_patch<S, false, false><<<gm, 256, 0, s>>>(in, rhs, out, nx, ny, C, _h2, fac, redo, nullptr, nullptr, nullptr, 0, 0, 0.0);
                    else k_sweep2d_multi<S, TX, TY, 0, false><<<gm, 256, 0, s>>>(in, rhs, out, nx, ny, C, _h2, fac, redo, nullptr, nullptr);
                }
                result = out;
            } else {
                result = (groups & 1) ? L.tmp : u;
            }
            if (result != u) FPR_HIP(ctx, hipMemcpyAsync(u, result, N * sizeof(double), hipMemcpyDeviceToDevice, s));
            FPR_CHECK_LAUNCH(ctx);
            ctx->last_coarse_iters += ctx->state_h->iters;
            *rms_out_host = ctx->state_h->last_rms;
            *rms_is_host = true;
            return FPR_OK;
        }
        const int chunk = 64;
        int launched = 0;
        double* a = u;
        double* b = L.tmp;
        while (launched < iters) {
            const int m = (iters - launched < chunk) ? iters - launched : chunk;
            for (int i = 0; i < m; ++i) {
                k_sweep2d<true, true><<<g, blk2, 0, s>>>(a, rhs, b, nx, ny, C, _h2, fac, ctx->partials, ctx->state);
                k_jacobi_check<<<1, 256, 0, s>>>(ctx->state, ctx->partials, np, (double)N);
                double* t = a; a = b; b = t;
            }
            FPR_CHECK_LAUNCH(ctx);
            launched += m;
            if (int rc = read_state(ctx)) return rc;
            if (ctx->state_h->done) break;
        }
        // the solution sits in u after an even number of executed sweeps, else in tmp
        if (ctx->state_h->iters & 1)
            FPR_HIP(ctx, hipMemcpyAsync(u, L.tmp, N * sizeof(double), hipMemcpyDeviceToDevice, s));
    } else if (solver == FPR_COARSE_CG) {
        if (int rc = cg_solve(ctx, u, rhs, h, h, c, tol, iters, nx, ny)) return rc;  // :162
    } else {
        return fpr_fail(ctx, FPR_ERR_INVALID, "unknown coarse solver %d", solver);  // :163-165 error()
    }
    ctx->last_coarse_iters += ctx->state_h->iters;
    *rms_out_host = ctx->state_h->last_rms;
    *rms_is_host = true;
    return FPR_OK;
}

// ---- finest level of fpr_mgsolve2d when consecutive cycles share a pass (k_seam_march) -------------------------------
struct TopGeom {
    int nx, ny, rpc, rpc_s, nstrips, nstrips_r, nstrips_s, ntf;
    dim3 gm, gr, gs;
    double C, _h2, fac;
};

static TopGeom top_geom(fpr_ctx* ctx, int nx, int ny, double h, double c)
{
    TopGeom g;
    g.nx = nx; g.ny = ny;
    g.C = 4.0 + c * (h * h); g._h2 = 1 / (h * h);
    g.fac = (4.0 / 5.0) * ((h * h) / (4.0 + c * (h * h)));
    g.ntf = fpr_opt(ctx, "mg_nt", 0) ? 256 : 0;
    g.nstrips = (nx + 59) / 60;      // as vcycle_level (one column per lane)
    g.nstrips_r = (nx + 57) / 58;
    g.nstrips_s = (nx + 53) / 54;    // k_seam_march: 54 owned columns per strip
    int rpc = (int)fpr_opt(ctx, "mg_rows_per_chunk", 0);
    if (rpc <= 0) {
        const long target = fpr_opt(ctx, "mg_wave_target", 4096);
        rpc = 64;
        while (rpc > 16 && (long)g.nstrips * ((ny + rpc - 1) / rpc) < target) rpc >>= 1;
    }
    g.rpc = rpc;
    // k_seam_march holds 3 waves per SIMD (151 VGPRs) = 3 workgroups per CU, and a workgroup works for most of the pass:
    // the chunks are made as tall as a single round allows (all workgroups resident at once, ~95 % of the slots) -- the
    // 9 overlap rows weigh less and no second, half-empty round trails (4097^2: 38 chunks of 108 rows 127 us, 65 chunks of
    // 64 rows 133 us, 52 of 80 rows 150 us)
    int rs = (int)fpr_opt(ctx, "mg_seam_rows_per_chunk", 0);
    if (rs <= 0) {
        if (ctx->ncu <= 0) {
            int v = 0;
            ctx->ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ? v : 256;
        }
        const int gx = (g.nstrips_s + 3) / 4;
        int chunks = (int)(0.95 * 3 * ctx->ncu) / gx;
        if (chunks < 1) chunks = 1;
        rs = (ny + chunks - 1) / chunks;
        if (rs < 32) rs = 32;
    }
    {   // rows are addressed relative to the first row of a chunk with a 32-bit byte offset
        const long cap = (long)(0x7fffffffL / ((long)nx * 8)) - 16;
        if (rs > cap) rs = (int)(cap > 16 ? cap : 16);
    }
    g.rpc_s = rs;
    g.gm = dim3((g.nstrips + 3) / 4, (ny + rpc - 1) / rpc);
    g.gr = dim3((g.nstrips_r + 3) / 4, (ny + rpc - 1) / rpc);
    g.gs = dim3((g.nstrips_s + 3) / 4, (ny + g.rpc_s - 1) / g.rpc_s);
    return g;
}

// pre-smoothing pair + residual + injection + zero coarse guess (:124-132): uin -> out, L.res_c, corr_zero
static int top_pre(fpr_ctx* ctx, const TopGeom& g, const double* uin, const double* rhs, double* out, double* res_c,
                   double* corr_zero, int apply_BCs, const int* skp)
{
    hipStream_t s = ctx->stream[0];
    const bool timed = fpr_ktimer_begin(ctx, FPR_KT_MG_PRE, s);
    k_smooth2_march<false, false, true><<<g.gr, 256, 0, s>>>(uin, rhs, out, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc, g.nstrips_r, nullptr,
                                                             nullptr, apply_BCs | g.ntf, res_c, corr_zero, skp);   // (:355-357 included)
    fpr_ktimer_end(ctx, timed, s);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// prolongation + correction + post-smoothing pair (:136-143): X - P(corr) -> uout; norm: block sums into ctx->partials
static int top_post(fpr_ctx* ctx, const TopGeom& g, const double* X, const double* rhs, const double* corr, double* uout,
                    bool norm, int apply_BCs, const int* skp)
{
    hipStream_t s = ctx->stream[0];
    const bool timed = fpr_ktimer_begin(ctx, FPR_KT_MG_POST, s);
    if (norm) k_smooth2_march<true, true, false><<<g.gm, 256, 0, s>>>(X, rhs, uout, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc, g.nstrips, ctx->partials, corr, apply_BCs | g.ntf, nullptr, nullptr, skp);
    else k_smooth2_march<false, true, false><<<g.gm, 256, 0, s>>>(X, rhs, uout, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc, g.nstrips, nullptr, corr, apply_BCs | g.ntf, nullptr, nullptr, skp);
    fpr_ktimer_end(ctx, timed, s);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// the seam: post pair of cycle k (+ norm) and pre pair + residual + injection of cycle k+1: X - P(corr) -> Y, res_c, corr_zero
static int top_seam(fpr_ctx* ctx, const TopGeom& g, const double* X, const double* rhs, const double* corr, double* Y,
                    double* res_c, double* corr_zero, int apply_BCs, const int* skp)
{
    hipStream_t s = ctx->stream[0];
    const bool timed = fpr_ktimer_begin(ctx, FPR_KT_MG_SEAM, s);
    if (apply_BCs) k_seam_march<true><<<g.gs, 256, 0, s>>>(X, rhs, Y, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc_s, g.nstrips_s, ctx->partials, corr, res_c, corr_zero, skp);
    else k_seam_march<false><<<g.gs, 256, 0, s>>>(X, rhs, Y, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc_s, g.nstrips_s, ctx->partials, corr, res_c, corr_zero, skp);
    fpr_ktimer_end(ctx, timed, s);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

static int vcycle_run(fpr_ctx* ctx, double* u, const double* rhs, double h, double c, double tol, int css, int solver,
                      int apply_BCs, int nx, int ny, double* rms_host)
{
    std::vector<FprLevel>* A;
    if (int rc = get_arena(ctx, nx, ny, css, &A)) return rc;
    double r = 0.0;
    bool is_host = true;
    ctx->used_small = false;
    ctx->top_is_coarsest = false;
    if (!ctx->cyc_skip)   // (cycles enqueued ahead accumulate over the whole solve: k_cycle_init zeroed it)
        FPR_HIP(ctx, hipMemsetAsync(&ctx->state->acc_iters, 0, sizeof(int), ctx->stream[0]));
    if (int rc = vcycle_level(ctx, *A, 0, u, rhs, h, c, tol, css, solver, apply_BCs, true, &r, &is_host)) return rc;
    if (rms_host) {
        if (ctx->used_small)  // fetch the device-side iteration count / coarse rms with the same sync
            FPR_HIP(ctx, hipMemcpyAsync(ctx->state_h, ctx->state, sizeof(FprSolveState), hipMemcpyDeviceToHost, ctx->stream[0]));
        if (!is_host) {
            double ssum;
            if (int rc = read_scalar(ctx, ctx->scalars, &ssum)) return rc;
            r = sqrt(ssum / ((double)nx * (double)ny));  // multigrid.jl:252
            if (ctx->top_is_coarsest) r = ctx->state_h->last_rms;
        }
        if (ctx->used_small) ctx->last_coarse_iters += ctx->state_h->acc_iters;
        *rms_host = r;
    }
    return FPR_OK;
}

extern "C" int fpr_vcycle2d(fpr_ctx* ctx, double* u_f, const double* rhs, double h, double c, double tol,
                            int coarse_solve_size, int coarse_solver, int apply_BCs, int nx, int ny, double* rms_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, u_f && rhs, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    ctx->last_coarse_iters = 0;
    return vcycle_run(ctx, u_f, rhs, h, c, tol, coarse_solve_size, coarse_solver, apply_BCs, nx, ny, rms_host);
}

extern "C" int fpr_mgsolve2d(fpr_ctx* ctx, double* u, const double* f, double h, double c, double tol, int niters,
                             int apply_BCs, int coarse_solve_size, int coarse_solver, int nx, int ny, double* rms_host,
                             int* ncycles_host, double* history_host, double* frms_host, int* converged_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, u && f, "null pointer");
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    {   // multigrid.jl:45-46
        const int m = coarse_solve_size - 1;
        if (coarse_solve_size > (nx < ny ? nx : ny) || m <= 0 || (m & (m - 1)) != 0)
            return fpr_fail(ctx, FPR_ERR_ASSERT, "@assert failed: coarse_solve_size=%d (must be 2^l+1 and <= min(nx,ny))",
                            coarse_solve_size);
    }
    const size_t N = (size_t)nx * ny;
    double f_rms = 0.0, tolf = 0.0;   // :53, :70
    double r_rms = 0.0;
    int n = 0;
    ctx->last_coarse_iters = 0;
    int ahead = (int)fpr_opt(ctx, "mg_ahead", 1);
    if (ahead > FPR_CYC_SLOTS - 2) ahead = FPR_CYC_SLOTS - 2;
    const bool streams = ahead > 0 && niters > 1 && vcycle_streams(ctx, nx, ny, coarse_solve_size, coarse_solver);
    if (!streams) {
        double fs;
        if (int rc = fpr_sumsq_scaled(ctx, f, N, 1.0, &fs)) return rc;
        f_rms = sqrt(fs / (double)N);
        tolf = tol * f_rms;
    }
    if (streams) {
        // Cycles are enqueued `ahead` deep before the host waits for the norm of the oldest one: the exit test (:70) is
        // evaluated on the device by k_cycle_finish, and all launches of a cycle that follows the one that met it return
        // at once -- fields, norms and cycle count are those of the plain loop, without a host round trip per cycle.
        struct Guard { fpr_ctx* c; ~Guard() { c->cyc_skip = nullptr; } } guard{ctx};
        // rms(f) and the threshold tol * rms(f) stay on the device too (same operations as on the host): no round trip
        // before the first cycle; the host learns both from the first record
        if (int rc = fprx_cycle_init(ctx, f, N, tol)) return rc;
        ctx->cyc_skip = &ctx->cyc->stop;
        const int* skp = ctx->cyc_skip;
        int enq = 0;
        if (fpr_opt(ctx, "mg_seam", 1) && fpr_opt(ctx, "mg_vx", 1) != 2) {
            // ---- consecutive cycles share their pass over the finest grid (k_seam_march) ----
            // unit k = the end of cycle k: either the plain post-smoothing pass (k = niters, or the norms seen so far say
            // that cycle k will meet the exit test) or a seam pass that also starts cycle k+1, followed by cycle k+1's
            // coarser levels.  If a seam's norm ends the loop, everything enqueued behind it returns at once and the plain
            // post-smoothing pass is replayed from the seam's inputs, which nothing has touched.
            std::vector<FprLevel>* A;
            if (int rc = get_arena(ctx, nx, ny, coarse_solve_size, &A)) return rc;
            FprLevel& L = (*A)[0];
            if (!L.res_c) return fpr_fail(ctx, FPR_ERR_INVALID, "level arena exhausted");
            const size_t nc = (size_t)(1 + (nx - 1) / 2) * (size_t)(1 + (ny - 1) / 2);
            if (!L.tmp2) FPR_HIP(ctx, hipMalloc(&L.tmp2, N * sizeof(double)));
            if (!L.corr_c2) FPR_HIP(ctx, hipMalloc(&L.corr_c2, nc * sizeof(double)));
            double* corr[2] = {L.corr_c, L.corr_c2};
            const TopGeom g = top_geom(ctx, nx, ny, h, c);
            const int npm = (int)(g.gm.x * g.gm.y), nps = (int)(g.gs.x * g.gs.y);
            if (npm > FPR_MAX_PARTIALS || nps > FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
            struct Unit { bool seam; const double* X; int p; } units[FPR_CYC_SLOTS];
            double* X = nullptr;
            int p = 0;
            bool need_head = true;
            double r_prev = 0.0, r_last = 0.0;   // the last two norms the host has seen
            ctx->used_small = false;
            auto lower = [&](int pp) -> int {
                double dummy; bool dh;
                return vcycle_level(ctx, *A, 1, corr[pp], L.res_c, h * 2, c, tol, coarse_solve_size, coarse_solver, apply_BCs, false, &dummy, &dh);
            };
            auto enqueue_unit = [&]() -> int {
                const int k = enq + 1;
                if (need_head) {   // first cycle, or the loop goes on after a plain post-smoothing pass
                    if (apply_BCs)
                        if (int rc = fpr_bc2d(ctx, u, nx, ny)) return rc;  // :60-62
                    if (int rc = top_pre(ctx, g, u, f, L.tmp, L.res_c, corr[0], apply_BCs, skp)) return rc;
                    if (int rc = lower(0)) return rc;
                    X = L.tmp; p = 0; need_head = false;
                }
                bool last = (k == niters);
                // mg_seam_predict: 1 = extrapolate (default); 0 = never (every cycle but the niters-th ends in a seam, the
                // last one is replayed); 2 = odd cycles end in a plain pass (exercises the restart after a wrong guess)
                const long predict = fpr_opt(ctx, "mg_seam_predict", 1);
                if (predict == 2 && (k & 1)) last = true;
                if (!last && predict == 1 && n >= 2 && r_last < r_prev && r_last > 0.0) {   // geometric extrapolation of the norm to cycle k
                    double pred = r_last;
                    const double rate = r_last / r_prev;
                    for (int q = n; q < k; ++q) pred *= rate;
                    last = pred < tolf;
                }
                const int slot = enq % FPR_CYC_SLOTS;
                if (last) {
                    if (int rc = top_post(ctx, g, X, f, corr[p], u, true, apply_BCs, skp)) return rc;
                    if (int rc = fprx_cycle_finish(ctx, ctx->partials, npm, ctx->scalars, (double)N, slot)) return rc;
                    units[slot] = {false, X, p};
                    need_head = true;
                } else {
                    double* Y = (X == L.tmp) ? L.tmp2 : L.tmp;
                    if (int rc = top_seam(ctx, g, X, f, corr[p], Y, L.res_c, corr[1 - p], apply_BCs, skp)) return rc;
                    if (int rc = fprx_cycle_finish(ctx, ctx->partials, nps, ctx->scalars, (double)N, slot)) return rc;
                    if (int rc = lower(1 - p)) return rc;   // cycle k+1 below the finest level
                    units[slot] = {true, X, p};
                    X = Y; p = 1 - p;
                }
                ++enq;
                return FPR_OK;
            };
            while (true) {
                // (nothing is enqueued behind a cycle that is expected to end the loop: if it does, nothing is skipped)
                while (enq < niters && enq - n < 1 + ahead && !(need_head && enq > n))
                    if (int rc = enqueue_unit()) return rc;
                if (n >= enq) break;
                const int slot = n % FPR_CYC_SLOTS;
                FprCycleCtl rec;
                if (int rc = fprx_cycle_wait(ctx, slot, n + 1, &rec)) return rc;
                r_rms = rec.rms;
                f_rms = rec.frms; tolf = rec.tolf;
                r_prev = r_last; r_last = r_rms;
                ctx->last_coarse_iters = rec.coarse_iters;
                if (history_host) history_host[n] = r_rms;
                ++n;
                if (rec.stop) {   // :70 (taken on the device)
                    if (units[slot].seam)   // u at the end of this cycle was never stored: replay its post-smoothing pass
                        if (int rc = top_post(ctx, g, units[slot].X, f, corr[units[slot].p], u, false, apply_BCs, nullptr)) return rc;
                    break;
                }
            }
        } else {
        auto enqueue_cycle = [&]() -> int {
            if (apply_BCs)
                if (int rc = fpr_bc2d(ctx, u, nx, ny)) return rc;  // :60-62
            const int slot = enq % FPR_CYC_SLOTS;
            ctx->cyc_slot = slot;   // k_cycle_finish writes the record of this cycle into pinned host memory itself
            if (int rc = vcycle_run(ctx, u, f, h, c, tol, coarse_solve_size, coarse_solver, apply_BCs, nx, ny, nullptr)) return rc;
            ++enq;
            return FPR_OK;
        };
        while (true) {
            while (enq < niters && enq - n < 1 + ahead)
                if (int rc = enqueue_cycle()) return rc;
            if (n >= enq) break;
            const int slot = n % FPR_CYC_SLOTS;
            FprCycleCtl rec;
            if (int rc = fprx_cycle_wait(ctx, slot, n + 1, &rec)) return rc;
            r_rms = rec.rms;
            f_rms = rec.frms; tolf = rec.tolf;
            ctx->last_coarse_iters = rec.coarse_iters;
            if (history_host) history_host[n] = r_rms;
            ++n;
            if (rec.stop) break;  // :70 (taken on the device)
        }
        }
    } else
    for (int iter = 1; iter <= niters; ++iter) {
        if (apply_BCs)
            if (int rc = fpr_bc2d(ctx, u, nx, ny)) return rc;  // :60-62
        if (int rc = vcycle_run(ctx, u, f, h, c, tol, coarse_solve_size, coarse_solver, apply_BCs, nx, ny, &r_rms)) return rc;
        if (history_host) history_host[n] = r_rms;
        ++n;
        if (r_rms < tolf) break;  // :70
    }
    if (rms_host) *rms_host = r_rms;
    if (ncycles_host) *ncycles_host = n;
    if (frms_host) *frms_host = f_rms;
    if (converged_host) *converged_host = !(r_rms > tolf);  // :78-80 @warn condition
    return FPR_OK;
}

extern "C" long fpr_last_coarse_iters(fpr_ctx* ctx) { return ctx ? ctx->last_coarse_iters : 0; }
