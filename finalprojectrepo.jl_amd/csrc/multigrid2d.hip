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
#include <cmath>

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

#include "mg_march.hpp"   // k_smooth2_march, k_seam_march, k_smooth2_march2: the register-rolling marches of the fine levels

// the two-sweep pass without a carried finish
template <bool N, bool P, bool R, class... A>
static inline void march_go(fpr_ctx* ctx, dim3 g, hipStream_t s, A... a)
{
    k_smooth2_march_v2<N, P, R><<<g, 256, 0, s>>>(a..., FprFinishArgs{});
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
template <int S, int P, bool NORM, bool STATE>
__global__ __launch_bounds__((P / 2) * (P / 2)) void k_jacobi_patch(const double* __restrict__ uin, const double* __restrict__ f,
                                                                    double* __restrict__ uout, int nx, int ny, double C,
                                                                    double _h2, double fac, int nsw, double* __restrict__ partials,
                                                                    FprSolveState* __restrict__ state,
                                                                    const double* __restrict__ prev_partials, int prev_nsw,
                                                                    int prev_group, double Ntot)
{
    // region = P x P points, own tile = the inner T x T (S cells away from the region's edge); (P/2)^2 threads, 2x2 points each
    constexpr int T = P - 2 * S;
    constexpr int HT = P / 2;                    // threads per side
    constexpr int NT = HT * HT, NWV = (NT + 63) / 64;
    constexpr int SPW = (S + NWV - 1) / NWV;     // sweeps whose partial lists one wave sums in the replay
    static_assert(T > 0 && P % 2 == 0 && S <= 64, "patch geometry");
    __shared__ __attribute__((aligned(16))) double img[2][P * P];
    __shared__ double red[NWV][S];
    __shared__ int stop_flag;
    const int tid = threadIdx.x;
    // the field loads are issued first so that they overlap the replay of the previous group's exit test
    const int ty = tid / HT, tx = tid - ty * HT;
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
        // wave w sums the partials of sweeps w, w+NWV, ... (per sweep the same strided sum as k_jacobi_check_multi), lanes
        // 0..prev_nsw-1 of wave 0 then evaluate sqrt(sum/N) < thresh in parallel and a ballot finds the first sweep that met it.
        const int done0 = state->done;
        const double thresh = state->thresh;
        const int nblk = gridDim.x * gridDim.y;
        const int lane = tid & 63, wv = tid >> 6;
        double part_acc[SPW];
#pragma unroll
        for (int h = 0; h < SPW; ++h) part_acc[h] = 0.0;
        if (prev_nsw > 0) {
#pragma unroll
            for (int h = 0; h < SPW; ++h) {
                const int sidx = wv + NWV * h;
                if (sidx < prev_nsw)
                    for (int i = lane; i < nblk; i += 64) part_acc[h] += prev_partials[(size_t)sidx * nblk + i];
            }
        }
        if (done0) return;
        if (prev_nsw > 0) {
#pragma unroll
            for (int h = 0; h < SPW; ++h) {
                const int sidx = wv + NWV * h;
                const double a = fpr_wave_sum_all(part_acc[h]);
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
            double wl0, wl1, er0, er1;
            if constexpr (HT == 16) {
                // a row of threads is one 16-lane DPP row: the x-neighbours of a patch come from the adjacent lanes' registers
                // (zero beyond the region's edge: like the clamped reads, such garbage stays more than S cells from the own tile)
                wl0 = fpr_dpp<0x111>(u[0][1]); wl1 = fpr_dpp<0x111>(u[1][1]);   // lane i <- lane i-1: its right column
                er0 = fpr_dpp<0x101>(u[0][0]); er1 = fpr_dpp<0x101>(u[1][0]);   // lane i <- lane i+1: its left column
            } else {
                wl0 = cur[xl + P * ly]; wl1 = cur[xl + P * (ly + 1)];
                er0 = cur[xr + P * ly]; er1 = cur[xr + P * (ly + 1)];
            }
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
            const double v = fpr_wave_sum_all(acc[s]);
            if (lane == 0) red[wv][s] = v;
        }
        __syncthreads();
        if (tid < S) {
            const int blk = blockIdx.x + gridDim.x * blockIdx.y, nblk = gridDim.x * gridDim.y;
            double v = red[0][tid];
#pragma unroll
            for (int w = 1; w < NWV; ++w) v += red[w][tid];
            partials[(size_t)tid * nblk + blk] = v;
        }
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
// behind a persistent Jacobi launch that gave up: the solve goes on (iters and thresh stay)
__global__ void k_state_resume(FprSolveState* st)
{
    st->done = 0;
    st->redo = 0;
    st->group = -1;
}

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
        a = fpr_wave_sum_all(a);   // (the same sum as the replay in the prologue of k_jacobi_patch)
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

#include "mg_cg.hpp"      // cg! in five, three and two launches per iteration

#include "mg_small.hpp"   // k_mg_small: the LDS-resident sub-hierarchy

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
#include <atomic>
// upper half of every granule's tag: solves so far in this PROCESS (all contexts; 31 bits: a wrap needs 2^31 coarse solves)
static long long fpr_next_epoch(fpr_ctx* ctx)
{
    static std::atomic<long long> epoch{0};
    const long long e = (epoch.fetch_add(1) + 1) & 0x7fffffffLL;
    ctx->jacp_epoch = e ? e : ((epoch.fetch_add(1) + 1) & 0x7fffffffLL);
    return ctx->jacp_epoch;
}

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
        // The data-tagged hand-offs (k_jacobi_persist_tag, k_cg_persistent's tagged edges) accept a granule as soon as its tag is the one
        // waited for: memory that comes back from hipMalloc may hold another context's granules.  Tag 0 is never wanted, and
        // epochs come from one process-wide counter (fpr_next_epoch), so no two solves of a process share a tag.
        FPR_HIP(ctx, hipMemsetAsync(ctx->cg_buf, 0, 5 * n * sizeof(double), ctx->stream[0]));
    }
    double* b = ctx->cg_buf;
    w->r = b; w->p = b + n; w->ph = b + 2 * n; w->x = b + 3 * n; w->p2 = b + 4 * n; w->n = n;
    return FPR_OK;
}

#include "mg_cg_persistent.hpp"   // k_cg_persistent: cg! as one launch
#include "mg_jacobi_persistent.hpp"   // k_jacobi_persist: many groups of sweeps per launch, neighbour-to-neighbour hand-offs

// runs cg! on the compute stream; leaves iters / last_rms in ctx->state_h (synchronises)
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
    if (2L * np2 > FPR_MAX_PARTIALS || 2L * fg > FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");   // (s, e) pairs
    // r = p = p_hat = b, x = 0, rho = b.b, state: what every form of the solve starts from (b itself is never written, so a
    // form that gives up -- the persistent kernel on a barrier time-out -- can be followed by another from the start)
    auto start = [&]() -> int {
        k_cg_init<<<fg, 256, 0, s>>>(b, w.r, w.p, w.ph, w.x, N);
        FPR_CHECK_LAUNCH(ctx);
        if (int rc = fprx_dot2_dev(ctx, b, b, N, ctx->scalars + 2)) return rc;   // krylov.jl:57,64 as a Dot2 sum, like every dot product of cg!
        k_state_init<<<1, 1, 0, s>>>(ctx->state, ctx->scalars + 2, tol, (double)N, 1);
        FPR_CHECK_LAUNCH(ctx);
        return FPR_OK;
    };
    if (int rc = start()) return rc;
    const int chunk = 64;
    // cg_fused: 3 (default where it applies) = one persistent launch for the whole solve, 2 = two dependent launches per
    // iteration, 1 = three, 0 = five (one per operation)
    long fused = fpr_opt(ctx, "cg_fused", 3);
    if (fused == 3) {
        // geometry: 64 workgroups of 256 threads (one wave per SIMD)
        constexpr int nbx = 8, nb = nbx * nbx, nt = 256;
        const int twm = (nx + nbx - 1) / nbx, thm = (ny + nbx - 1) / nbx;
        const size_t lds = (size_t)(twm + 2) * (thm + 2) * sizeof(double);
        // its workgroups synchronise through memory, so all of them have to be resident at once: ask the runtime once
        // whether a compute unit takes one (this many threads, this much LDS) and whether the device has enough units
        int& resident = ctx->cgp_resident64;
        if (resident < 0) {
            int per_cu = 0, ncu = 0;
            const hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cg_persistent<64, 8, 256>, nt, 64 * 1024);
            const bool ok = eo == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess;
            resident = (ok && per_cu >= 1 && ncu >= nb) ? 1 : 0;    // one workgroup per unit: a unit of its own for every workgroup
        }
        if (resident == 1 && Nmax > 0 && (long)twm * thm <= (long)nt * CGP_PPT &&
            2L * (twm + 2) + 2L * thm <= (long)nt * CGP_RPT && lds <= 64 * 1024) {
            CgpArgs a;
            a.b = b; a.x_out = x_in; a.r_glob = w.r; a.part = ctx->partials;
            a.ctr = (unsigned*)(ctx->scalars + 40);   // two words of the scalar block: barrier counter, abort flag
            a.st = ctx->state;
            a.nx = nx; a.ny = ny; a.Nmax = Nmax;
            a.hx2 = hx * hx; a.hy2 = hy * hy; a.c = c; a.tol = tol; a.N = (double)N;
            {   // a division by a power of two is an exact scaling, and so is the multiplication by its (exact) reciprocal
                int ex = 0, ey = 0;
                const bool px = std::frexp(a.hx2, &ex) == 0.5 && ex > -1000 && ex < 1000;
                const bool py = std::frexp(a.hy2, &ey) == 0.5 && ey > -1000 && ey < 1000;
                a.pow2 = (px && py) ? 1 : 0;
                a.ihx2 = a.pow2 ? 1.0 / a.hx2 : 0.0;
                a.ihy2 = a.pow2 ? 1.0 / a.hy2 : 0.0;
            }
            a.fences = fpr_opt(ctx, "handoff_fences", 0) != 0 ? 1 : 0;
            FPR_HIP(ctx, hipMemsetAsync(a.ctr, 0, 2 * sizeof(unsigned), s));
            k_cgp_slots_init<<<1, 64, 0, s>>>(reinterpret_cast<unsigned long long*>(a.part), nb);
            // An ordinary launch: the workgroups are resident together on any device this library runs on (one per CU, 256 CUs),
            // every wait is bounded, and a cooperative launch moves the process onto the runtime's cooperative queue -- after it,
            // kernels of two streams no longer overlap (measured: the side-by-side T / W solves of the NS step 1.61 -> 1.96 ms,
            // tools/exp_ns_only.py).
            const bool timed = fpr_ktimer_begin(ctx, FPR_KT_MG_CG, s);
            k_cg_persistent<64, 8, 256><<<dim3(nb), dim3(nt), lds, s>>>(a);
            fpr_ktimer_end(ctx, timed, s);
            FPR_CHECK_LAUNCH(ctx);
            if (int rc = read_state(ctx)) return rc;
            if (ctx->state_h->done >= 0) return FPR_OK;   // x_in holds the solution (krylov.jl:88)
            // A grid barrier timed out (a workgroup never became resident: the card is shared with other work): x_in is
            // partly written, b is intact -- solve again from the start with the two-launch form, which needs no residency
            ctx->options["cg_persistent_timeouts"] = fpr_opt(ctx, "cg_persistent_timeouts", 0) + 1;
            if (int rc = start()) return rc;
        }
        fused = 2;
    }
    const dim3 gcg((nx + CGX - 1) / CGX, (ny + CGY - 1) / CGY);
    const int npcg = (int)(gcg.x * gcg.y);
    if (2L * npcg > FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
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

// The finest level's two ping-pong partners from the caller (role of prealloc_dict's fine-level buffers, multigrid.jl:25-38, 49-51):
// the passes over the finest grid stream u / f and these two at equal offsets, and on MI355X their speed depends on which physical
// pages the allocations received -- a host that places its arrays can place these two as well.  NULL = the library's own.
extern "C" int fpr_mg_arena_provide(fpr_ctx* ctx, int nx, int ny, double* tmp, double* tmp2)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    FPR_REQUIRE(ctx, (((uintptr_t)tmp | (uintptr_t)tmp2) & 7) == 0, "buffers must be 8-byte aligned");
    std::vector<FprLevel>* A = nullptr;
    if (int rc = get_arena(ctx, nx, ny, 0, &A)) return rc;
    FprLevel& L = (*A)[0];
    for (int s = 0; s < 3; ++s) FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[s]));   // nothing in flight uses the buffers that leave
    const size_t bytes = (size_t)nx * ny * sizeof(double);
    auto swap_in = [&](double*& slot, bool& own, double* mine) -> int {
        if (mine) {
            if (slot && own) FPR_HIP(ctx, hipFree(slot));
            slot = mine; own = false;
        } else if (!own || !slot) {      // back to a buffer of the library's own
            slot = nullptr; own = true;
            FPR_HIP(ctx, hipMalloc(&slot, bytes));
        }
        return FPR_OK;
    };
    if (int rc = swap_in(L.tmp, L.own_tmp, tmp)) return rc;
    if (int rc = swap_in(L.tmp2, L.own_tmp2, tmp2)) return rc;
    return FPR_OK;
}

// ... and the three arrays of the first coarse level that the passes over the finest grid stream beside them at a quarter of the
// rate: the injected residual (res_c) and the two alternating correction buffers (corr_c, corr_c2), each (1 + (nx-1)/2) x (1 + (ny-1)/2)
// doubles.  Which mode a 4097^2 seam pass runs in (106 or 117-120 us) depended on the context's own allocation of these as much as on
// the four big arrays (tools/exp_mg_arena_rounds.py).  All three or none (NULL, NULL, NULL = the library's own again).
extern "C" int fpr_mg_arena_provide_coarse(fpr_ctx* ctx, int nx, int ny, double* res_c, double* corr_c, double* corr_c2)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (int rc = check_dims(ctx, nx, ny)) return rc;
    const bool all = res_c && corr_c && corr_c2, none = !res_c && !corr_c && !corr_c2;
    FPR_REQUIRE(ctx, all || none, "all three buffers or none");
    FPR_REQUIRE(ctx, (((uintptr_t)res_c | (uintptr_t)corr_c | (uintptr_t)corr_c2) & 15) == 0, "buffers must be 16-byte aligned");
    FPR_REQUIRE(ctx, none || (res_c != corr_c && res_c != corr_c2 && corr_c != corr_c2), "three distinct buffers");
    std::vector<FprLevel>* A = nullptr;
    if (int rc = get_arena(ctx, nx, ny, 0, &A)) return rc;
    FprLevel& L = (*A)[0];
    FPR_REQUIRE(ctx, L.res_c != nullptr, "this grid has no coarse level");
    for (int s = 0; s < 3; ++s) FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[s]));   // nothing in flight uses the buffers that leave
    const size_t nc = (size_t)(1 + (nx - 1) / 2) * (size_t)(1 + (ny - 1) / 2);
    if (all) {
        if (L.own_coarse) {   // the library's own step aside, untouched: NULL x 3 brings exactly these back
            if (!L.corr_c2) FPR_HIP(ctx, hipMalloc(&L.corr_c2, nc * sizeof(double)));
            L.coarse_own[0] = L.res_c; L.coarse_own[1] = L.corr_c; L.coarse_own[2] = L.corr_c2;
        }
        L.res_c = res_c; L.corr_c = corr_c; L.corr_c2 = corr_c2;
        L.own_coarse = false;
    } else if (!L.own_coarse) {
        L.res_c = L.coarse_own[0]; L.corr_c = L.coarse_own[1]; L.corr_c2 = L.coarse_own[2];
        L.coarse_own[0] = L.coarse_own[1] = L.coarse_own[2] = nullptr;
        L.own_coarse = true;
    }
    return FPR_OK;
}

#include "mg_mid.hpp"     // k_mid_down, k_mid_up: three launch-bound levels in two launches

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
                        double tol, int css, int solver, int apply_BCs, bool top, double* rms_out_host, bool* rms_is_host);

// Levels dA, dA+1, dA+2 (A > B > C) in two launches around the LDS-resident sub-hierarchy that starts at dA+3 (k_mid_down, k_mg_small,
// k_mid_up); *taken = false when the levels do not fit that form (nothing launched).  uA / rhsA / hA: solution, right-hand side and mesh
// width of level A.  A finish handed over by the loop (fprx_cycle_finish_defer) rides in one more row of k_mid_down's workgroups.  (A fourth
// level recomputed in k_mid_down's prologue, and the four levels as one launch with tagged halos between workgroups, were built, are bit-exact
// and slower: EXPERIMENTS 13.14, 13.16.)
static int mid_path(fpr_ctx* ctx, std::vector<FprLevel>& A, size_t d, double* uA, const double* rhsA, double hA, double c, double tol, int css,
                    int solver, int apply_BCs, bool* taken)
{
    *taken = false;
    if (!(solver == FPR_COARSE_JACOBI && fpr_opt(ctx, "mg_mid", 1) && fpr_opt(ctx, "mg_small", 1) &&
          fpr_opt(ctx, "mg_multi", 1) == 1 && fpr_opt(ctx, "mg_fuse_restrict", 1) && fpr_opt(ctx, "mg_fuse_prolong", 1) &&
          d + 3 < A.size()))
        return FPR_OK;
    hipStream_t s = ctx->stream[0];
    const int* skp = ctx->cyc_skip;
    const int nx = A[d].nx, ny = A[d].ny;
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
    if (!ok) return FPR_OK;
    static bool attr_set = false;
    if (!attr_set) {
        FPR_HIP(ctx, hipFuncSetAttribute((const void*)k_mid_down, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FPR_HIP(ctx, hipFuncSetAttribute((const void*)k_mid_up, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    MidArgs a;
    double hk = hA;
    for (int k = 0; k < 3; ++k) {
        FprLevel& Lk = A[d + k];
        a.L[k].f = (k == 0) ? rhsA : A[d + k - 1].res_c;
        a.L[k].tmp = Lk.tmp;
        a.L[k].fout = Lk.res_c;
        a.L[k].nx = Lk.nx; a.L[k].ny = Lk.ny;
        a.L[k].C = 4.0 + c * (hk * hk);
        a.L[k]._h2 = 1 / (hk * hk);
        a.L[k].fac = (4.0 / 5.0) * ((hk * hk) / (4.0 + c * (hk * hk)));
        hk = hk * 2;   // the recursion passes h*2 (multigrid.jl:133)
    }
    a.zfuse = fpr_opt(ctx, "mg_zero_fuse", 1) != 0;
    a.nxD = A[d + 3].nx; a.nyD = A[d + 3].ny;
    a.uD = A[d + 2].corr_c;
    a.uA = uA;
    a.apply_BCs = apply_BCs;
    a.skip = skp;
    a.fin = ctx->fin;                 // (partials == null: nothing handed over)
    ctx->fin = FprFinishArgs{};
    const dim3 gd((a.nxD - 1) / MID_TD > 0 ? (a.nxD - 1) / MID_TD : 1, ((a.nyD - 1) / MID_TD > 0 ? (a.nyD - 1) / MID_TD : 1) + (a.fin.partials ? 1 : 0));
    const dim3 gu((nx - 1) / MID_TA > 0 ? (nx - 1) / MID_TA : 1, (ny - 1) / MID_TA > 0 ? (ny - 1) / MID_TA : 1);
    k_mid_down<<<gd, MID_NT_DOWN, lds_down * sizeof(double), s>>>(a);   // :124-132 of levels d, d+1, d+2 (and of the level above)
    FPR_CHECK_LAUNCH(ctx);
    a.fin = FprFinishArgs{};
    double dummy; bool dh;
    if (int rc = vcycle_level(ctx, A, d + 3, A[d + 2].corr_c, A[d + 2].res_c, hk, c, tol, css, solver, apply_BCs, false, &dummy, &dh))
        return rc;  // :133 (k_mg_small)
    k_mid_up<<<gu, MID_NT_UP, lds_up * sizeof(double), s>>>(a);       // :136-143 of levels d+2, d+1, d
    FPR_CHECK_LAUNCH(ctx);
    *taken = true;
    return FPR_OK;
}

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
            a.row_solve = true;
            // below the top level u is the zero guess the level above has just stored (:132): not loaded, and the two pre-smoothing
            // sweeps of every level of the sub-hierarchy are one pass (option mg_zero_fuse; mg_zero_guess = 0: u is read like any field)
            a.zfuse = fpr_opt(ctx, "mg_zero_fuse", 1) != 0 ? ((!top && fpr_opt(ctx, "mg_zero_guess", 1) != 0) ? 2 : 1) : 0;
            a.prof = nullptr;   // tools/exp_mg_small_prof.py: device address of 32 int64, or 0
            if (int rc = fprx_cycle_finish_flush(ctx)) return rc;
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
    if (!top) {
        bool taken = false;
        if (int rc = mid_path(ctx, A, d, u, rhs, h, c, tol, css, solver, apply_BCs, &taken)) return rc;
        if (taken) return FPR_OK;
    }

    if ((nx < ny ? nx : ny) > css) {  // multigrid.jl:121
        if (d + 1 >= A.size() || !L.res_c) return fpr_fail(ctx, FPR_ERR_INVALID, "level arena exhausted");
        const int nxc = 1 + (nx - 1) / 2, nyc = 1 + (ny - 1) / 2;
        // a finish handed over by the loop (fprx_cycle_finish_defer) rides on the restricting two-sweep pass below; every other way
        // down launches it first
        const bool carry_fin = ctx->fin.partials && !top && fpr_opt(ctx, "mg_multi", 1) == 1 && nx >= 64 && ny >= 16 &&
                               fpr_opt(ctx, "mg_fuse_restrict", 1) != 0;
        if (!carry_fin)
            if (int rc = fprx_cycle_finish_flush(ctx)) return rc;
        if (fpr_opt(ctx, "mg_multi", 1) == 1 && nx >= 64 && ny >= 16) {
            // temporal blocking, register-rolling march: each smoothing pair is ONE pass (u -> tmp, tmp -> u)
            const bool fuse_r = fpr_opt(ctx, "mg_fuse_restrict", 1) != 0;
            const int sw = 60, sw_r = 58;             // columns a 64-column strip owns (plain / restricting pass)
            const int nstrips = (nx + sw - 1) / sw;
            const int nstrips_r = (nx + sw_r - 1) / sw_r;  // strips of the restricting pre-smoothing pass
            int rpc = 64;    // enough chunks for >= ~16 waves per CU (4096 strip-chunks), chunks of at least 16 rows
            while (rpc > 16 && (long)nstrips * ((ny + rpc - 1) / rpc) < 4096) rpc >>= 1;
            rpc += rpc & 1;   // chunks start on even rows (k_smooth2_march_v2)
            const dim3 gm((nstrips + 3) / 4, (ny + rpc - 1) / rpc);
            const int npm = (int)(gm.x * gm.y);
            const bool fuse_p = fpr_opt(ctx, "mg_fuse_prolong", 1) != 0;
            // below the top level u is the zero guess the level above has just stored (:132): the pre-smoothing pass is told so
            // (bit 9) and does not read it (k_smooth2_march_v2; option mg_zero_guess = 0: read it like any field)
            const int uz = (!top && fpr_opt(ctx, "mg_zero_guess", 1) != 0) ? 512 : 0;
            if (fuse_r) {  // pre-smoothing pair + residual + injection + zero coarse guess in ONE pass (:124-132)
                const dim3 gr((nstrips_r + 3) / 4, (ny + rpc - 1) / rpc);
                const bool timed = top && fpr_ktimer_begin(ctx, FPR_KT_MG_PRE, s);
                if (carry_fin) {   // + one workgroup row: the finish of the cycle before (k_smooth2_march_v2)
                    const FprFinishArgs fa = ctx->fin;
                    ctx->fin = FprFinishArgs{};
                    k_smooth2_march_v2<false, false, true><<<dim3(gr.x, gr.y + 1), 256, 0, s>>>(u, rhs, L.tmp, nx, ny, C, _h2, fac, rpc, nstrips_r, nullptr, nullptr, apply_BCs | uz, L.res_c, L.corr_c, skp, fa);
                } else
                march_go<false, false, true>(ctx, gr, s, u, rhs, L.tmp, nx, ny, C, _h2, fac, rpc, nstrips_r, nullptr, nullptr, apply_BCs | uz, L.res_c, L.corr_c, skp);
                fpr_ktimer_end(ctx, timed, s);
            } else {
                march_go<false, false, false>(ctx, gm, s, u, rhs, L.tmp, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, nullptr, uz, nullptr, nullptr, skp);  // :124-125
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
                if (fuse_p) march_go<true, true, false>(ctx, gm, s, L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, ctx->partials, L.corr_c, apply_BCs, nullptr, nullptr, skp);
                else march_go<true, false, false>(ctx, gm, s, L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, ctx->partials, nullptr, 0, nullptr, nullptr, skp);
                fpr_ktimer_end(ctx, timed, s);
                FPR_CHECK_LAUNCH(ctx);
                if (skp) { if (int rc = fprx_cycle_finish(ctx, ctx->partials, npm, ctx->scalars, (double)nx * (double)ny, ctx->cyc_slot)) return rc; }
                else if (int rc = fprx_finish_sum(ctx, ctx->partials, npm, ctx->scalars, false, 0)) return rc;
                *rms_is_host = false;
            } else {
                if (fuse_p) march_go<false, true, false>(ctx, gm, s, L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, L.corr_c, apply_BCs, nullptr, nullptr, skp);
                else march_go<false, false, false>(ctx, gm, s, L.tmp, rhs, u, nx, ny, C, _h2, fac, rpc, nstrips, nullptr, nullptr, 0, nullptr, nullptr, skp);
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
    if (int rc = fprx_cycle_finish_flush(ctx)) return rc;
    const int iters = 20 * css;  // :149, :161
    if (solver == FPR_COARSE_JACOBI) {
        if (int rc = fprx_sumsq_scaled_dev(ctx, rhs, N, 1.0, ctx->scalars + 3, 0)) return rc;  // :150
        k_state_init<<<1, 1, 0, s>>>(ctx->state, ctx->scalars + 3, tol, (double)N, 0);
        FPR_CHECK_LAUNCH(ctx);
        ctx->state_h->done = 0; ctx->state_h->iters = 0; ctx->state_h->last_rms = 0.0;
        if (fpr_opt(ctx, "mg_multi", 1) && nx >= 32 && ny >= 32) {
            // groups of S fused sweeps per launch; the exit test is replayed per sweep on the device
            // k_jacobi_patch: 8 sweeps per launch on 32 x 32 regions (own tile 16 x 16, 256 threads), 1.25 us per sweep of the
            // 257^2 grid (2.9 us launch boundary + loads + norms per launch, 0.5 us per sweep).  16 sweeps on 48 x 48 regions
            // (576 threads) were measured at 1.40 us per sweep: a sweep of the larger region takes 1.0 us, twice as long.
            // Sweeps per launch of the patch kernel: 8 (own tile 16 x 16; default), or 7 (18 x 18) / 6 (20 x 20) on the same 32 x 32 region
            // (option mg_patch_sweeps).  257^2 is 17 x 17 = 289 tiles of 16 -- 33 of the 256 CUs hold two workgroups -- but 15 x 15 = 225
            // tiles of 18; measured (tools/exp_patch_sweeps.py) 12.1 us per launch of 8 sweeps, 10.4 of 7, 9.3 of 6: per sweep 1.51 / 1.49 /
            // 1.55 us, no gain -- a launch is mostly its boundary, prologue and epilogue, not its sweeps.
            constexpr int S8 = 8, TX = 16, TY = 16;
            constexpr int PP = 32;
            const bool patch = fpr_opt(ctx, "mg_patch", 1) != 0;
            int PS;
            {
                // default: 8 sweeps per group (own tiles of 16 x 16) -- or 7 (18 x 18) where that brings the number of workgroups from above
                // the number of compute units to below it: the persistent kernels hand tiles from neighbour to neighbour, and a compute unit
                // that holds two workgroups sets the pace for everybody (257^2: 289 -> 225 workgroups, 0.68 -> 0.54 us per sweep)
                if (ctx->ncu <= 0) {
                    int v = 0;
                    ctx->ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ? v : 256;
                }
                auto tiles = [&](int ps) { const int t = PP - 2 * ps; return (long)((nx + t - 1) / t) * ((ny + t - 1) / t); };
                PS = (patch && tiles(8) > ctx->ncu && tiles(7) <= ctx->ncu && fpr_opt(ctx, "mg_jacobi_persist", 1) != 0) ? 7 : 8;
            }
            const int S = patch ? PS : S8;
            const int TXp = patch ? PP - 2 * PS : TX, TYp = patch ? PP - 2 * PS : TY;
            const dim3 gm((nx + TXp - 1) / TXp, (ny + TYp - 1) / TYp);
            const int nblk = (int)(gm.x * gm.y);
            if ((size_t)nblk * S > (size_t)FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
            const int Sg = S;          // sweeps per launch
            const int groups = (iters + Sg - 1) / Sg;
            int g_resume = 0;          // the plain loop below starts at this group (> 0: behind a persistent launch that gave up)
            // ---- the persistent form: launches of up to 16 groups of 8 sweeps with neighbour-to-neighbour hand-offs inside
            //      (mg_jacobi_persistent.hpp); the plain form below stays for A/B, for grids it does not fit and as the replay ----
            // (options mg_patch_sweeps = 7: groups of 7 sweeps on own tiles of 18 x 18 -- 225 workgroups for 257^2 instead of 289; mg_jacp_py = 1: 2 x 1 register
            //  patches, 512 threads, two waves per SIMD)
            // Option handoff_fences = 1 asks for hand-offs inside the HIP memory model: the data-tagged granules are not (a 16-byte sc1 access
            // being untorn is a property of the hardware, not of the model), so the coarse solve then runs as plain launches of k_jacobi_patch
            // -- kernel boundaries are its hand-offs (1.5 us per sweep against 0.54; flags + agent-scope release / acquire cost 1.21:
            // profiles/r6_handoff_fences.txt)
            constexpr int PYo = 1;          // rows of a thread's register patch: 2 x 1 patches, 512 threads, two waves per SIMD
            if (patch && (PS == 8 || PS == 7) && Sg == PS && fpr_opt(ctx, "mg_jacobi_persist", 1) != 0 && fpr_opt(ctx, "handoff_fences", 0) == 0 &&
                (size_t)N * 8 < 0x7fffffffu) {
                const int jnt = (PP / 2) * (PP / PYo);
                if (ctx->jacp_resident < 0 || ctx->jacp_resident_key != PS * 100 + PYo * 10 + 1) {
                    int per_cu = 0;
                    hipError_t oe;
                    oe = PS == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_jacobi_persist_tag<8, PP, 1>, jnt, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_jacobi_persist_tag<7, PP, 1>, jnt, 0);
                    const bool ok = oe == hipSuccess;
                    ctx->jacp_resident_key = PS * 100 + PYo * 10 + 1;
                    if (ctx->ncu <= 0) {
                        int v = 0;
                        ctx->ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ? v : 256;
                    }
                    ctx->jacp_resident = ok ? per_cu * ctx->ncu : 0;    // workgroups the device holds at once
                }
                const int GMAX = 32;    // groups per launch: 256 sweeps, whose exit tests one workgroup replays behind the launch
                // ---- k_jacobi_persist_tag: a cell travels as a 16-byte {value, tag} granule, no flags, no drains ----
                {
                    CgWork w2;
                    if (nblk <= ctx->jacp_resident / 2 &&
                        (size_t)2 * GMAX * PS * nblk <= (size_t)FPR_MAX_PARTIALS - 2048 && (size_t)N * 16 < 0x7fffffffu && cg_work(ctx, 3 * N, &w2) == FPR_OK) {
                        // five rotating granule buffers (2 N doubles each): a launch reads one, writes three, and leaves the input of the launch
                        // BEFORE it alone -- the exit test of that launch runs beside it (below) and may still ask for a replay from there
                        void* Gb[5];
                        for (int k = 0; k < 5; ++k) Gb[k] = w2.r + (size_t)2 * N * k;
                        double* P0 = w2.r + (size_t)10 * N;          // plain scratch (N doubles): the input of a launch that is replayed
                        // The exit tests of a launch (k_jacobi_check_groups: 8 us) run in line behind it (on a side stream beside the next launch they
                        // cost two cross-stream event waits per launch: 3.63 against 3.37 ms per five-level V-cycle, EXPERIMENTS 13.5); a launch that
                        // starts before the tests of its predecessor have found the exit only wastes itself (it reads `done` at its start).
                        hipStream_t sc = s;
                        const size_t phalf = (size_t)GMAX * PS * nblk;
                        int* flags = reinterpret_cast<int*>(ctx->partials + FPR_MAX_PARTIALS - 1024);
                        int* abort_flag = flags + 2040;
                        int* counter = flags + 2041;
                        double* gsums = ctx->partials + FPR_MAX_PARTIALS - 2048;
                        FPR_HIP(ctx, hipMemsetAsync(flags, 0, 2048 * sizeof(int), s));
                        const long long tag_base = fpr_next_epoch(ctx) << 32;   // tags of one solve never meet another solve's
                        struct RecT { int x, w[3], g0, G; };             // x = -1: the plain field u (first launch)
                        std::vector<RecT> recs;
                        int cur = -1, prevx = -1, gdone = 0, poll_after = 1, since_poll = 0;
                        const unsigned ugrid = (unsigned)((N + 255) / 256);
                        auto state_now = [&]() -> int { return read_state(ctx); };      // the host's view of the solve: behind the tests enqueued so far
                        while (gdone < groups) {
                            const int G = groups - gdone < GMAX ? groups - gdone : GMAX;
                            RecT r;
                            r.x = cur; r.g0 = gdone; r.G = G;
                            for (int k = 0, q = 0; k < 5 && q < 3; ++k) if (k != cur && k != prevx) r.w[q++] = k;
                            double* parts = ctx->partials + (recs.size() & 1) * phalf;
                            // (this launch overwrites the partial sums of the launch before the last one: its tests are through)
                            JacTagArgs a;
                            a.X = u; a.Xg = cur >= 0 ? Gb[cur] : nullptr; a.x_tagged = cur >= 0 ? 1 : 0;
                            for (int q = 0; q < 3; ++q) a.W[q] = Gb[r.w[q]];
                            a.rhs = rhs; a.nx = nx; a.ny = ny; a.C = C; a._h2 = _h2; a.fac = fac;
                            a.ngroups = G;
                            a.nsw_last = (gdone + G == groups) ? iters - (groups - 1) * PS : PS;
                            a.partials = parts; a.abort_flag = abort_flag; a.state = ctx->state;
                            a.g0 = gdone; a.tag_base = tag_base;
                            a.prof = nullptr;
                            if ((long)recs.size() + 1 == fpr_opt(ctx, "mg_jacobi_persist_test_abort", 0))      // (test hook: this launch "times out")
                                FPR_HIP(ctx, hipMemsetAsync(abort_flag, 1, 1, s));
                            const bool timed = fpr_ktimer_begin(ctx, FPR_KT_MG_PATCH, s);
                            if (PS == 8) k_jacobi_persist_tag<8, PP, 1><<<gm, jnt, 0, s>>>(a);
                            else k_jacobi_persist_tag<7, PP, 1><<<gm, jnt, 0, s>>>(a);
                            fpr_ktimer_end(ctx, timed, s);
                            k_jacobi_check_groups<<<G, 256, 0, sc>>>(ctx->state, parts, nblk, G, PS, a.nsw_last, (double)N, gdone, abort_flag, gsums, counter);
                            FPR_CHECK_LAUNCH(ctx);
                            recs.push_back(r);
                            prevx = cur;
                            cur = r.w[G % 3];                  // the last group (index G - 1) wrote W[G % 3]
                            gdone += G;
                            if (++since_poll >= poll_after || gdone >= groups) {
                                since_poll = 0;
                                if (poll_after < 8) poll_after *= 2;
                                if (int rc = state_now()) return rc;
                                if (ctx->state_h->done) break;
                            }
                        }
                        // (the compute stream is behind every test now: state_now() made it wait for the side stream)
                        // the input of launch `r` as plain doubles in `dst`
                        auto input_to = [&](const RecT& r, double* dst) -> int {
                            if (r.x < 0) { if (dst != u) FPR_HIP(ctx, hipMemcpyAsync(dst, u, N * sizeof(double), hipMemcpyDeviceToDevice, s)); }
                            else k_jacp_untag<<<ugrid, 256, 0, s>>>(Gb[r.x], dst, N);
                            FPR_CHECK_LAUNCH(ctx);
                            return FPR_OK;
                        };
                        if (ctx->state_h->done < 0) {
                            // a wait timed out (see the flag form below): resume with the plain launches from the input of the launch that gave up
                            const RecT* r = nullptr;
                            for (const RecT& q : recs) if (q.g0 == ctx->state_h->group) r = &q;
                            if (!r) return fpr_fail(ctx, FPR_ERR_HIP, "k_jacobi_persist_tag: a hand-off timed out and the launch that gave up is unknown");
                            ctx->jacp_resident = 0;
                            ctx->options["mg_jacobi_persist_timeouts"] = fpr_opt(ctx, "mg_jacobi_persist_timeouts", 0) + 1;
                            FPR_HIP(ctx, hipMemsetAsync(flags, 0, 2048 * sizeof(int), s));
                            if (int rc = input_to(*r, (r->g0 & 1) ? L.tmp : u)) return rc;      // the plain loop reads group gi from u (even) / L.tmp (odd)
                            k_state_resume<<<1, 1, 0, s>>>(ctx->state);
                            FPR_CHECK_LAUNCH(ctx);
                            ctx->state_h->done = 0;
                            g_resume = r->g0;
                            goto plain_jacobi_groups;
                        }
                        if (ctx->state_h->done) {
                            // the criterion was met inside launch `r`: replay the exact number of sweeps from that launch's input with the ordinary launches
                            const int gs = ctx->state_h->group, redo = ctx->state_h->redo;
                            const RecT* r = nullptr;
                            for (const RecT& q : recs) if (gs >= q.g0 && gs < q.g0 + q.G) r = &q;
                            if (!r) return fpr_fail(ctx, FPR_ERR_INVALID, "k_jacobi_persist_tag: exit group outside the launches");
                            if (int rc = input_to(*r, P0)) return rc;
                            int left = (gs - r->g0) * PS + redo;
                            const double* in = P0;
                            int o = 0;
                            while (left > 0) {
                                const int m = left < PS ? left : PS;
                                double* out = (o & 1) ? u : L.tmp;
                                if (PS == 8) k_jacobi_patch<8, PP, false, false><<<gm, (PP / 2) * (PP / 2), 0, s>>>(in, rhs, out, nx, ny, C, _h2, fac, m, nullptr, nullptr, nullptr, 0, 0, 0.0);
                                else k_jacobi_patch<7, PP, false, false><<<gm, (PP / 2) * (PP / 2), 0, s>>>(in, rhs, out, nx, ny, C, _h2, fac, m, nullptr, nullptr, nullptr, 0, 0, 0.0);
                                in = out; ++o; left -= m;
                            }
                            FPR_CHECK_LAUNCH(ctx);
                            if (in != u) FPR_HIP(ctx, hipMemcpyAsync(u, in, N * sizeof(double), hipMemcpyDeviceToDevice, s));
                        } else {
                            k_jacp_untag<<<ugrid, 256, 0, s>>>(Gb[cur], u, N);
                            FPR_CHECK_LAUNCH(ctx);
                        }
                        ctx->last_coarse_iters += ctx->state_h->iters;
                        *rms_out_host = ctx->state_h->last_rms;
                        *rms_is_host = true;
                        return FPR_OK;
                    }
                }
            }
        plain_jacobi_groups:
            double* a = (g_resume & 1) ? L.tmp : u;
            double* b = (g_resume & 1) ? u : L.tmp;
            int gdone = g_resume;
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
                        const bool timed = nsw == Sg && fpr_ktimer_begin(ctx, FPR_KT_MG_PATCH, s);
#define FPR_PATCH_GO(PSV)                                                                                                                   \
    k_jacobi_patch<PSV, PP, true, true><<<gm, (PP / 2) * (PP / 2), 0, s>>>(a, rhs, b, nx, ny, C, _h2, fac, nsw, slot, ctx->state, pslot, \
                                                                           pending ? Sg : 0, gi - 1, (double)N)
                        if (PS == 8) FPR_PATCH_GO(8); else if (PS == 7) FPR_PATCH_GO(7); else FPR_PATCH_GO(6);
#undef FPR_PATCH_GO
                        fpr_ktimer_end(ctx, timed, s);
                        if (gi == gend - 1)  // last group of the chunk: stand-alone check before the host polls
                            k_jacobi_check_multi<<<1, 256, 0, s>>>(ctx->state, slot, nblk, nsw, (double)N, gi);
                    } else {
                        k_sweep2d_multi<S8, TX, TY, 2, true><<<gm, 256, 0, s>>>(a, rhs, b, nx, ny, C, _h2, fac, nsw, ctx->partials, ctx->state);
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
                    if (patch) {
#define FPR_PATCH_REDO(PSV) \
    k_jacobi_patch<PSV, PP, false, false><<<gm, (PP / 2) * (PP / 2), 0, s>>>(in, rhs, out, nx, ny, C, _h2, fac, redo, nullptr, nullptr, nullptr, 0, 0, 0.0)
                        if (PS == 8) FPR_PATCH_REDO(8); else if (PS == 7) FPR_PATCH_REDO(7); else FPR_PATCH_REDO(6);
#undef FPR_PATCH_REDO
                    }
                    else k_sweep2d_multi<S8, TX, TY, 0, false><<<gm, 256, 0, s>>>(in, rhs, out, nx, ny, C, _h2, fac, redo, nullptr, nullptr);
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
    int nx, ny, rpc, rpc_s, nstrips, nstrips_r, nstrips_s, seam_cols;
    dim3 gm, gr, gs;
    double C, _h2, fac;
};

static TopGeom top_geom(fpr_ctx* ctx, int nx, int ny, double h, double c)
{
    TopGeom g;
    g.nx = nx; g.ny = ny;
    g.C = 4.0 + c * (h * h); g._h2 = 1 / (h * h);
    g.fac = (4.0 / 5.0) * ((h * h) / (4.0 + c * (h * h)));
    g.nstrips = (nx + 59) / 60;      // as vcycle_level (one column per lane)
    g.nstrips_r = (nx + 57) / 58;
    // seam pass: one column per lane (k_seam_march_v2: 54 owned of 64) or two (k_seam_march_v3: 118 of 128).  Two columns per lane halve
    // the number of strips: taken where one workgroup per CU still leaves chunks of 128 rows or more (4097^2: 9 x 27 workgroups, chunks of
    // 152 rows: 112 against 118 us; 2049^2 would get 43-row chunks: 38 against 33 us)
    {
        if (ctx->ncu <= 0) {
            int v = 0;
            ctx->ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ? v : 256;
        }
        const int gx3 = ((nx + 117) / 118 + 3) / 4;
        const int chunks3 = (int)(0.95 * ctx->ncu) / gx3;
        g.seam_cols = (chunks3 >= 1 && ny / chunks3 >= 128 && nx >= 256) ? 2 : 1;
    }
    g.nstrips_s = g.seam_cols == 2 ? (nx + 117) / 118 : (nx + 53) / 54;
    int rpc = 64;
    while (rpc > 16 && (long)g.nstrips * ((ny + rpc - 1) / rpc) < 4096) rpc >>= 1;
    rpc += rpc & 1;   // chunks start on even rows (k_smooth2_march_v2)
    g.rpc = rpc;
    // A seam workgroup works for most of the pass: the chunks are made as tall as a single round allows (all workgroups resident at once,
    // ~95 % of the slots) -- the 9 overlap rows weigh less and no second, half-empty round trails.  One column per lane (108 VGPRs): TWO
    // workgroups per CU (what counts is that no CU gets one workgroup more than the others: 25 chunks of 164 rows 116 us where 50 of 82 rows
    // take 121 us; counts just above a multiple of the CU count are the slow ones -- profiles/r4_mg_seam_chunks.txt); two columns per lane
    // (twice the loads in flight per wave): ONE workgroup per CU keeps the chunks as tall.
    int rs;
    {
        const int gx = (g.nstrips_s + 3) / 4;
        const int wg_per_cu = g.seam_cols == 2 ? 1 : 2;
        int chunks = (int)(0.95 * wg_per_cu * ctx->ncu) / gx;
        if (chunks < 1) chunks = 1;
        rs = (ny + chunks - 1) / chunks;
        if (rs < 32) rs = 32;
    }
    {   // rows are addressed relative to the first row of a chunk with a 32-bit byte offset
        const long cap = (long)(0x7fffffffL / ((long)nx * 8)) - 16;
        if (rs > cap) rs = (int)(cap > 16 ? cap : 16);
    }
    rs += rs & 1;   // k_seam_march_v2: chunks start on even rows (the parity of a row of its unrolled loop is a constant) --
                    // after the cap, which may be odd (one row more than the cap stays far inside the 16 rows of slack)
    g.rpc_s = rs;
    g.gm = dim3((g.nstrips + 3) / 4, (ny + rpc - 1) / rpc);
    g.gr = dim3((g.nstrips_r + 3) / 4, (ny + rpc - 1) / rpc);
    g.gs = dim3((g.nstrips_s + 3) / 4, (ny + g.rpc_s - 1) / g.rpc_s);
    return g;
}

// pre-smoothing pair + residual + injection + zero coarse guess (:124-132): uin -> out, L.res_c, corr_zero
static int top_pre(fpr_ctx* ctx, const TopGeom& g, const double* uin, const double* rhs, double* out, double* res_c,
                   double* corr_zero, int apply_BCs, const int* skp, double* fsq_partials = nullptr)
{
    hipStream_t s = ctx->stream[0];
    const bool timed = fpr_ktimer_begin(ctx, FPR_KT_MG_PRE, s);
    if (fsq_partials)   // the first pass of a solve also leaves sum(f.^2) as block partials (one per workgroup)
        k_smooth2_march_v2<false, false, true, true><<<g.gr, 256, 0, s>>>(uin, rhs, out, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc, g.nstrips_r, fsq_partials,
                                                                          nullptr, apply_BCs, res_c, corr_zero, skp, FprFinishArgs{});
    else
    march_go<false, false, true>(ctx, g.gr, s, uin, rhs, out, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc, g.nstrips_r, nullptr,
                                                             nullptr, apply_BCs, res_c, corr_zero, skp);   // (:355-357 included)
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
    if (norm) march_go<true, true, false>(ctx, g.gm, s, X, rhs, uout, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc, g.nstrips, ctx->partials, corr, apply_BCs, nullptr, nullptr, skp);
    else march_go<false, true, false>(ctx, g.gm, s, X, rhs, uout, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc, g.nstrips, nullptr, corr, apply_BCs, nullptr, nullptr, skp);
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
    if (g.seam_cols == 2) {     // (six rows in flight per half: 112 us at 4097^2; four: 114-115; twelve: 115)
        if (apply_BCs) k_seam_march_v3<true, 4, 2><<<g.gs, 256, 0, s>>>(X, rhs, Y, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc_s, g.nstrips_s, ctx->partials, corr, res_c, corr_zero, skp);
        else k_seam_march_v3<false, 6, 2><<<g.gs, 256, 0, s>>>(X, rhs, Y, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc_s, g.nstrips_s, ctx->partials, corr, res_c, corr_zero, skp);
    } else {
        if (apply_BCs) k_seam_march_v2<true><<<g.gs, 256, 0, s>>>(X, rhs, Y, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc_s, g.nstrips_s, ctx->partials, corr, res_c, corr_zero, skp);
        else k_seam_march_v2<false><<<g.gs, 256, 0, s>>>(X, rhs, Y, g.nx, g.ny, g.C, g._h2, g.fac, g.rpc_s, g.nstrips_s, ctx->partials, corr, res_c, corr_zero, skp);
    }
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
    FPR_HIP(ctx, hipSetDevice(ctx->device));   // the calling thread may be a worker that has never selected the context's device
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
    double kept[16];   // the first norms of this solve (for the next solve of the same system, mg_hist)
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
        // an error return in the middle of the loop leaves launches enqueued AHEAD of the host: drain them before the caller
        // gets control back (they would otherwise still write u after the call has returned); outputs are undefined then
        struct Guard {
            fpr_ctx* c;
            bool ok;
            ~Guard() { c->cyc_skip = nullptr; c->fin = FprFinishArgs{}; if (!ok) hipStreamSynchronize(c->stream[0]); }   // (a finish still handed over belongs to a failed loop)
        } guard{ctx, false};
        // rms(f) and the threshold tol * rms(f) stay on the device too (same operations as on the host): no round trip
        // before the first cycle; the host learns both from the first record
        ctx->cyc_skip = &ctx->cyc->stop;
        const int* skp = ctx->cyc_skip;
        int enq = 0;
        const bool seam_path = fpr_opt(ctx, "mg_seam", 1) != 0;
        // sum(f.^2) rides on the first pass over the finest grid where that is the two-sweep march (k_smooth2_march_v2<..., FSQ>): the cycle
        // state is reset now, f_rms and the threshold follow behind that pass (fprx_cycle_init_from); otherwise a pass over f of its own
        bool fsq = seam_path && fpr_opt(ctx, "mg_fold_fsq", FPR_FOLD_FSQ_DEFAULT);
        if (!seam_path)
            if (int rc = fprx_cycle_init(ctx, f, N, tol)) return rc;
        if (seam_path) {
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
            const int npm = (int)(g.gm.x * g.gm.y), nps = (int)(g.gs.x * g.gs.y), npr = (int)(g.gr.x * g.gr.y);
            if (npm > FPR_MAX_PARTIALS || nps > FPR_MAX_PARTIALS) return fpr_fail(ctx, FPR_ERR_INVALID, "grid too large for partial buffer");
            if (npr > FPR_MAX_PARTIALS) fsq = false;
            if (int rc = fsq ? fprx_cycle_reset(ctx, tol) : fprx_cycle_init(ctx, f, N, tol)) return rc;
            struct Unit { bool seam; const double* X; int p; } units[FPR_CYC_SLOTS];
            double* X = nullptr;
            int p = 0;
            bool need_head = true;
            double r_prev = 0.0, r_last = 0.0;   // the last two norms the host has seen
            // the last solve with these arrays (none remembered: prev_cycles = 0): its norms relative to its threshold
            int prev_cycles = 0;
            double prev_rel[16];
            if (fpr_opt(ctx, "mg_seam_history", 1))
                for (const auto& e : ctx->mg_hist)
                    if (e.u == u && e.f == f && e.nx == nx && e.ny == ny && e.cycles > 0) {
                        prev_cycles = e.cycles < 16 ? e.cycles : 16;
                        for (int q = 0; q < prev_cycles; ++q) prev_rel[q] = e.rel[q];
                    }
            // norm after cycle q (1-based) of the remembered solve over its threshold; beyond its end: its last rate goes on
            auto prel = [&](int q) -> double {
                if (q <= prev_cycles) return prev_rel[q - 1];
                const double rate = prev_cycles >= 2 ? prev_rel[prev_cycles - 1] / prev_rel[prev_cycles - 2] : 0.5;
                double v = prev_rel[prev_cycles - 1];
                for (int i = prev_cycles; i < q; ++i) v *= rate;
                return v;
            };
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
                    if (int rc = top_pre(ctx, g, u, f, L.tmp, L.res_c, corr[0], apply_BCs, skp, fsq ? ctx->partials : nullptr)) return rc;
                    if (fsq) {   // (the first pass of the solve only)
                        if (int rc = fprx_cycle_init_from(ctx, ctx->partials, npr, N, tol)) return rc;
                        fsq = false;
                    }
                    if (int rc = lower(0)) return rc;
                    X = L.tmp; p = 0; need_head = false;
                }
                bool last = (k == niters);
                // mg_seam_predict (a test hook): 1 = extrapolate (default); 0 = never (every cycle but the niters-th ends in a seam, the
                // last one is replayed); 2 = odd cycles end in a plain pass (exercises the restart after a wrong guess)
                const long predict = fpr_opt(ctx, "mg_seam_predict", 1);
                if (predict == 2 && (k & 1)) last = true;
                if (!last && predict == 1 && prev_cycles > 0 && (n == 0 || (r_last > 0.0 && tolf > 0.0))) {
                    // A time stepper solves the same systems step after step and their convergence histories hardly move, while the
                    // rate WITHIN a solve does (the T solve of part2.jl:221: 0.06, 0.28, 0.34, 0.34 ...): take the reduction from
                    // the last norm seen (cycle n) to cycle k from the remembered solve.  Before the first norm: its own history.
                    const double pred_rel = n == 0 ? prel(k) : (r_last / tolf) * (prel(k) / prel(n));
                    last = pred_rel < 1.0;
                } else if (!last && predict == 1 && n >= 2 && r_last < r_prev && r_last > 0.0) {   // geometric extrapolation of the norm to cycle k
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
                    // (the finish of cycle k travels with cycle k+1's first pass below the finest level where that is a two-sweep march)
                    if (int rc = fprx_cycle_finish_defer(ctx, ctx->partials, nps, ctx->scalars, (double)N, slot)) return rc;
                    if (int rc = lower(1 - p)) return rc;   // cycle k+1 below the finest level
                    if (ctx->fin.partials) return fpr_fail(ctx, FPR_ERR_INVALID, "internal: the finish of cycle %d was not launched", k);
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
                if (n < 16) kept[n] = r_rms;
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
            if (n < 16) kept[n] = r_rms;
            ++n;
            if (rec.stop) break;  // :70 (taken on the device)
        }
        }
        guard.ok = true;
    } else
    for (int iter = 1; iter <= niters; ++iter) {
        if (apply_BCs)
            if (int rc = fpr_bc2d(ctx, u, nx, ny)) return rc;  // :60-62
        if (int rc = vcycle_run(ctx, u, f, h, c, tol, coarse_solve_size, coarse_solver, apply_BCs, nx, ny, &r_rms)) return rc;
        if (history_host) history_host[n] = r_rms;
        if (n < 16) kept[n] = r_rms;
        ++n;
        if (r_rms < tolf) break;  // :70
    }
    {   // remember the cycle count for the next solve of the same system (round-robin over 8 entries)
        int at = -1;
        for (int q = 0; q < 8; ++q)
            if (ctx->mg_hist[q].u == u && ctx->mg_hist[q].f == f && ctx->mg_hist[q].nx == nx && ctx->mg_hist[q].ny == ny) at = q;
        if (at < 0) { at = ctx->mg_hist_next; ctx->mg_hist_next = (ctx->mg_hist_next + 1) & 7; }
        fpr_ctx::MgHist& e = ctx->mg_hist[at];
        e.u = u; e.f = f; e.nx = nx; e.ny = ny;
        e.cycles = tolf > 0.0 ? (n < 16 ? n : 16) : 0;
        for (int q = 0; q < e.cycles; ++q) e.rel[q] = kept[q] / tolf;
    }
    if (rms_host) *rms_host = r_rms;
    if (ncycles_host) *ncycles_host = n;
    if (frms_host) *frms_host = f_rms;
    if (converged_host) *converged_host = !(r_rms > tolf);  // :78-80 @warn condition
    return FPR_OK;
}

extern "C" long fpr_last_coarse_iters(fpr_ctx* ctx) { return ctx ? ctx->last_coarse_iters : 0; }
