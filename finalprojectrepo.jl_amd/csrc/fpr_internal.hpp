// fpr_internal.hpp -- context, error plumbing and device-side reduction helpers shared by the
// gfx950 translation units of libfpr_hip.so.  Not part of the public ABI (include/fpr.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fpr.h"

constexpr int FPR_WAVE = 64;            // gfx950 wavefront
constexpr int FPR_MAX_PARTIALS = 1 << 19;  // per-slot block partials (doubles)
constexpr int FPR_CORE_PARTIALS = 1 << 16; // per list of a core launch between ranks (fpr_diffusion3d_step2_halo)

// device-side solver state shared by the coarse Jacobi / CG kernels (one per context)
struct FprSolveState {
    int done;        // 1 once the stopping criterion was met
    int iters;       // iterations executed (including the one that met the criterion)
    int acc_iters;   // coarse-solver iterations accumulated over one V-cycle (k_mg_small adds to it)
    int redo;        // multi-sweep Jacobi: sweeps of group `group` that must be applied (exit inside a group)
    int group;       // multi-sweep Jacobi: group in which the stopping criterion was met
    int pad;
    double last_rms; // r_rms of the last executed iteration
    double thresh;   // tol * rms(rhs)   (Jacobi)  or  tol * ||b||  (CG)
    double rho, rho_old, alpha, beta, pq;  // CG scalars
    double rho2[2];  // CG: rho double-buffered by iteration parity (fused 3-launch iteration)
};

// MGsolve with cycles enqueued ahead of the host's convergence check (multigrid2d.hip, fpr_mgsolve2d): the decision
// "r_rms < tol * f_rms" (multigrid.jl:70) is taken on the device when a cycle finishes, and every launch of a later cycle
// returns at once when it finds `stop` set -- so a cycle enqueued before the host has seen the previous norm changes
// nothing if the loop had already ended.
struct FprCycleCtl {
    int stop;          // 1 once a finished cycle met the criterion
    int ncycles;       // cycles executed
    int coarse_iters;  // coarse-solver iterations of all executed cycles
    int seq;           // host record only: = ncycles, written LAST (the host polls it; 0 = not yet reported)
    double tolf;       // tol * f_rms
    double rms;        // r_rms of the last executed cycle
    double frms;       // f_rms (multigrid.jl:53), computed on the device by k_cycle_init
};
constexpr int FPR_CYC_SLOTS = 8;
// what the finish of a cycle needs (k_cycle_finish, or the extra workgroup row of the pass that follows it: fpr_cycle_finish_body)
struct FprFinishArgs {
    const double* partials;       // null = nothing to finish
    int n;
    double* out;
    double npoints;
    FprCycleCtl* ctl;
    const FprSolveState* st;
    FprCycleCtl* rec_host;
};

struct FprLevel {  // one multigrid level's scratch (role of prealloc_dict, multigrid.jl:25-38)
    int nx = 0, ny = 0;
    double* tmp = nullptr;    // ping-pong partner of u on this level
    double* res_c = nullptr;  // restricted residual = rhs of the next coarser level
    double* corr_c = nullptr; // coarse correction = u of the next coarser level
    // finest level of fpr_mgsolve2d's seam passes only (allocated on first use): a second ping-pong partner, so that u itself
    // is never scratch, and a second coarse correction, so that one pass can read cycle k's while it zeroes cycle k+1's
    double* tmp2 = nullptr;
    double* corr_c2 = nullptr;
    bool own_tmp = true, own_tmp2 = true;   // false: the caller's buffer (fpr_mg_arena_provide), not freed by the library
    bool own_coarse = true;                 // the same for res_c, corr_c, corr_c2 (fpr_mg_arena_provide_coarse)
    double* coarse_own[3] = {nullptr, nullptr, nullptr};   // the library's own three while the caller's stand in (they come back as they were)
};

#ifndef FPR_FOLD_FINISH_DEFAULT
#define FPR_FOLD_FINISH_DEFAULT 1  // the finish of cycle k (norm, exit test, record) runs in an extra workgroup row of cycle k+1's first pass below the finest
                                   // level instead of a launch of its own between the two (option mg_fold_finish)
#endif
#ifndef FPR_FOLD_FSQ_DEFAULT
#define FPR_FOLD_FSQ_DEFAULT 1     // sum(f.^2) (f_rms, multigrid.jl:53) as block partials of the solve's first pass over the finest grid instead of a pass of
                                   // its own over f (option mg_fold_fsq)
#endif

struct FprGrid {  // implicit global grid of the decomposed diffusion path (role of ImplicitGlobalGrid's global state)
    bool on = false;
    int n[3] = {0, 0, 0};        // local array size, halos included
    int dims[3] = {1, 1, 1}, coords[3] = {0, 0, 0}, periods[3] = {0, 0, 0};
    int nb[6] = {-1, -1, -1, -1, -1, -1};   // neighbour rank per face (2*dim + side), -1 = none
    double* sendbuf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // packed x / y planes
    double* recvbuf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double* stage = nullptr;     // gather!: one local array (rank 0)
};

struct fpr_ctx {
    int device = 0;
    void* comm = nullptr;        // ncclComm_t (comm.hip), or the FprHosted of a hosted transport; nullptr = single rank
    bool comm_hosted = false;    // fpr_comm_init_hosted: bytes travel through the host's callbacks instead of RCCL (rehearsals, tests)
    int comm_rank = 0, comm_size = 1;
    FprGrid grid;
    // 0 compute, 1 comm, 2 core (= 0 unless fpr_reserve_comm_cus split the device: then 1 is a library-owned stream whose
    // kernels run on `comm_cus` compute units only and 2 one whose kernels run on all the others)
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};
    bool own_stream[2] = {false, false};
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipStream_t caller_comm = nullptr;   // the comm stream given at creation, while a masked one stands in for it
    hipStream_t masked[2] = {nullptr, nullptr};   // library-owned CU-masked streams (comm, core)
    int comm_cus = 0;                  // units of the comm stream (0: device not split)
    int comm_cus_asked = 0;            // what fpr_reserve_comm_cus was asked for (it may round up)
    bool pair_pending = false;         // a fused pair of fpr_diffusion3d_step2_halo left on the core / comm streams (join = 0)
    double* core_partials = nullptr;   // 2 parities x 3 lists x FPR_CORE_PARTIALS: the core launch's partials (pairs use 2 lists), finished on the comm stream
    double* shell3 = nullptr;          // fpr_diffusion3d_step3_halo: levels 1 and 2 of the z-shells (6 slabs of 6 planes: L1 low / high, L2 low / high, residuals)
    size_t shell3_doubles = 0;
    int pair_parity = 0;
    // fpr_mgsolve2d: V-cycles the last solve with the same (u, f, nx, ny) took -- a time stepper solves the same systems step after
    // step (part2.jl:187,221,226) and their convergence histories hardly move: the seam pass's guess of the last cycle takes the
    // reduction from cycle n to cycle k from there instead of assuming the last rate seen
    struct MgHist { const double* u; const double* f; int nx, ny, cycles; double rel[16]; } mg_hist[8] = {};   // rel[q] = norm after cycle q+1 / (tol * rms(f))
    int mg_hist_next = 0;
    hipEvent_t ns_ev = nullptr;        // fpr_ns_step2d: orders this context's compute stream against the other context's
    void* ns_worker = nullptr;         // fpr_ns_step2d: the host thread that runs the T solve beside the W solve (navier2d.hip)
    int ns_t_cycles = 0, ns_w_cycles = 0, ns_s_cycles = 0;   // fpr_ns_run2d: V-cycles of the T / W / S solves of the previous step
    int ns_seq = 0;                    // fpr_ns_velocity_max2d: sequence number of the report in pinned host memory
    double* xstrips = nullptr;         // compact strips of the columns next to x-faces with a neighbour (diffusion3d_xstrip.hpp)
    size_t xstrips_doubles = 0;
    const double* xs_field = nullptr;  // the field whose columns next to the x-faces the level-0 strips hold (Hout of the last pair) ...
    const double* xs_ht = nullptr;     // ... and the Ht whose columns the HT strips hold; valid only while that pair is pending
    int xs_faces = 0, xs_n[3] = {0, 0, 0};
    int* tickets = nullptr;            // 9 counters of the ticketed reserved form (Diff3Args2::ticket), zero between launches
    unsigned* reserved_map = nullptr;  // 2048 bits, one per (XCC, SE, SH, CU) key: the comm stream's compute units (fpr_reserve_comm_cus), found by a probe launch
    bool core_unmasked = false;        // the core stream has every unit; workgroups of a core launch that land on a comm unit leave at once
    int cgp_resident64 = -1;      // the same for the 64-workgroup geometry of k_cg_persistent
    int jacp_resident = -1;       // k_jacobi_persist: workgroups of 256 threads the device holds at once (-1 = not asked yet)
    long long jacp_epoch = 0;          // k_jacobi_persist_tag: solves so far (upper half of every granule's tag)
    int jacp_resident_key = 0;         // (sweeps per group * 10 + patch rows) the occupancy answer above belongs to
    double* partials = nullptr;     // FPR_MAX_PARTIALS doubles: block partial sums (slot 0)
    double* partials2 = nullptr;    // second slot (comm stream / second reduction of a kernel)
    int ncu = 0;                       // compute units of the device (queried on first use)
    double* scalars = nullptr;      // 64 device doubles for results of reductions
    FprSolveState* state = nullptr; // device
    FprSolveState* state_h = nullptr;  // pinned host mirror
    double* host_scalars = nullptr;    // pinned, 64 doubles
    FprCycleCtl* cyc = nullptr;        // device
    FprCycleCtl* cyc_h = nullptr;      // pinned, FPR_CYC_SLOTS records (one per cycle in flight)
    const int* cyc_skip = nullptr;     // &cyc->stop while fpr_mgsolve2d runs cycles ahead, else null (launches unconditional)
    int cyc_slot = 0;                  // record slot (of cyc_h) the cycle being enqueued reports into
    FprFinishArgs fin = {};            // fin.partials != null: a finish handed to the next pass below the finest level (fprx_cycle_finish_defer)
    std::map<std::pair<int, int>, std::vector<FprLevel>> arenas;  // keyed by finest (nx, ny)
    double* cg_buf = nullptr;          // CG work vectors (krylov.jl:59-62)
    size_t cg_cap = 0;                 // capacity of cg_buf in doubles
    std::map<std::string, long> options;
    long last_coarse_iters = 0;
    bool used_small = false;           // last V-cycle ran k_mg_small
    bool top_is_coarsest = false;      // ... and the whole problem was the coarsest level
    bool ktimer_on = false;            // fpr_kernel_timer
    std::vector<hipEvent_t> ktimer_ev; // pairs (start, stop)
    std::vector<int> ktimer_kind;      // kernel kind of each pair (FPR_KT_*)
    size_t ktimer_used = 0;
    char err[512] = {0};
};

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

inline int fpr_fail(fpr_ctx* ctx, int code, const char* fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define FPR_HIP(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fpr_fail((ctx), FPR_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,        \
                            hipGetErrorString(e_));                                                 \
    } while (0)

#define FPR_CHECK_LAUNCH(ctx) FPR_HIP(ctx, hipGetLastError())

#define FPR_REQUIRE(ctx, cond, msg)                                                                 \
    do {                                                                                            \
        if (!(cond)) return fpr_fail((ctx), FPR_ERR_INVALID, "invalid argument: %s", msg);          \
    } while (0)

// fpr_kernel_timer: one hipEvent pair around a launch, recorded on the stream the kernel is launched on
inline bool fpr_ktimer_begin(fpr_ctx* ctx, int kind, hipStream_t s)
{
    if (!ctx->ktimer_on || ctx->ktimer_used + 2 > ctx->ktimer_ev.size()) return false;
    if (hipEventRecord(ctx->ktimer_ev[ctx->ktimer_used], s) != hipSuccess) return false;
    ctx->ktimer_kind[ctx->ktimer_used / 2] = kind;
    return true;
}
inline void fpr_ktimer_end(fpr_ctx* ctx, bool timed, hipStream_t s)
{
    if (!timed) return;
    (void)hipEventRecord(ctx->ktimer_ev[ctx->ktimer_used + 1], s);
    ctx->ktimer_used += 2;
}

inline long fpr_opt(fpr_ctx* ctx, const char* key, long dflt)
{
    auto it = ctx->options.find(key);
    return it == ctx->options.end() ? dflt : it->second;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// Neighbour-lane moves by DPP wave shifts (gfx9 family: v_mov_b32_dpp wave_shr:1 / wave_shl:1): no LDS
// crossbar round trip as with ds_bpermute (__shfl_up/__shfl_down).  Edge lanes keep their own value,
// like __shfl_up / __shfl_down do.
__device__ __forceinline__ double fpr_lane_up1(double v)  // lane i <- lane i-1
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fpr_lane_down1(double v)  // lane i <- lane i+1
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// the same shifts with ZERO in the lane that has no source (lane 0 / lane 63) instead of the lane's own value: the
// destination needs no copy of the source first (two v_mov less per shift).  For kernels whose outermost lanes only feed.
__device__ __forceinline__ double fpr_lane_up1z(double v)  // lane i <- lane i-1, lane 0 <- 0
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fpr_lane_down1z(double v)  // lane i <- lane i+1, lane 63 <- 0
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// shifts inside a 16-lane DPP row (zero where no lane is the source); n is uniform, 1..15
template <int CTRL>
__device__ __forceinline__ double fpr_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N> __device__ __forceinline__ double fpr_row_ror(double v) { return fpr_dpp<0x120 + N>(v); }
// sum over the 64 lanes of a wave by DPP row shifts (an inclusive scan inside each 16-lane row, then the four row totals):
// ~25 instructions where a shuffle tree takes 12 LDS-crossbar round trips.  The same value in every lane.
__device__ __forceinline__ double fpr_wave_sum_all(double v)
{
    v += fpr_dpp<0x111>(v);
    v += fpr_dpp<0x112>(v);
    v += fpr_dpp<0x114>(v);
    v += fpr_dpp<0x118>(v);   // lane 15 of every row: the row's total
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return ((r0 + r1) + r2) + r3;
}

__device__ __forceinline__ double fpr_wave_sum(double v)
{
#pragma unroll
    for (int off = FPR_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, FPR_WAVE);
    return v;  // valid in lane 0
}

__device__ __forceinline__ double fpr_wave_max(double v)
{
#pragma unroll
    for (int off = FPR_WAVE / 2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, FPR_WAVE));
    return v;
}

// Block-wide sum in a fixed order (deterministic).  NT = threads per block (multiple of 64, <= 1024).
// Result valid in thread 0.  `red` = __shared__ double[16].
template <int NT>
__device__ __forceinline__ double fpr_block_sum(double v, double* red)
{
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    v = fpr_wave_sum(v);
    if constexpr (NT > FPR_WAVE) {
        if ((tid & (FPR_WAVE - 1)) == 0) red[tid / FPR_WAVE] = v;
        __syncthreads();
        if (tid == 0) {
            double s = red[0];
#pragma unroll
            for (int w = 1; w < NT / FPR_WAVE; ++w) s += red[w];
            v = s;
        }
    }
    return v;
}

template <int NT>
__device__ __forceinline__ double fpr_block_max(double v, double* red)
{
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    v = fpr_wave_max(v);
    if constexpr (NT > FPR_WAVE) {
        if ((tid & (FPR_WAVE - 1)) == 0) red[tid / FPR_WAVE] = v;
        __syncthreads();
        if (tid == 0) {
            double s = red[0];
#pragma unroll
            for (int w = 1; w < NT / FPR_WAVE; ++w) s = fmax(s, red[w]);
            v = s;
        }
    }
    return v;
}

// Deterministic sum of n block partials by ONE block of 256 threads: strided accumulation per
// thread, then the fixed block tree.  Result valid in thread 0.
__device__ __forceinline__ double fpr_sum_partials_256(const double* __restrict__ part, int n, double* red)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    return fpr_block_sum<256>(s, red);
}

// end of a V-cycle: sum(res.^2) of the last post-smoothing sweep exactly as k_finish<0> sums it, r_rms (multigrid.jl:252) and the
// loop's exit test (:70) on the device.  One block of 256 threads.
// (WIDE: a block of more than 256 threads -- the first 256 sum as a block of 256 would, every thread takes part in the barrier)
template <bool WIDE = false>
__device__ __forceinline__ void fpr_cycle_finish_body(const FprFinishArgs& a, double* red)
{
    if (a.ctl->stop) return;
    double s;
    if constexpr (WIDE) {
        double v = 0.0;
        if (threadIdx.x < 256)
            for (int i = threadIdx.x; i < a.n; i += 256) v += a.partials[i];
        v = fpr_wave_sum(v);
        if (threadIdx.x < 256 && (threadIdx.x & (FPR_WAVE - 1)) == 0) red[threadIdx.x / FPR_WAVE] = v;
        __syncthreads();
        s = 0.0;
        if (threadIdx.x == 0) {
            s = red[0];
#pragma unroll
            for (int w = 1; w < 256 / FPR_WAVE; ++w) s += red[w];
        }
    } else {
        s = fpr_sum_partials_256(a.partials, a.n, red);
    }
    if (threadIdx.x == 0) {
        a.out[0] = s;
        const double r = sqrt(s / a.npoints);
        FprCycleCtl c = *a.ctl;
        c.rms = r;
        c.ncycles += 1;
        c.coarse_iters = a.st->acc_iters;
        if (r < c.tolf) c.stop = 1;
        *a.ctl = c;
        // the record goes to pinned host memory straight from here (no copy command, no event between two cycles);
        // the host polls `seq`, which is written last
        FprCycleCtl* rec_host = a.rec_host;
        rec_host->stop = c.stop; rec_host->ncycles = c.ncycles; rec_host->coarse_iters = c.coarse_iters;
        rec_host->tolf = c.tolf; rec_host->rms = c.rms; rec_host->frms = c.frms;
        __threadfence_system();
        __hip_atomic_store(&rec_host->seq, c.ncycles, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- twofold-precision sums (Ogita / Rump / Oishi: Sum2, Dot2) ---------------------------------------------------------------
// A sum is carried as (s, e): s the running floating-point sum, e the plain sum of the rounding errors of every addition
// (TwoSum) and product (an explicit fma -- not a contraction: the library is built with -ffp-contract=off) made on the way.
// fl(s + e) is the exact sum rounded once, up to a relative (depth * eps)^2 -- whatever the order of the additions.  cg!'s three
// dot products (krylov.jl:64,69,83) are formed this way here AND in the oracle, so that both round the same near-exact numbers
// and the iteration no longer depends on the summation order (the reference's own order is unspecified: Julia's pairwise
// `sum` on the CPU, CUDA.jl's tree on the GPU).  Only s is on the critical path of a reduction tree; e trails it.
__device__ __forceinline__ void fpr_two_sum(double a, double b, double& s, double& err)
{
    const double t = a + b;
    const double bb = t - a;
    err = (a - (t - bb)) + (b - bb);
    s = t;
}
__device__ __forceinline__ void fpr_s2_add(double& s, double& e, double v)           // (s, e) += v
{
    double t, d;
    fpr_two_sum(s, v, t, d);
    s = t;
    e += d;
}
__device__ __forceinline__ void fpr_s2_merge(double& s, double& e, double s2, double e2)   // (s, e) += (s2, e2)
{
    double t, d;
    fpr_two_sum(s, s2, t, d);
    s = t;
    e = (e + e2) + d;
}
__device__ __forceinline__ void fpr_s2_add_prod(double& s, double& e, double a, double b)   // (s, e) += a * b, exactly
{
    const double p = a * b;
    const double pe = __builtin_fma(a, b, -p);
    double t, d;
    fpr_two_sum(s, p, t, d);
    s = t;
    e += d + pe;
}
// over the 64 lanes of a wave (DPP row shifts as fpr_wave_sum_all); the same (s, e) in every lane
__device__ __forceinline__ void fpr_wave_sum_all_s2(double& s, double& e)
{
    fpr_s2_merge(s, e, fpr_dpp<0x111>(s), fpr_dpp<0x111>(e));
    fpr_s2_merge(s, e, fpr_dpp<0x112>(s), fpr_dpp<0x112>(e));
    fpr_s2_merge(s, e, fpr_dpp<0x114>(s), fpr_dpp<0x114>(e));
    fpr_s2_merge(s, e, fpr_dpp<0x118>(s), fpr_dpp<0x118>(e));
    auto lane = [](double v, int l) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
    };
    double rs = lane(s, 15), re = lane(e, 15);
    fpr_s2_merge(rs, re, lane(s, 31), lane(e, 31));
    fpr_s2_merge(rs, re, lane(s, 47), lane(e, 47));
    fpr_s2_merge(rs, re, lane(s, 63), lane(e, 63));
    s = rs;
    e = re;
}
// Block-wide, fixed order; result valid in thread 0.  `red` = __shared__ double[32].
template <int NT>
__device__ __forceinline__ void fpr_block_sum_s2(double& s, double& e, double* red)
{
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    fpr_wave_sum_all_s2(s, e);
    if constexpr (NT > FPR_WAVE) {
        if ((tid & (FPR_WAVE - 1)) == 0) { red[tid / FPR_WAVE] = s; red[16 + tid / FPR_WAVE] = e; }
        __syncthreads();
        if (tid == 0) {
            double ts = red[0], te = red[16];
#pragma unroll
            for (int w = 1; w < NT / FPR_WAVE; ++w) fpr_s2_merge(ts, te, red[w], red[16 + w]);
            s = ts;
            e = te;
        }
    }
}
// n block partials stored as pairs (part[2 i] = s, part[2 i + 1] = e) summed by ONE block of 256 threads; valid in thread 0
__device__ __forceinline__ void fpr_sum_partials_256_s2(const double* __restrict__ part, int n, double* red, double& s, double& e)
{
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    double ts = 0.0, te = 0.0;
    for (int i = tid; i < n; i += 256) fpr_s2_merge(ts, te, part[2 * i], part[2 * i + 1]);
    fpr_block_sum_s2<256>(ts, te, red);
    s = ts;
    e = te;
}

#endif  // __HIPCC__

// cross-TU internal entry points (implemented in reduce.hip)
int fprx_sumsq_scaled_dev(fpr_ctx* ctx, const double* x, size_t n, double scale, double* out_dev, int stream_sel);
int fprx_dot_dev(fpr_ctx* ctx, const double* x, const double* y, size_t n, double* out_dev);
int fprx_dot2_dev(fpr_ctx* ctx, const double* x, const double* y, size_t n, double* out_dev);   // twofold-precision dot (Dot2), rounded once
// finish a two-stage reduction: out_dev[0] (= or +=) sum(partials[0..nparts))
int fprx_finish_sum(fpr_ctx* ctx, const double* partials, int nparts, double* out_dev, bool accumulate, int stream_sel);
int fprx_cycle_init(fpr_ctx* ctx, const double* f, size_t n, double tol);
int fprx_cycle_reset(fpr_ctx* ctx, double tol);
int fprx_cycle_init_from(fpr_ctx* ctx, const double* partials, int nparts, size_t n, double tol);
int fprx_cycle_finish(fpr_ctx* ctx, const double* partials, int nparts, double* sumsq_out_dev, double npoints, int slot);
int fprx_cycle_wait(fpr_ctx* ctx, int slot, int seq, FprCycleCtl* out);
// the same finish, handed to the next pass below the finest level (vcycle_level consumes ctx->fin; whatever launches first there without
// being able to carry it calls fprx_cycle_finish_flush, which launches k_cycle_finish after all)
int fprx_cycle_finish_defer(fpr_ctx* ctx, const double* partials, int nparts, double* sumsq_out_dev, double npoints, int slot);
int fprx_cycle_finish_flush(fpr_ctx* ctx);
int fprx_finish_sum2_plus(fpr_ctx* ctx, const double* p0, const double* p1, int nparts, const double* add2_dev, double* out2_dev,
                          int stream_sel);   // out[b] = sum(list b) + add[b]
void fprx_ns_worker_free(fpr_ctx* ctx);   // navier2d.hip
// fpr_halo_exchange3d_comm with the x-planes travelling from / into buffers of the caller (no pack / unpack kernels for them)
int fprx_halo_exchange3d_comm_x(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask, const double* const xsend[2],
                                double* const xrecv[2]);
int fprx_finish_sum2(fpr_ctx* ctx, const double* p0, const double* p1, int nparts, double* out2_dev, bool accumulate,
                     int stream_sel);
int fprx_finish_sum3(fpr_ctx* ctx, const double* p0, const double* p1, const double* p2, int nparts, double* out3_dev, int stream_sel,
                     const double* add3_dev = nullptr);   // add3_dev: three device doubles added to the three sums
// one group of sends / receives of whole z-planes between the caller's buffers and the z-neighbours (side 0 = low face), comm stream
int fprx_exchange_zplanes(fpr_ctx* ctx, const double* const zsend[2], double* const zrecv[2]);
