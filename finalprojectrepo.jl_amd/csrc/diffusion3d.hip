// diffusion3d.hip -- Part 1 entry points of libfpr_hip.so: fused 7-point pseudo-transient update,
// the split compute_flux!/compute_dHdtau!/update_H! form, Gaussian initial condition.
// Reference: scripts-part1/part1_kernel_programming.jl, part1_array_programming.jl, part1_utils.jl.
#include "diffusion3d_kernels.hpp"
#include "fpr_internal.hpp"
#include <cstring>

// defined in diffusion3d_launch.hpp (shared with tools/diffusion_tune.hip)
#include "diffusion3d_launch.hpp"
// two pseudo-iterations per pass (temporal blocking): main kernel and the kernel for boxes that are narrow in x
#include "diffusion3d_fused2.hpp"
#include "diffusion3d_fused3.hpp"
#include "diffusion3d_slab2.hpp"
// the shell next to an x-neighbour in compact strips
#include "diffusion3d_xstrip.hpp"

static int diff3_run(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2, double* dHdtau, int nx, int ny,
                     int nz, double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx, double D_dy,
                     double D_dz, const int* lo, const int* hi, bool norm, double scale, double* sumsq_dev,
                     bool accumulate, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, Ht && Htau && Htau2 && dHdtau, "null field pointer");
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3 && nz >= 3, "grid must be at least 3^3");
    FPR_REQUIRE(ctx, Htau != Htau2, "Htau and Htau2 must be distinct buffers (Jacobi ping-pong)");
    FPR_REQUIRE(ctx, stream_sel >= 0 && stream_sel <= 2, "stream_sel");
    Diff3Args a;
    a.Ht = Ht; a.Htau = Htau; a.Htau2 = Htau2; a.dHdtau = dHdtau;
    a.nx = nx; a.ny = ny; a.nz = nz;
    const int n[3] = {nx, ny, nz};
    for (int d = 0; d < 3; ++d) {
        a.lo[d] = lo ? (lo[d] < 1 ? 1 : lo[d]) : 1;
        a.hi[d] = hi ? (hi[d] > n[d] - 1 ? n[d] - 1 : hi[d]) : n[d] - 1;
    }
    a.dtau = dtau; a._dt = _dt; a._dx = _dx; a._dy = _dy; a._dz = _dz;
    a.D_dx = D_dx; a.D_dy = D_dy; a.D_dz = D_dz;
    a.scale = scale;
    // option fp_contract = 1 (opt-in, default 0): the contracted form of the point update for launches over the WHOLE interior on the
    // compute stream (single-rank runs); boxes of a decomposed run stay exact, so that shell and core agree bit for bit
    a.fma = (!lo && !hi && stream_sel == 0 && fpr_opt(ctx, "fp_contract", 0) != 0) ? 1 : 0;
    a.partials = stream_sel == 1 ? ctx->partials2 : ctx->partials;
    const bool empty = a.lo[0] >= a.hi[0] || a.lo[1] >= a.hi[1] || a.lo[2] >= a.hi[2];
    int nparts = 0;
    if (!empty) {
        const Diff3Tuning t;      // the library runs the default form only (the other tilings live in tools/diffusion_tune.hip, -DFPR_TUNE)
        const bool timed = fpr_ktimer_begin(ctx, FPR_KT_DIFF3_STEP, ctx->stream[stream_sel]);
        // a box up to 8 cells wide in x (the slab next to an x-neighbour) goes to the kernel whose lanes run along y;
        // needs 8-byte loads through buffer descriptors only, so any size / alignment qualifies
        const bool narrow = (a.hi[0] - a.lo[0]) <= 8 &&
                            (long)nx * ny * 8 * 12 < (1L << 31) && ny >= 3;
        hipError_t e = narrow ? diff3_launch_slab1(a, norm, ctx->stream[stream_sel], FPR_MAX_PARTIALS, &nparts)
                              : diff3_launch(a, norm, t, ctx->stream[stream_sel], FPR_MAX_PARTIALS, &nparts);
        fpr_ktimer_end(ctx, timed, ctx->stream[stream_sel]);
        if (e != hipSuccess) return fpr_fail(ctx, FPR_ERR_HIP, "diffusion3d launch: %s", hipGetErrorString(e));
    }
    if (norm) {
        if (empty) {
            if (!accumulate) FPR_HIP(ctx, hipMemsetAsync(sumsq_dev, 0, sizeof(double), ctx->stream[stream_sel]));
            return FPR_OK;
        }
        return fprx_finish_sum(ctx, a.partials, nparts, sumsq_dev, accumulate, stream_sel);
    }
    return FPR_OK;
}

extern "C" int fpr_diffusion3d_step(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2, double* dHdtau,
                                    int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz,
                                    double D_dx, double D_dy, double D_dz)
{
    return diff3_run(ctx, Ht, Htau, Htau2, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, nullptr,
                     nullptr, false, 0.0, nullptr, false, 0);
}

extern "C" int fpr_diffusion3d_step_norm(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2,
                                         double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx,
                                         double _dy, double _dz, double D_dx, double D_dy, double D_dz, double scale,
                                         double* sumsq_dev)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, sumsq_dev, "sumsq_dev is null");
    return diff3_run(ctx, Ht, Htau, Htau2, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, nullptr,
                     nullptr, true, scale, sumsq_dev, false, 0);
}

extern "C" int fpr_diffusion3d_step_norm_host(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2,
                                              double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx,
                                              double _dy, double _dz, double D_dx, double D_dy, double D_dz, double scale,
                                              double* sumsq_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, sumsq_host, "sumsq_host is null");
    double* pinned = ctx->host_scalars + 8;  // device-visible pinned host memory: the finish kernel writes it directly
    int rc = diff3_run(ctx, Ht, Htau, Htau2, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, nullptr,
                       nullptr, true, scale, pinned, false, 0);
    if (rc) return rc;
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
    *sumsq_host = *(volatile double*)pinned;
    return FPR_OK;
}

// ------------------------------------------------------------------------------------------------
// Two pseudo-iterations in one pass: step(Ht, A -> [B]); step(Ht, [B] -> C) with B never materialised.
// ------------------------------------------------------------------------------------------------
static bool diff3_fuse2_ok(fpr_ctx* ctx, const double* Ht, const double* A, const double* B, const double* C,
                           const double* dH, int nx, int ny, int nz)
{
    return fpr_opt(ctx, "diff3_fuse2", 1) != 0 && diff3_can_fuse2(Ht, A, B, C, dH, nx, ny, nz) &&
           (long)nx * ny * 8 * 12 < (1L << 31);
}

// sumsq2_dev: two doubles (first, second iteration); nullptr = no norm
static int diff3_run2(fpr_ctx* ctx, const double* Ht, const double* A, const double* B, double* C, double* dH, int nx,
                      int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx, double D_dy,
                      double D_dz, const int* lo, const int* hi, double scale, double* sumsq2_dev, bool accumulate,
                      int stream_sel, int zlo2 = 0, int zhi2 = 0, const int* skip = nullptr, int* nparts_only = nullptr,
                      int reserve_cus = 0, double* partial_base = nullptr, int partial_cap = 0)
{   // partial_base / partial_cap: two lists of partial_cap doubles for the partials instead of the stream's scratch
    // nparts_only: the launch reduces both norms to per-workgroup partials (partials1 / partials2 of the stream's scratch) and
    // leaves the finishing to the caller: *nparts_only = their number
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, Ht && A && B && C, "null field pointer");   // dH may be null: residual not stored
    FPR_REQUIRE(ctx, A != C && B != C && A != B, "Htau, Hmid and Hout must be three distinct buffers");
    FPR_REQUIRE(ctx, stream_sel >= 0 && stream_sel <= 2, "stream_sel");
    FPR_REQUIRE(ctx, diff3_fuse2_ok(ctx, Ht, A, B, C, dH, nx, ny, nz), "problem not supported by the fused two-step kernel");
    Diff3Args2 a;
    a.skip = skip;
    a.lane_off = 1;
    a.Ht = Ht; a.A = A; a.B = B; a.C = C; a.dH = dH;
    a.nx = nx; a.ny = ny; a.nz = nz;
    const int n[3] = {nx, ny, nz};
    for (int d = 0; d < 3; ++d) {
        a.lo[d] = lo ? (lo[d] < 1 ? 1 : lo[d]) : 1;
        a.hi[d] = hi ? (hi[d] > n[d] - 1 ? n[d] - 1 : hi[d]) : n[d] - 1;
    }
    a.dtau = dtau; a._dt = _dt; a._dx = _dx; a._dy = _dy; a._dz = _dz;
    a.D_dx = D_dx; a.D_dy = D_dy; a.D_dz = D_dz;
    a.scale = scale;
    a.fma = (!lo && !hi && stream_sel == 0 && reserve_cus == 0 && zhi2 <= zlo2 && fpr_opt(ctx, "fp_contract", 0) != 0) ? 1 : 0;   // as diff3_run
    double* base = stream_sel == 1 ? ctx->partials2 : ctx->partials;
    const bool is_core = reserve_cus > 0 || stream_sel == 2;   // fpr_diffusion3d_step2_core (kernel timer kind)
    // a launch on the core stream of a split device (fpr_reserve_comm_cus) has that many units less, whatever the caller says
    if (stream_sel == 2 && ctx->comm_cus > reserve_cus) reserve_cus = ctx->comm_cus;
    const int pcap = partial_base ? partial_cap : FPR_MAX_PARTIALS / 2;
    a.partials1 = partial_base ? partial_base : base;
    a.partials2 = a.partials1 + pcap;
    const bool norm = sumsq2_dev != nullptr || nparts_only != nullptr;
    if (zlo2 < 1) zlo2 = 1;
    if (zhi2 > nz - 1) zhi2 = nz - 1;
    if (a.lo[2] >= a.hi[2] && zhi2 > zlo2) {   // empty first z-range: the second one takes its place
        a.lo[2] = zlo2; a.hi[2] = zhi2;
        zlo2 = zhi2 = 0;
    }
    const bool empty = a.lo[0] >= a.hi[0] || a.lo[1] >= a.hi[1] || a.lo[2] >= a.hi[2];
    int nparts = 0;
    if (!empty) {
        if (ctx->ncu <= 0) {
            int v = 0;
            ctx->ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ? v : 256;
        }
        // boxes up to 12 cells wide in x (the slab next to an x-neighbour of a decomposed run) go to the kernel whose
        // lanes run along y; everything else to the wave-tile kernel.  A second z-range rides along in either case.
        const bool narrow = (a.hi[0] - a.lo[0]) <= 12;
        const bool timed = fpr_ktimer_begin(ctx, is_core ? FPR_KT_DIFF3_CORE : FPR_KT_DIFF3_STEP2, ctx->stream[stream_sel]);
        hipError_t e;
        if (narrow) {
            e = diff3_launch_slab2(a, norm, ctx->stream[stream_sel], pcap, &nparts);
            if (e == hipSuccess && zhi2 > zlo2) {
                Diff3Args2 b = a;
                b.lo[2] = zlo2; b.hi[2] = zhi2;
                b.partials1 = a.partials1 + nparts; b.partials2 = a.partials2 + nparts;
                int np2 = 0;
                e = diff3_launch_slab2(b, norm, ctx->stream[stream_sel], pcap - nparts, &np2);
                nparts += np2;
            }
        } else {
            if (reserve_cus > 0 && fpr_opt(ctx, "diff3_bal_g", 0) > 0) reserve_cus = -(int)fpr_opt(ctx, "diff3_bal_g", 0);   // tests
            long bal_info = 0;
            // a launch on the comm stream of a split device has that stream's compute units only: chunk it for them
            const int ncu_plan = (stream_sel == 1 && ctx->comm_cus > 0) ? ctx->comm_cus : ctx->ncu;
            // the reserved form of a core launch takes tickets: one workgroup per device slot, the ones the split leaves no slot for find no work
            // (a launch leaves its counters zeroed; launches of ONE stream follow each other, and every stream selector has a block of
            // its own, so a ticketed launch on the compute stream cannot meet a pending pair's core launch in the same counters)
            const bool tickets = reserve_cus != 0 && stream_sel != 1;
            if (tickets && !ctx->tickets) {
                FPR_HIP(ctx, hipMalloc(&ctx->tickets, 3 * 16 * sizeof(int)));
                FPR_HIP(ctx, hipMemset(ctx->tickets, 0, 3 * 16 * sizeof(int)));   // once, complete before any stream goes on
            }
            const unsigned* rmap = (tickets && ctx->core_unmasked && stream_sel == 2) ? ctx->reserved_map : nullptr;
            if (rmap && fpr_opt(ctx, "diff3_reserved_test", 0)) {
                // tests: a WRONG map (1: every unit marked as a comm unit, 2: every other key) -- the claim protocol must still serve every
                // unit of work (workgroups on 'comm units' take work once every workgroup has started), only later
                FPR_HIP(ctx, hipMemsetAsync(ctx->reserved_map + 64, fpr_opt(ctx, "diff3_reserved_test", 0) == 1 ? 0xFF : 0xAA, 64 * sizeof(unsigned),
                                            ctx->stream[stream_sel]));
                rmap = ctx->reserved_map + 64;
            }
            // (option diff3_zc2: planes per z-chunk, a test hook -- small test grids otherwise run as one chunk)
            e = diff3_launch2(a, norm, (int)fpr_opt(ctx, "diff3_zc2", 0), 0, ctx->stream[stream_sel], pcap, &nparts, 0, ncu_plan, zlo2, zhi2,
                              reserve_cus, &bal_info,
                              tickets ? ctx->tickets + 16 * stream_sel : nullptr, rmap);
            if (reserve_cus != 0) ctx->options["diff3_last_bal"] = bal_info;   // diagnostic (fpr_get_option): which form ran
        }
        fpr_ktimer_end(ctx, timed, ctx->stream[stream_sel]);
        if (e != hipSuccess) return fpr_fail(ctx, FPR_ERR_HIP, "fused diffusion3d launch: %s", hipGetErrorString(e));
    }
    if (nparts_only) {
        *nparts_only = nparts;
        return FPR_OK;
    }
    if (norm) {
        if (empty) {
            if (!accumulate) FPR_HIP(ctx, hipMemsetAsync(sumsq2_dev, 0, 2 * sizeof(double), ctx->stream[stream_sel]));
            return FPR_OK;
        }
        return fprx_finish_sum2(ctx, a.partials1, a.partials2, nparts, sumsq2_dev, accumulate, stream_sel);
    }
    return FPR_OK;
}

extern "C" int fpr_diffusion3d_can_step2(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid,
                                         const double* Hout, const double* dHdtau, int nx, int ny, int nz)
{
    if (!ctx) return FPR_ERR_INVALID;
    return (Ht && Htau && Hmid && Hout && Htau != Hout && Hmid != Hout && Htau != Hmid &&   // dHdtau may be NULL
            diff3_fuse2_ok(ctx, Ht, Htau, Hmid, Hout, dHdtau, nx, ny, nz))
               ? 1
               : 0;
}

extern "C" int fpr_diffusion3d_step2(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid, double* Hout,
                                     double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy,
                                     double _dz, double D_dx, double D_dy, double D_dz, double scale, double* sumsq2_dev)
{
    return diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, nullptr,
                      nullptr, scale, sumsq2_dev, false, 0);
}

extern "C" int fpr_diffusion3d_step2_box(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid,
                                         double* Hout, double* dHdtau, int nx, int ny, int nz, double dtau, double _dt,
                                         double _dx, double _dy, double _dz, double D_dx, double D_dy, double D_dz,
                                         const int lo[3], const int hi[3], double scale, double* sumsq2_dev, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, lo && hi, "null box");
    return diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi,
                      scale, sumsq2_dev, true, stream_sel);
}

// ------------------------------------------------------------------------------------------------
// Three pseudo-iterations in one pass (diffusion3d_fused3.hpp): step(X -> [Y']); step([Y'] -> [X']); step([X'] -> Y), the two fields
// in between never materialised.  X and Y are the reference's two ping-pong buffers in either order: an odd depth needs no third buffer.
// ------------------------------------------------------------------------------------------------
static bool diff3_fuse3_ok(fpr_ctx* ctx, const double* Ht, const double* X, const double* Y, const double* dH, int nx, int ny, int nz)
{
    return fpr_opt(ctx, "diff3_fuse3", 1) != 0 && diff3_can_fuse3(Ht, X, Y, Y, dH, nx, ny, nz);
}

// sumsq3_dev: three doubles (first, second, third iteration); nullptr = no norm.  nparts_only: leave the three lists of per-workgroup
// partials (thirds of the stream's scratch) to the caller.  Whole interior, compute stream.
// lo / hi: output box (nullptr = the whole interior); stream_sel 2 = the core launch of a triple between ranks (fpr_diffusion3d_step3_halo):
// planned for the compute units the split leaves to the core stream, partials in partial_base (three lists of partial_cap doubles).
static int diff3_run3(fpr_ctx* ctx, const double* Ht, const double* X, double* Y, double* dH, int nx, int ny, int nz, double dtau,
                      double _dt, double _dx, double _dy, double _dz, double D_dx, double D_dy, double D_dz, double scale,
                      double* sumsq3_dev, const int* skip = nullptr, int* nparts_only = nullptr, const int* lo = nullptr, const int* hi = nullptr,
                      int stream_sel = 0, double* partial_base = nullptr, int partial_cap = 0)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, Ht && X && Y, "null field pointer");   // dH may be null: residual not stored
    FPR_REQUIRE(ctx, X != Y && Ht != Y && dH != Y && dH != X, "Htau, Hout and dHdtau must be distinct buffers");
    FPR_REQUIRE(ctx, diff3_fuse3_ok(ctx, Ht, X, Y, dH, nx, ny, nz), "problem not supported by the fused three-step kernel");
    Diff3Args3 a;
    memset(&a, 0, sizeof a);
    a.skip = skip;
    a.lane_off = 1;
    a.Ht = Ht; a.X = X; a.Bnd = Y; a.Y = Y; a.dH = dH;
    a.nx = nx; a.ny = ny; a.nz = nz;
    const int n[3] = {nx, ny, nz};
    for (int d = 0; d < 3; ++d) {
        a.lo[d] = lo ? (lo[d] < 1 ? 1 : lo[d]) : 1;
        a.hi[d] = hi ? (hi[d] > n[d] - 1 ? n[d] - 1 : hi[d]) : n[d] - 1;
    }
    a.dtau = dtau; a._dt = _dt; a._dx = _dx; a._dy = _dy; a._dz = _dz;
    a.D_dx = D_dx; a.D_dy = D_dy; a.D_dz = D_dz;
    a.scale = scale;
    FPR_REQUIRE(ctx, stream_sel == 0 || stream_sel == 2, "stream_sel");
    const int pcap = partial_base ? partial_cap : FPR_MAX_PARTIALS / 3;
    a.partials1 = partial_base ? partial_base : ctx->partials; a.partials2 = a.partials1 + pcap; a.partials3 = a.partials2 + pcap;
    const bool norm = sumsq3_dev != nullptr || nparts_only != nullptr;
    if (ctx->ncu <= 0) {
        int v = 0;
        ctx->ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ? v : 256;
    }
    // a launch on the core stream of a split device has the units the comm stream does not own: one whole unit of work per such unit,
    // the left-over units in thin slices
    const int ncu_plan = (stream_sel == 2 && ctx->comm_cus > 0 && !ctx->core_unmasked) ? ctx->ncu - ctx->comm_cus : ctx->ncu;
    int nparts = 0;
    const bool timed = fpr_ktimer_begin(ctx, FPR_KT_DIFF3_STEP3, ctx->stream[stream_sel]);
    const hipError_t e = diff3_launch3(a, norm, (int)fpr_opt(ctx, "diff3_zc2", 0), 0, ctx->stream[stream_sel], pcap, &nparts, ncu_plan);
    fpr_ktimer_end(ctx, timed, ctx->stream[stream_sel]);
    if (e != hipSuccess) return fpr_fail(ctx, FPR_ERR_HIP, "fused three-step diffusion3d launch: %s", hipGetErrorString(e));
    if (nparts_only) {
        *nparts_only = nparts;
        return FPR_OK;
    }
    if (norm) return fprx_finish_sum3(ctx, a.partials1, a.partials2, a.partials3, nparts, sumsq3_dev, stream_sel);
    return FPR_OK;
}

extern "C" int fpr_diffusion3d_can_step3(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hout, const double* dHdtau,
                                         int nx, int ny, int nz)
{
    if (!ctx) return FPR_ERR_INVALID;
    return (Ht && Htau && Hout && Htau != Hout && Ht != Hout && dHdtau != Hout && dHdtau != Htau &&   // dHdtau may be NULL
            diff3_fuse3_ok(ctx, Ht, Htau, Hout, dHdtau, nx, ny, nz))
               ? 1
               : 0;
}

extern "C" int fpr_diffusion3d_step3(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Hout, double* dHdtau, int nx, int ny,
                                     int nz, double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx, double D_dy,
                                     double D_dz, double scale, double* sumsq3_dev)
{
    return diff3_run3(ctx, Ht, Htau, Hout, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, sumsq3_dev);
}

// ---- the solver's loop with its exit test on the device (pairs enqueued ahead of the host) ---------------------------
// part1_kernel_programming.jl:179-192 is "iterate, read the norm on the host, compare, iterate".  At 128^3 a fused pair of
// iterations takes ~15 us on the device and the host round trip behind it ~40 us.  As for MGsolve: the norms of
// a pair stay on the device, k_diff3_check evaluates `err > tol` (:179) for the first and the second iteration of the pair and
// raises a stop flag that every later pair (k_diff3_march2's `skip`) honours; the record of the pair goes to pinned host
// memory from the kernel, sequence number last.  The FprCycleCtl fields are reused: ncycles = iterations executed,
// coarse_iters = 1 if the loop ended on the FIRST iteration of the pair, tolf = tol, rms = err, frms = sqrt(N).
__global__ void k_diff3_ctl_init(FprCycleCtl* ctl, double tol, double sqrtN)
{
    ctl->stop = 0; ctl->ncycles = 0; ctl->coarse_iters = 0; ctl->seq = 0;
    ctl->tolf = tol; ctl->rms = 0.0; ctl->frms = sqrtN;
}

// finishes the norms of a fused launch (two or three iterations) exactly as k_finish2 / k_finish3 do (same order of summation) and takes the
// loop's decision: the first iteration j whose norm is wanted and ends the loop stops it -- coarse_iters = j if later iterations of the launch
// exist (the host then redoes j iterations from the launch's intact input), 0 if it is the launch's last
__global__ __launch_bounds__(256) void k_diff3_check(FprCycleCtl* ctl, const double* __restrict__ p1, const double* __restrict__ p2,
                                                      const double* __restrict__ p3, int depth, int nparts, int n1, int n2, int n3,
                                                      int it_base, int seq, FprCycleCtl* rec_host)
{
    __shared__ double red[16];
    if (ctl->stop) return;
    double sj[3];
    sj[0] = fpr_sum_partials_256(p1, nparts, red);
    __syncthreads();
    sj[1] = fpr_sum_partials_256(p2, nparts, red);
    if (depth > 2) {
        __syncthreads();
        sj[2] = fpr_sum_partials_256(p3, nparts, red);
    }
    if (threadIdx.x != 0) return;
    FprCycleCtl c = *ctl;
    c.coarse_iters = 0;
    c.ncycles = it_base + depth;
    const int want[3] = {n1, n2, n3};
    for (int j = 0; j < depth; ++j) {
        if (!want[j]) continue;
        const double e = sqrt(sj[j]) / c.frms;   // :191 after iteration j + 1 of the launch
        c.rms = e;
        if (!(e > c.tolf)) {
            c.stop = 1;
            c.coarse_iters = (j + 1 < depth) ? j + 1 : 0;
            c.ncycles = it_base + j + 1;
            break;
        }
    }
    *ctl = c;
    rec_host->stop = c.stop; rec_host->ncycles = c.ncycles; rec_host->coarse_iters = c.coarse_iters;
    rec_host->tolf = c.tolf; rec_host->rms = c.rms; rec_host->frms = c.frms;
    __threadfence_system();
    __hip_atomic_store(&rec_host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int fpr_diffusion3d_solve(fpr_ctx* ctx, double* Ht, double* Htau, double* Htau2, double* Htau3, double* dHdtau,
                                     int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx,
                                     double D_dy, double D_dz, double dt, double total_N, int nt, double tol, long iter_max,
                                     long fixed_iters, int check_every, long* iters_host, double* err_host, int* swapped_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, Ht && Htau && Htau2 && dHdtau, "null field pointer");
    FPR_REQUIRE(ctx, nt >= 0 && check_every >= 1, "nt >= 0 and check_every >= 1");
    const size_t N = (size_t)nx * ny * nz;
    const double sqrtN = sqrt(total_N);
    volatile double* pinned = ctx->host_scalars + 8;   // [0], [1]: sums of the first / second iteration of a launch

    // The reference's two work buffers keep their own boundary values (the kernel writes interior cells only), so
    // the field alternates between an "even" buffer (Htau's boundary) and an "odd" one (Htau2's).  With a third
    // buffer E1 that carries Htau's boundary two iterations run as one fused launch even -> even, reading only the
    // boundary of Htau2 (diffusion3d_fused2.hpp); single steps handle odd states and odd iteration counts.
    // Htau3: the caller's third work buffer (the library allocates no field-sized memory).  NULL = one iteration per
    // launch.  Given but unusable (size / alignment outside the fused kernel's range): an error, not a silent detour.
    double* E1 = nullptr;
    bool fuse = false;
    if (Htau3 && fpr_opt(ctx, "diff3_fuse2", 1) != 0) {
        FPR_REQUIRE(ctx, Htau3 != Ht && Htau3 != Htau && Htau3 != Htau2 && Htau3 != dHdtau, "Htau3 must be a buffer of its own");
        if (!diff3_fuse2_ok(ctx, Ht, Htau, Htau2, Htau3, dHdtau, nx, ny, nz))
            return fpr_fail(ctx, FPR_ERR_INVALID, "Htau3 given, but %dx%dx%d / these pointers are outside the fused two-iteration "
                            "kernel's range (nx even >= 128, ny >= 16, 16-byte aligned arrays): pass NULL (see "
                            "fpr_diffusion3d_can_step2)", nx, ny, nz);
        E1 = Htau3;
        fuse = true;
        int rc = fpr_copy(ctx, E1, Htau, N);   // boundary of the even buffers (interior is overwritten)
        if (rc) return rc;
    }
    // Three iterations per launch (k_diff3_march3) need no third buffer: the field goes from the buffer it is in to the reference's other
    // one -- Htau <-> Htau2 as in the reference, three iterations at a time; taken whenever three iterations are left, whatever the parity
    const bool fuse3 = diff3_fuse3_ok(ctx, Ht, Htau, Htau2, dHdtau, nx, ny, nz) && (!E1 || diff3_fuse3_ok(ctx, Ht, E1, Htau2, dHdtau, nx, ny, nz));
    double* cur = Htau;   // current field
    int parity = 0;       // 0: cur is an even buffer (Htau or E1), 1: cur == Htau2
    long swaps = 0;
    // The loop needs the residual's NORM every iteration, the residual ARRAY only when it returns (":191" reads it, nothing
    // else does): fused pairs therefore run without storing dHdtau (24 instead of 32 bytes per cell and launch), and
    // if the call ends on such a pair the pair is replayed once with the store -- from its still intact input, before
    // the commit of Ht -- so that dHdtau holds what the reference's residual_H holds.  Option diff3_lazy_residual = 0:
    // every launch stores it.
    // an error return in the middle of a run of pairs enqueued ahead of the host: drain them before the caller gets control
    // back (they would otherwise still write the fields after the call has returned); outputs are undefined after an error
    struct Drain {
        fpr_ctx* c;
        bool ok;
        ~Drain() { if (!ok) hipStreamSynchronize(c->stream[0]); }
    } drain{ctx, false};
    const bool lazy_res = fpr_opt(ctx, "diff3_lazy_residual", 1) != 0;
    int ahead = (int)fpr_opt(ctx, "diff3_ahead", 2);   // fused pairs enqueued ahead of the host's view of the norm (0: wait for every norm)
    if (ahead > FPR_CYC_SLOTS - 2) ahead = FPR_CYC_SLOTS - 2;
    const double* stale_in = nullptr;   // input of the last fused launch if dHdtau has not been written since
    double* stale_out = nullptr;
    int stale_depth = 2;
    // j single iterations from `in` (1 <= j <= 2): what the reference has done when it stops inside a fused launch
    auto redo_singles = [&](const double* in, int par, int j, double** cur_out) -> int {
        const double* src = in;
        double* out = nullptr;
        for (int q = 0; q < j; ++q) {
            out = par ? Htau : Htau2;
            if (int rc = diff3_run(ctx, Ht, src, out, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, nullptr, nullptr, false, 0.0,
                                   nullptr, false, 0))
                return rc;
            src = out; par ^= 1;
        }
        *cur_out = out;
        return FPR_OK;
    };
    for (int t = 0; t < nt; ++t) {
        long it = 0;
        double err = 2 * tol;  // :178
        auto more = [&]() { return fixed_iters > 0 ? it < fixed_iters : (err > tol && it < iter_max); };  // :179
        auto want_norm = [&](long j) { return fixed_iters > 0 ? (j == fixed_iters) : (j % check_every == 0); };
        while (more()) {
            const long left = fixed_iters > 0 ? fixed_iters - it : iter_max - it;
            if (fuse3 && left >= 3 && fixed_iters <= 0 && ahead > 0) {
                // ---- a run of fused TRIPLES enqueued `ahead` deep before the host looks at the norm of the oldest one ----
                struct Trip { const double* in; double* out; long it_base; int slot, seq, par; bool rec; } ring[FPR_CYC_SLOTS];
                hipStream_t s = ctx->stream[0];
                k_diff3_ctl_init<<<1, 1, 0, s>>>(ctx->cyc, tol, sqrtN);
                FPR_CHECK_LAUNCH(ctx);
                long it_enq = it;
                double* cur_enq = cur;
                int par_enq = parity;
                int enq = 0, seen = 0, nrec = 0;
                const int pcap3 = FPR_MAX_PARTIALS / 3;
                while (true) {
                    while (enq - seen < 1 + ahead && iter_max - it_enq >= 3) {
                        double* out = par_enq ? Htau : Htau2;
                        const bool n1 = want_norm(it_enq + 1), n2 = want_norm(it_enq + 2), n3 = want_norm(it_enq + 3);
                        int nparts = 0;
                        int rc = diff3_run3(ctx, Ht, cur_enq, out, lazy_res ? nullptr : dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy,
                                            D_dz, dt, nullptr, &ctx->cyc->stop, (n1 || n2 || n3) ? &nparts : nullptr);
                        if (rc) return rc;
                        Trip& p = ring[enq % FPR_CYC_SLOTS];
                        p.in = cur_enq; p.out = out; p.it_base = it_enq; p.par = par_enq; p.rec = n1 || n2 || n3; p.slot = 0; p.seq = 0;
                        if (p.rec) {
                            p.slot = nrec % FPR_CYC_SLOTS;
                            p.seq = ++nrec;
                            __atomic_store_n(&ctx->cyc_h[p.slot].seq, 0, __ATOMIC_RELEASE);
                            k_diff3_check<<<1, 256, 0, s>>>(ctx->cyc, ctx->partials, ctx->partials + pcap3, ctx->partials + 2 * pcap3, 3, nparts,
                                                            n1 ? 1 : 0, n2 ? 1 : 0, n3 ? 1 : 0, (int)it_enq, p.seq, &ctx->cyc_h[p.slot]);
                            FPR_CHECK_LAUNCH(ctx);
                        }
                        cur_enq = out; par_enq ^= 1; it_enq += 3; ++enq;
                    }
                    if (seen == enq) break;   // iter_max is less than three iterations away: the paths below finish
                    const Trip& p = ring[seen % FPR_CYC_SLOTS];
                    ++seen;
                    bool stop_now = false;
                    if (p.rec) {
                        FprCycleCtl rec;
                        if (int rc = fprx_cycle_wait(ctx, p.slot, p.seq, &rec)) return rc;
                        err = rec.rms;
                        if (rec.stop && rec.coarse_iters) {
                            // the reference stops after iteration j < 3 of this launch: redo those j iterations one by one from the launch's
                            // input (intact: everything enqueued behind it has returned at once)
                            const int j = (int)rec.coarse_iters;
                            if (int rc = redo_singles(p.in, p.par, j, &cur)) return rc;
                            stale_in = nullptr;
                            parity = p.par ^ (j & 1); swaps += j; it = p.it_base + j;
                            break;
                        }
                        stop_now = rec.stop != 0;
                    }
                    stale_in = lazy_res ? p.in : nullptr;
                    stale_out = p.out; stale_depth = 3;
                    cur = p.out; parity = p.par ^ 1; swaps += 3; it = p.it_base + 3;
                    if (stop_now) break;
                }
                if (seen > 0) continue;
            }
            if (fuse3 && left >= 3) {
                // one triple, the host waits for its norms (fixed iteration counts, diff3_ahead = 0)
                double* out = parity ? Htau : Htau2;
                const bool n1 = want_norm(it + 1), n2 = want_norm(it + 2), n3 = want_norm(it + 3);
                const bool any = n1 || n2 || n3;
                int rc = diff3_run3(ctx, Ht, cur, out, lazy_res ? nullptr : dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, dt,
                                    any ? (double*)pinned : nullptr);
                if (rc) return rc;
                if (any) FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
                int jstop = 0;
                if (fixed_iters <= 0) {
                    if (n1 && !(sqrt(pinned[0]) / sqrtN > tol)) jstop = 1;
                    else if (n2 && !(sqrt(pinned[1]) / sqrtN > tol)) jstop = 2;
                }
                if (jstop) {
                    err = sqrt(pinned[jstop - 1]) / sqrtN;
                    const int par0 = parity;
                    if (int rc2 = redo_singles(cur, par0, jstop, &cur)) return rc2;
                    stale_in = nullptr;
                    parity = par0 ^ (jstop & 1); swaps += jstop; it += jstop;
                    continue;
                }
                stale_in = lazy_res ? cur : nullptr;
                stale_out = out; stale_depth = 3;
                cur = out; parity ^= 1; swaps += 3; it += 3;
                if (n3) err = sqrt(pinned[2]) / sqrtN;
                else if (n2) err = sqrt(pinned[1]) / sqrtN;
                else if (n1) err = sqrt(pinned[0]) / sqrtN;
                continue;
            }
            if (fuse && parity == 0 && left >= 2 && fixed_iters <= 0 && ahead > 0) {
                // ---- a run of fused pairs enqueued `ahead` deep before the host looks at the norm of the oldest one ----
                struct Pair { const double* in; double* out; long it_base; int slot, seq; bool rec; } ring[FPR_CYC_SLOTS];
                hipStream_t s = ctx->stream[0];
                k_diff3_ctl_init<<<1, 1, 0, s>>>(ctx->cyc, tol, sqrtN);
                FPR_CHECK_LAUNCH(ctx);
                long it_enq = it;
                double* cur_enq = cur;
                int enq = 0, seen = 0, nrec = 0;
                bool stopped = false;
                while (true) {
                    while (enq - seen < 1 + ahead && iter_max - it_enq >= 2) {
                        double* out = (cur_enq == Htau) ? E1 : Htau;
                        const bool n1 = want_norm(it_enq + 1), n2 = want_norm(it_enq + 2);
                        int nparts = 0;
                        int rc = diff3_run2(ctx, Ht, cur_enq, Htau2, out, lazy_res ? nullptr : dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy,
                                            _dz, D_dx, D_dy, D_dz, nullptr, nullptr, dt, nullptr, false, 0, 0, 0, &ctx->cyc->stop,
                                            (n1 || n2) ? &nparts : nullptr);
                        if (rc) return rc;
                        Pair& p = ring[enq % FPR_CYC_SLOTS];
                        p.in = cur_enq; p.out = out; p.it_base = it_enq; p.rec = n1 || n2; p.slot = 0; p.seq = 0;
                        if (p.rec) {
                            p.slot = nrec % FPR_CYC_SLOTS;
                            p.seq = ++nrec;
                            __atomic_store_n(&ctx->cyc_h[p.slot].seq, 0, __ATOMIC_RELEASE);
                            k_diff3_check<<<1, 256, 0, s>>>(ctx->cyc, ctx->partials, ctx->partials + FPR_MAX_PARTIALS / 2, nullptr, 2, nparts,
                                                            n1 ? 1 : 0, n2 ? 1 : 0, 0, (int)it_enq, p.seq, &ctx->cyc_h[p.slot]);
                            FPR_CHECK_LAUNCH(ctx);
                        }
                        cur_enq = out; it_enq += 2; ++enq;
                    }
                    if (seen == enq) break;   // iter_max is less than two iterations away: the single-iteration path finishes
                    const Pair& p = ring[seen % FPR_CYC_SLOTS];
                    ++seen;
                    if (p.rec) {
                        FprCycleCtl rec;
                        if (int rc = fprx_cycle_wait(ctx, p.slot, p.seq, &rec)) return rc;
                        err = rec.rms;
                        if (rec.stop && rec.coarse_iters) {
                            // the reference stops after the FIRST iteration of this pair: redo that one iteration from the pair's
                            // input (intact: everything enqueued behind the pair has returned at once)
                            int rc = diff3_run(ctx, Ht, p.in, Htau2, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz,
                                               nullptr, nullptr, false, 0.0, nullptr, false, 0);
                            if (rc) return rc;
                            stale_in = nullptr;
                            cur = Htau2; parity = 1; ++swaps; it = p.it_base + 1;
                            stopped = true;
                            break;
                        }
                        stale_in = lazy_res ? p.in : nullptr;
                        stale_out = p.out; stale_depth = 2;
                        cur = p.out; swaps += 2; it = p.it_base + 2;
                        if (rec.stop) { stopped = true; break; }
                    } else {
                        stale_in = lazy_res ? p.in : nullptr;
                        stale_out = p.out; stale_depth = 2;
                        cur = p.out; swaps += 2; it = p.it_base + 2;
                    }
                }
                (void)stopped;   // err <= tol ends the loop through more(); otherwise iter_max is near and the code below takes over
                if (seen > 0) continue;
            }
            if (fuse && parity == 0 && left >= 2) {
                double* out = (cur == Htau) ? E1 : Htau;
                const bool n1 = want_norm(it + 1), n2 = want_norm(it + 2);
                int rc = diff3_run2(ctx, Ht, cur, Htau2, out, lazy_res ? nullptr : dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz,
                                    D_dx, D_dy, D_dz, nullptr, nullptr, dt, (n1 || n2) ? (double*)pinned : nullptr, false, 0);
                if (rc) return rc;
                stale_in = lazy_res ? cur : nullptr;
                stale_out = out; stale_depth = 2;
                if (n1 || n2) FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
                if (n1 && fixed_iters <= 0) {
                    const double e1 = sqrt(pinned[0]) / sqrtN;  // :191 after the first of the two iterations
                    if (!(e1 > tol)) {
                        // the reference stops here: redo that one iteration from the (intact) input
                        rc = diff3_run(ctx, Ht, cur, Htau2, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz,
                                       nullptr, nullptr, false, 0.0, nullptr, false, 0);
                        if (rc) return rc;
                        stale_in = nullptr;
                        cur = Htau2; parity = 1; ++swaps; ++it;
                        err = e1;
                        continue;
                    }
                    err = e1;
                }
                cur = out; swaps += 2; it += 2;
                if (n2) err = sqrt(pinned[1]) / sqrtN;
                continue;
            }
            // single iteration: even -> odd (into Htau2) or odd -> even (into Htau)
            double* out = parity ? Htau : Htau2;
            const bool need_norm = want_norm(it + 1);
            int rc = diff3_run(ctx, Ht, cur, out, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, nullptr,
                               nullptr, need_norm, dt, (double*)pinned, false, 0);
            if (rc) return rc;
            stale_in = nullptr;
            cur = out; parity ^= 1;  // :190
            ++swaps;
            if (need_norm) {
                FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
                err = sqrt(*pinned) / sqrtN;  // :191
            }
            ++it;
        }
        if (iters_host) iters_host[t] = it;
        if (err_host) err_host[t] = err;
        if (t == nt - 1 && stale_in) {   // the call ends on a fused launch that did not store its residual: replay it with the store
            int rc = stale_depth == 3
                         ? diff3_run3(ctx, Ht, stale_in, stale_out, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, dt, nullptr)
                         : diff3_run2(ctx, Ht, stale_in, Htau2, stale_out, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy,
                                      D_dz, nullptr, nullptr, dt, nullptr, false, 0);
            if (rc) return rc;
            stale_in = nullptr;
        }
        int rc = fpr_copy(ctx, Ht, cur, N);  // Ht .= Htau  :203
        if (rc) return rc;
    }
    if (cur == E1) {   // hand the field back in the caller's even buffer
        int rc = fpr_copy(ctx, Htau, E1, N);
        if (rc) return rc;
    }
    if (swapped_host) *swapped_host = (int)(swaps & 1);
    drain.ok = true;
    return FPR_OK;
}

extern "C" int fpr_diffusion3d_step2_box2(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid,
                                          double* Hout, double* dHdtau, int nx, int ny, int nz, double dtau, double _dt,
                                          double _dx, double _dy, double _dz, double D_dx, double D_dy, double D_dz,
                                          const int lo[3], const int hi[3], int zlo2, int zhi2, double scale,
                                          double* sumsq2_dev, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, lo && hi, "null box");
    FPR_REQUIRE(ctx, zlo2 >= hi[2] || zhi2 <= lo[2] || zhi2 <= zlo2, "the two z-ranges must not overlap");
    return diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi,
                      scale, sumsq2_dev, true, stream_sel, zlo2, zhi2);
}

// The CORE box of a decomposed run's fused pair, launched so that `reserve_cus` compute units stay without a workgroup
// of it: the shell launches and RCCL's send / receive kernels of the pair run beside it on the comm stream (role of
// @hide_communication, part1_kernel_programming.jl:185-188, for two iterations at once).  Same results as _step2_box.
extern "C" int fpr_diffusion3d_step2_core(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid,
                                          double* Hout, double* dHdtau, int nx, int ny, int nz, double dtau, double _dt,
                                          double _dx, double _dy, double _dz, double D_dx, double D_dy, double D_dz,
                                          const int lo[3], const int hi[3], double scale, double* sumsq2_dev, int stream_sel,
                                          int reserve_cus, int accumulate)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, lo && hi && reserve_cus >= 0, "null box / negative reserve");
    return diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi,
                      scale, sumsq2_dev, accumulate != 0, stream_sel, 0, 0, nullptr, nullptr, reserve_cus);
}

// ---- two iterations per launch BETWEEN ranks: the whole choreography of a fused pair in one call ---------------------------
// (GlobalGrid.step2 in finalprojectrepo.jl_amd/grid.py and diffusion_3D_step_τ2_halo! in julia/FPRHip.jl are its twins; the phased
// Python form stays for emulated ranks.)  SHELL = the one-cell layer of interior cells next to a face with a neighbour (disjoint
// thin boxes: z faces peeled first, then y, then x), CORE = the rest.  The device is split (fpr_reserve_comm_cus):
//   core stream : [x-slabs' first iteration] CORE as one fused launch ---------------------------------------------> (+ shell sums)
//   comm stream : z / y slabs' first iteration -> exchange(level 1 in Hmid) -> fused launches on the shell -> exchange(Hout) --^
// join = 0 leaves the pair on those two streams; the next pair continues from there, fpr_diffusion3d_join (or any entry point of
// the grid: fpr_halo_exchange3d*, fpr_allreduce_sum*) orders the compute stream behind it.
static int diff3_chain_work(const FprGrid& g, bool xstrips)
{
    // an x-slab in the field costs about five z-slabs (lanes along y, one cache line per access), a y-slab two; an x-shell in
    // compact strips (diffusion3d_xstrip.hpp) about as much as a z-slab
    const int w[3] = {xstrips ? 1 : 5, 2, 1};
    int work = 0;
    for (int f = 0; f < 6; ++f)
        if (g.nb[f] >= 0) work += w[f >> 1];
    return work;
}

static int diff3_comm_units(const FprGrid& g, bool xstrips)
{
    const int work = diff3_chain_work(g, xstrips);
    const bool anyx = g.nb[0] >= 0 || g.nb[1] >= 0;
    // Measured at 512^3 (tools/attic/dbg_faces2.py, two faces per dimension): z 16 units +7 %, 32 +9-11 %; y 16 +9.1 %, 24 +10.9 %,
    // 32 +12.2 %; yz 24 +12 %, 16 +12.6 %, 32 +14.2 %.  Shares that are no multiple of 32 use the unmasked core stream
    // (fpr_reserve_comm_cus).  With x-faces in strips (tools/exp_faces_units.py, one process): x, xy, one face per dimension: 16 best;
    // all six faces: 32 (the chain is as long as the core launch on 24).  With x-faces in the field the chain is nearly as long
    // as the core launch: 32, and 64 above two x-faces' worth of work.
    if (anyx && !xstrips) return work > 10 ? 64 : 32;
    if (anyx) return work >= 8 ? 32 : 16;
    return work > 4 ? 24 : 16;
}

// Strips of the x-faces with a neighbour: storage (kept in the context), per-face geometry.  boxes: the peeled shell boxes.
static int diff3_xstrips_setup(fpr_ctx* ctx, Diff3StripArgs& s, int nx, int ny, int nz, const int (*blo)[3], const int (*bhi)[3],
                               const int* bface, int nbx, int* nfaces_out, int* nblk_out)
{
    const long plane = (long)ny * nz;
    s.stride = ((plane + 511) & ~511L) + 64;
    s.nyt = (ny - 2 + 61) / 62;
    s.ntz = (nz - 2 + XS_ZC - 1) / XS_ZC;
    const int nblk = (int)(((long)s.nyt * s.ntz + 3) / 4);
    s.pcap = nblk;
    const size_t need = (size_t)2 * XS_COUNT * s.stride + (size_t)4 * nblk;
    if (ctx->xstrips_doubles < need) {
        // (a pending pair may still read the old strips: wait for it before they go)
        if (ctx->xstrips) { FPR_HIP(ctx, hipDeviceSynchronize()); FPR_HIP(ctx, hipFree(ctx->xstrips)); ctx->xstrips = nullptr; ctx->xstrips_doubles = 0; }
        FPR_HIP(ctx, hipMalloc(&ctx->xstrips, need * sizeof(double)));
        FPR_HIP(ctx, hipMemset(ctx->xstrips, 0, need * sizeof(double)));
        FPR_HIP(ctx, hipDeviceSynchronize());   // (a memset of device memory may return before it has run: not beside the first launches)
        ctx->xstrips_doubles = need;
        ctx->xs_field = nullptr;
    }
    int nf = 0;
    for (int b = 0; b < nbx; ++b) {
        if ((bface[b] >> 1) != 0) continue;
        Diff3StripFace& F = s.f[nf];
        F.high = bface[b] & 1;
        F.s = ctx->xstrips + (size_t)F.high * XS_COUNT * s.stride;
        F.xo = F.high ? nx - 2 : 1;
        F.jlo = blo[b][1]; F.jhi = bhi[b][1];
        F.klo = blo[b][2]; F.khi = bhi[b][2];
        ++nf;
    }
    if (nf == 1) s.f[1] = s.f[0];
    s.partials = ctx->xstrips + (size_t)2 * XS_COUNT * s.stride;
    *nfaces_out = nf;
    *nblk_out = nblk;
    return FPR_OK;
}

static int diff3_join(fpr_ctx* ctx, bool async)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (!ctx->pair_pending) return FPR_OK;
    ctx->pair_pending = false;
    if (async) {
        if (int rc = fpr_stream_wait(ctx, 0, 2)) return rc;   // the core launch (which has taken in the shell chain's exchanges)
        return fpr_stream_wait(ctx, 0, 1);                    // the pair's sums, finished on the comm stream
    }
    // The host waits for the two streams; the compute stream then needs no wait at all.  A wait parked on the compute stream while
    // pairs are still in flight is a third busy queue with a blocked barrier packet: every stream wait INSIDE the pairs then takes
    // longer, and a rank whose shell chain is nearly as long as its core launch (x-faces) ran its pairs 30 % slower for as long as
    // the wait was pending (tools/attic/dbg3.py: 942 -> 1207-1279 us per pair at 512^3, corner rank of (2,2,2)).  Callers of a join
    // read the sums or exchange halos next, i.e. wait for the pairs anyway.
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[2]));
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[1]));
    return FPR_OK;
}

extern "C" int fpr_diffusion3d_join(fpr_ctx* ctx) { return diff3_join(ctx, false); }

extern "C" int fpr_diffusion3d_step2_halo(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Hmid, double* Hout,
                                          double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy,
                                          double _dz, double D_dx, double D_dy, double D_dz, double scale, double* sumsq2_dev,
                                          int join)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, ctx->grid.on && nx == ctx->grid.n[0] && ny == ctx->grid.n[1] && nz == ctx->grid.n[2],
                "fpr_grid_init has not been called for arrays of this size");
    const FprGrid& g = ctx->grid;
    int mask = 0;
    for (int f = 0; f < 6; ++f)
        if (g.nb[f] >= 0) mask |= 1 << f;
    if (!mask) {   // no neighbour: the plain fused launch on the compute stream
        if (int rc = fpr_diffusion3d_join(ctx)) return rc;
        return diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, nullptr, nullptr,
                          scale, sumsq2_dev, false, 0);
    }
    const int n[3] = {nx, ny, nz};
    for (int f = 0; f < 6; ++f)
        FPR_REQUIRE(ctx, g.nb[f] < 0 || n[f >> 1] >= 8, "a decomposed dimension needs at least 8 cells for shell + core");
    // boundary boxes (0-based [lo, hi)) and the core
    int lo[3] = {1, 1, 1}, hi[3] = {nx - 1, ny - 1, nz - 1};
    int blo[6][3], bhi[6][3], bdim[6], bface[6], nbx = 0;
    for (int d = 2; d >= 0; --d)
        for (int side = 0; side < 2; ++side) {
            if (g.nb[2 * d + side] < 0 || hi[d] - lo[d] < 1) continue;
            for (int e = 0; e < 3; ++e) { blo[nbx][e] = lo[e]; bhi[nbx][e] = hi[e]; }
            if (side == 0) { bhi[nbx][d] = lo[d] + 1; lo[d] += 1; }
            else { blo[nbx][d] = hi[d] - 1; hi[d] -= 1; }
            bface[nbx] = 2 * d + side;
            bdim[nbx++] = d;
        }
    // x-faces: the shell column, its neighbours and the travelling planes in compact strips (diffusion3d_xstrip.hpp)
    bool xs_on = (g.nb[0] >= 0 || g.nb[1] >= 0) && fpr_opt(ctx, "diff3_xstrips", 1) != 0 && (long)ny * nz * 8 < 0x7ffffff0L &&
                       (long)nx * ny * 8 * 12 < (1L << 31);
    if (!xs_on) ctx->xs_field = nullptr;   // a pair in the field form may overwrite the array the strips were taken from
    const long k_opt = fpr_opt(ctx, "diff3_comm_units", 0);   // experiments: 8, 16, 32, 64
    int k = k_opt > 0 ? (int)k_opt : diff3_comm_units(g, xs_on);
    // a device already split with MORE units for the comm stream and a masked core stream (32: fpr_diffusion3d_step3_halo) serves this pair as
    // it is: a chain of triples that ends in a pair does not stop to split the device anew
    if (k_opt <= 0 && ctx->comm_cus > k && !ctx->core_unmasked) k = ctx->comm_cus;
    if (int rc = fpr_reserve_comm_cus(ctx, k)) return rc;
    double* sqs = ctx->scalars + 46;   // the shell chain's two sums (comm stream)
    // an error half way leaves launches on the core / comm streams: the compute stream is ordered behind both before the call
    // returns, so that nothing the caller enqueues next overtakes them (the outputs are undefined after an error, fpr.h)
    struct Rejoin {
        fpr_ctx* c; bool armed;
        ~Rejoin() { if (armed) { char msg[sizeof(c->err)]; memcpy(msg, c->err, sizeof(msg)); fpr_stream_wait(c, 0, 1); fpr_stream_wait(c, 0, 2); c->pair_pending = false; memcpy(c->err, msg, sizeof(msg)); } }
    } rejoin{ctx, true};
    const bool was_pending = ctx->pair_pending;
    if (!ctx->pair_pending) {   // (a pending pair left the comm stream behind its core launch and the core stream behind its chain)
        if (int rc = fpr_stream_wait(ctx, 1, 0)) return rc;   // fork: the pair's inputs are ready
        if (int rc = fpr_stream_wait(ctx, 2, 0)) return rc;
    }
#define D3ARGS nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz
    Diff3StripArgs xs{};
    int xs_faces = 0, xs_nblk = 0;
    const double* xsend[2] = {nullptr, nullptr};
    double* xrecv[2] = {nullptr, nullptr};
    if (xs_on) {
        if (int rc = diff3_xstrips_setup(ctx, xs, nx, ny, nz, blo, bhi, bface, nbx, &xs_faces, &xs_nblk)) return rc;
        xs.A = Htau; xs.Ht = Ht; xs.B = Hmid; xs.C = Hout; xs.dH = dHdtau;
        xs.nx = nx; xs.ny = ny; xs.nz = nz;
        xs.dtau = dtau; xs._dt = _dt; xs._dx = _dx; xs._dy = _dy; xs._dz = _dz; xs.D_dx = D_dx; xs.D_dy = D_dy; xs.D_dz = D_dz;
        xs.scale = scale;
        if (xs_faces == 0) xs_on = false;
    }
    const dim3 xs_cp((ny + 63) / 64, (nz + 3) / 4, xs_faces), xs_cpi((ny - 2 + 63) / 64, (nz - 2 + 3) / 4, xs_faces);
    double* xs_p1 = xs.partials;
    double* xs_p2 = xs_on ? xs.partials + (size_t)2 * xs_nblk : nullptr;
    if (xs_on) {
        // The strided accesses (one sector per row) stay out of the chain beside the core launch: one launch on the core stream
        // between two core launches.  The strips of this pair's level 0 are there already when the pair continues a chain: the
        // previous pair of this context is still pending (nobody can have touched the fields) and wrote this Htau from this Ht.
        int fm = 0;
        for (int q = 0; q < xs_faces; ++q) fm |= 1 << xs.f[q].high;
        const bool cont = was_pending && ctx->xs_field == Htau && ctx->xs_ht == Ht && ctx->xs_faces == fm && ctx->xs_n[0] == nx &&
                          ctx->xs_n[1] == ny && ctx->xs_n[2] == nz;
        if (!cont) {
            Diff3StripArgs ga = xs;
            ga.C = const_cast<double*>(Htau);
            k_xstrip_turn<false, true><<<xs_cp, dim3(64, 4), 0, ctx->stream[2]>>>(ga);
            FPR_CHECK_LAUNCH(ctx);
            if (int rc = fpr_stream_wait(ctx, 1, 2)) return rc;   // the chain reads the strips
        }
        ctx->xs_field = nullptr;   // (set again when this pair has been turned around)
    }
    // without strips: the x-slabs' first iteration on the core stream AHEAD of the core launch (25 us there, 180 us beside it)
    bool anyx = false;
    for (int b = 0; b < nbx; ++b)
        if (bdim[b] == 0 && !xs_on) {
            if (int rc = diff3_run(ctx, Ht, Htau, Hmid, dHdtau ? dHdtau : Hout, D3ARGS, blo[b], bhi[b], false, 0.0, nullptr, true, 2)) return rc;
            anyx = true;
        }
    if (anyx)
        if (int rc = fpr_stream_wait(ctx, 1, 2)) return rc;   // the chain follows them (recorded before the core launch)
    // The core launch leaves per-workgroup partials of its two sums in a buffer of its own (one per pair parity); they are
    // finished on the COMM stream once the core has ended -- on the core stream nothing stands between this core launch and the
    // next pair's (a finish and an add there cost 26 us per pair, profiles/r3_overlap_timeline_native.txt)
    if (sumsq2_dev && !ctx->core_partials)
        FPR_HIP(ctx, hipMalloc(&ctx->core_partials, (size_t)6 * FPR_CORE_PARTIALS * sizeof(double)));
    double* cpart = sumsq2_dev ? ctx->core_partials + (size_t)(ctx->pair_parity & 1) * 3 * FPR_CORE_PARTIALS : nullptr;
    ctx->pair_parity ^= 1;
    int core_nparts = 0;
    if (int rc = diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, D3ARGS, lo, hi, scale, nullptr, false, 2, 0, 0, nullptr,
                            sumsq2_dev ? &core_nparts : nullptr, k, cpart, FPR_CORE_PARTIALS)) return rc;
    if (sumsq2_dev)
        if (int rc = fpr_fill_on(ctx, sqs, 0.0, 2, 1)) return rc;
    if (xs_on) {   // level 1 of the x-shell columns: the planes that travel
        xs.partials = xs_p1;
        if (sumsq2_dev) k_diff3_xstrip<1, true><<<dim3(xs_nblk, xs_faces), 256, 0, ctx->stream[1]>>>(xs);
        else k_diff3_xstrip<1, false><<<dim3(xs_nblk, xs_faces), 256, 0, ctx->stream[1]>>>(xs);
        FPR_CHECK_LAUNCH(ctx);
        for (int q = 0; q < xs_faces; ++q) {
            xsend[xs.f[q].high] = xs.f[q].s + (size_t)XS_L1S * xs.stride;
            xrecv[xs.f[q].high] = xs.f[q].s + (size_t)XS_L1R * xs.stride;
        }
    }
    // (without a residual array the single-step launches leave their residual in the same cells of Hout: the fused launches on
    // these boxes overwrite them further down the chain)
    for (int b = 0; b < nbx; ++b)
        if (bdim[b] != 0)
            if (int rc = diff3_run(ctx, Ht, Htau, Hmid, dHdtau ? dHdtau : Hout, D3ARGS, blo[b], bhi[b], false, 0.0, nullptr, true, 1)) return rc;
    if (int rc = fprx_halo_exchange3d_comm_x(ctx, Hmid, nx, ny, nz, mask, xs_on ? xsend : nullptr, xs_on ? xrecv : nullptr)) return rc;
    if (xs_on && (mask & ~3)) {   // the y- / z-shell launches read the level-1 halo column from the field
        k_xstrip_frame<<<xs_cpi, dim3(64, 4), 0, ctx->stream[1]>>>(xs);
        FPR_CHECK_LAUNCH(ctx);
    }
    // fused launches on the shell boxes: their level-1 halo cells have just arrived in Hmid
    int b0 = 0;
    if (g.nb[4] >= 0 && g.nb[5] >= 0) {   // the two z-slabs (peeled first, same x / y extent) share one launch
        if (int rc = diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, D3ARGS, blo[0], bhi[0], scale, sumsq2_dev ? sqs : nullptr, true, 1,
                                blo[1][2], bhi[1][2])) return rc;
        b0 = 2;
    }
    for (int b = b0; b < nbx; ++b)
        if (bdim[b] != 0 || !xs_on)
            if (int rc = diff3_run2(ctx, Ht, Htau, Hmid, Hout, dHdtau, D3ARGS, blo[b], bhi[b], scale, sumsq2_dev ? sqs : nullptr, true, 1)) return rc;
#undef D3ARGS
    if (xs_on) {   // level 2 of the x-shell columns (level 1 of the halo column has arrived in its strip)
        xs.partials = xs_p2;
        if (sumsq2_dev) k_diff3_xstrip<2, true><<<dim3(xs_nblk, xs_faces), 256, 0, ctx->stream[1]>>>(xs);
        else k_diff3_xstrip<2, false><<<dim3(xs_nblk, xs_faces), 256, 0, ctx->stream[1]>>>(xs);
        FPR_CHECK_LAUNCH(ctx);
        if (sumsq2_dev)
            if (int rc = fprx_finish_sum2(ctx, xs_p1, xs_p2, xs_faces * xs_nblk, sqs, true, 1)) return rc;
        for (int q = 0; q < xs_faces; ++q) {
            xsend[xs.f[q].high] = xs.f[q].s + (size_t)XS_CS * xs.stride;
            xrecv[xs.f[q].high] = xs.f[q].s + (size_t)XS_CR * xs.stride;
        }
    }
    if (int rc = fprx_halo_exchange3d_comm_x(ctx, Hout, nx, ny, nz, mask, xs_on ? xsend : nullptr, xs_on ? xrecv : nullptr)) return rc;
    // Every stream wait between two core launches costs 5-8 us of the core stream's time (tools/attic/dbg_faces.py: with none
    // of them a z pair is +3.7 % over the plain launch, with these two +5.3 %, with a third at the top of the next call +5.9 %)
    if (int rc = fpr_stream_wait(ctx, 2, 1)) return rc;       // the core stream takes in the shell chain
    if (xs_on) {
        // the pair is turned around behind its core launch: the received halo column, the shell column and its residual go into
        // the fields, the columns next to the face come back as the next pair's level-0 strips
        // (the stores alone at the end of the chain, beside the core launch, and only the gather here: no gain for x or one
        // face per dimension, +10 % for xy -- tools/exp_faces_units.py, EXPERIMENTS 12.2)
        k_xstrip_turn<true, false><<<xs_cp, dim3(64, 4), 0, ctx->stream[2]>>>(xs);
        FPR_CHECK_LAUNCH(ctx);
        ctx->xs_field = Hout; ctx->xs_ht = Ht;
        ctx->xs_faces = 0;
        for (int q = 0; q < xs_faces; ++q) ctx->xs_faces |= 1 << xs.f[q].high;
        ctx->xs_n[0] = nx; ctx->xs_n[1] = ny; ctx->xs_n[2] = nz;
    }
    if (int rc = fpr_stream_wait(ctx, 1, 2)) return rc;       // the comm stream goes on behind the core launch: the next chain ...
    if (sumsq2_dev) {   // ... and the pair's sums = core partials + the shell chain's sums
        if (core_nparts > 0) {
            if (int rc = fprx_finish_sum2_plus(ctx, cpart, cpart + FPR_CORE_PARTIALS, core_nparts, sqs, sumsq2_dev, 1)) return rc;
        } else {   // empty core: the shell chain's sums are the pair's
            FPR_HIP(ctx, hipMemcpyAsync(sumsq2_dev, sqs, 2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream[1]));
        }
    }
    rejoin.armed = false;
    ctx->pair_pending = true;
    return join ? diff3_join(ctx, true) : FPR_OK;   // join = 1: the compute stream is ordered behind this pair (stream waits; the host goes on)
}

// ---- three iterations per launch BETWEEN ranks: z-slab decompositions (the process grid (1,1,N) of the scaling runs) -----------------
// Levels as in diffusion3d_fused3.hpp: L0 = X (halo planes valid), L1 / L2 never leave the chip in the CORE, L3 -> Y.  A cell's third iterate
// depends on L0 three planes away, so next to a z-face with a neighbour the two interior planes 1, 2 (nz-3, nz-2) -- the SHELL -- need the
// neighbour's L1 and L2 halo planes, which exist nowhere unless somebody stores them:
//   core stream : k_diff3_march3 on planes [3, nz-3) -- everything it needs of L1 / L2 it computes from X itself ----------------------------> join
//   comm stream : L1 on planes 1..4 (single-step launches into a 6-plane slab) -> exchange(L1 plane 1) -> L2 on planes 1..3 -> exchange(L2
//                 plane 1) -> L3 on planes 1, 2 into Y -> exchange(Y) ---------------------------------------------------------------------^
// (the same on the high side, mirrored).  One exchange per iteration, as in the reference (:185-188), each a single plane per face; the
// shell chain is ten thin launches beside a core launch of ~1 ms on the units the comm stream owns (fpr_reserve_comm_cus(32)).
// The slabs are 6-plane views: a single-step launch on a view is the reference's update on those planes (same diff3_point expression),
// so shell and core agree bit for bit with three exchanged single steps.  The x / y boundary cells of the L1 slabs are Y's (L1 lives in
// the reference's other buffer), those of the L2 slabs X's.
struct Diff3Ring {
    double* dst[4];
    const double* src[4];
    int nx, ny;
};

__global__ __launch_bounds__(256) void k_diff3_ring(Diff3Ring r)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int per = 2 * r.nx + 2 * (r.ny - 2);
    if (t >= per) return;
    int i, j;
    if (t < r.nx) { i = t; j = 0; }
    else if (t < 2 * r.nx) { i = t - r.nx; j = r.ny - 1; }
    else { const int q = t - 2 * r.nx; j = 1 + (q >> 1); i = (q & 1) ? r.nx - 1 : 0; }
    const size_t o = ((size_t)blockIdx.z * r.ny + j) * r.nx + i;
    r.dst[blockIdx.y][o] = r.src[blockIdx.y][o];
}

constexpr int S3_PLANES = 6;   // planes of a shell slab: halo, two shell planes, two planes of L1 / one of L2 the shell's L3 needs, one unused

extern "C" int fpr_diffusion3d_can_step3_halo(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hout,
                                              const double* dHdtau, int nx, int ny, int nz)
{
    if (!ctx) return FPR_ERR_INVALID;
    const FprGrid& g = ctx->grid;
    if (!g.on || nx != g.n[0] || ny != g.n[1] || nz != g.n[2]) return 0;
    for (int f = 0; f < 4; ++f)
        if (g.nb[f] >= 0) return 0;   // x / y neighbours: fused pairs (fpr_diffusion3d_step2_halo)
    if ((g.nb[4] >= 0 || g.nb[5] >= 0) && nz < 2 * S3_PLANES) return 0;
    return fpr_diffusion3d_can_step3(ctx, Ht, Htau, Hout, dHdtau, nx, ny, nz);
}

extern "C" int fpr_diffusion3d_step3_halo(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Hout, double* dHdtau, int nx,
                                          int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx,
                                          double D_dy, double D_dz, double scale, double* sumsq3_dev, int join)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, fpr_diffusion3d_can_step3_halo(ctx, Ht, Htau, Hout, dHdtau, nx, ny, nz) == 1,
                "three fused iterations between ranks need fpr_grid_init for arrays of this size, z-neighbours only, nz >= 12 and a problem "
                "the three-step kernel serves (fpr_diffusion3d_can_step3_halo)");
    const FprGrid& g = ctx->grid;
    const bool nbz[2] = {g.nb[4] >= 0, g.nb[5] >= 0};
#define D3ARGS dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz
    if (!nbz[0] && !nbz[1]) {   // no neighbour: the plain launch on the compute stream
        if (int rc = fpr_diffusion3d_join(ctx)) return rc;
        return diff3_run3(ctx, Ht, Htau, Hout, dHdtau, nx, ny, nz, D3ARGS, scale, sumsq3_dev);
    }
    const size_t pz = (size_t)nx * ny, slab = (size_t)S3_PLANES * pz;
    if (ctx->shell3_doubles < 6 * slab) {
        if (ctx->shell3) { FPR_HIP(ctx, hipDeviceSynchronize()); FPR_HIP(ctx, hipFree(ctx->shell3)); ctx->shell3 = nullptr; ctx->shell3_doubles = 0; }
        FPR_HIP(ctx, hipMalloc(&ctx->shell3, 6 * slab * sizeof(double)));
        FPR_HIP(ctx, hipMemset(ctx->shell3, 0, 6 * slab * sizeof(double)));
        FPR_HIP(ctx, hipDeviceSynchronize());
        ctx->shell3_doubles = 6 * slab;
    }
    const long k_opt = fpr_opt(ctx, "diff3_comm_units", 0);
    if (int rc = fpr_reserve_comm_cus(ctx, k_opt > 0 ? (int)k_opt : 32)) return rc;
    double* sqs = ctx->scalars + 50;   // the shell chain's three sums (comm stream)
    struct Rejoin {   // as in fpr_diffusion3d_step2_halo: an error half way must not leave launches the caller's next ones overtake
        fpr_ctx* c; bool armed;
        ~Rejoin() { if (armed) { char msg[sizeof(c->err)]; memcpy(msg, c->err, sizeof(msg)); fpr_stream_wait(c, 0, 1); fpr_stream_wait(c, 0, 2); c->pair_pending = false; memcpy(c->err, msg, sizeof(msg)); } }
    } rejoin{ctx, true};
    if (!ctx->pair_pending) {
        if (int rc = fpr_stream_wait(ctx, 1, 0)) return rc;   // fork: the triple's inputs are ready
        if (int rc = fpr_stream_wait(ctx, 2, 0)) return rc;
    }
    ctx->xs_field = nullptr;   // (a pair's x-strips, if any were kept, describe another field now)
    // ---- core ----
    if (sumsq3_dev && !ctx->core_partials)
        FPR_HIP(ctx, hipMalloc(&ctx->core_partials, (size_t)6 * FPR_CORE_PARTIALS * sizeof(double)));
    double* cpart = sumsq3_dev ? ctx->core_partials + (size_t)(ctx->pair_parity & 1) * 3 * FPR_CORE_PARTIALS : nullptr;
    ctx->pair_parity ^= 1;
    const int clo[3] = {1, 1, nbz[0] ? 3 : 1}, chi[3] = {nx - 1, ny - 1, nbz[1] ? nz - 3 : nz - 1};
    int core_nparts = 0;
    if (int rc = diff3_run3(ctx, Ht, Htau, Hout, dHdtau, nx, ny, nz, D3ARGS, scale, nullptr, nullptr, sumsq3_dev ? &core_nparts : nullptr, clo, chi,
                            2, cpart, FPR_CORE_PARTIALS)) return rc;
    // ---- shell chain ----
    if (sumsq3_dev)
        if (int rc = fpr_fill_on(ctx, sqs, 0.0, 3, 1)) return rc;
    double* const S1[2] = {ctx->shell3, ctx->shell3 + slab};
    double* const S2[2] = {ctx->shell3 + 2 * slab, ctx->shell3 + 3 * slab};
    double* const R[2] = {ctx->shell3 + 4 * slab, ctx->shell3 + 5 * slab};
    const size_t off[2] = {0, (size_t)(nz - S3_PLANES) * pz};
    {
        Diff3Ring r;
        for (int sd = 0; sd < 2; ++sd) {
            r.dst[sd] = S1[sd]; r.src[sd] = Hout + off[sd];
            r.dst[2 + sd] = S2[sd]; r.src[2 + sd] = Htau + off[sd];
        }
        r.nx = nx; r.ny = ny;
        k_diff3_ring<<<dim3((2 * nx + 2 * (ny - 2) + 255) / 256, 4, S3_PLANES), 256, 0, ctx->stream[1]>>>(r);
        FPR_CHECK_LAUNCH(ctx);
    }
    // a level on view planes [a, b) of side sd (low side: as given; high side: mirrored), its sum over the two shell planes only
    auto level = [&](int sd, const double* in, double* out, double* res, int a, int b, int lv) -> int {
        const int za = sd ? S3_PLANES - b : a, zb = sd ? S3_PLANES - a : b;        // mirrored box
        const int s0 = sd ? S3_PLANES - 3 : 1, s1 = sd ? S3_PLANES - 1 : 3;        // the shell planes of the view
        const int lo1[3] = {1, 1, s0}, hi1[3] = {nx - 1, ny - 1, s1};
        if (int rc = diff3_run(ctx, Ht + off[sd], in, out, res, nx, ny, S3_PLANES, D3ARGS, lo1, hi1, sumsq3_dev != nullptr, scale,
                               sumsq3_dev ? sqs + lv : nullptr, true, 1)) return rc;
        const int lo2[3] = {1, 1, sd ? za : s1}, hi2[3] = {nx - 1, ny - 1, sd ? s0 : zb};   // the planes beyond the shell
        if (lo2[2] < hi2[2])
            if (int rc = diff3_run(ctx, Ht + off[sd], in, out, res, nx, ny, S3_PLANES, D3ARGS, lo2, hi2, false, 0.0, nullptr, true, 1)) return rc;
        return FPR_OK;
    };
    auto exchange = [&](double* const P[2]) -> int {   // plane 1 of the low slab / plane 4 of the high slab travel; planes 0 / 5 receive
        const double* snd[2] = {P[0] + pz, P[1] + (size_t)(S3_PLANES - 2) * pz};
        double* rcv[2] = {P[0], P[1] + (size_t)(S3_PLANES - 1) * pz};
        return fprx_exchange_zplanes(ctx, snd, rcv);
    };
    for (int sd = 0; sd < 2; ++sd)
        if (nbz[sd])
            if (int rc = level(sd, Htau + off[sd], S1[sd], R[sd], 1, 5, 0)) return rc;
    if (int rc = exchange(S1)) return rc;
    for (int sd = 0; sd < 2; ++sd)
        if (nbz[sd])
            if (int rc = level(sd, S1[sd], S2[sd], R[sd], 1, 4, 1)) return rc;
    if (int rc = exchange(S2)) return rc;
    for (int sd = 0; sd < 2; ++sd)
        if (nbz[sd])
            if (int rc = level(sd, S2[sd], Hout + off[sd], dHdtau ? dHdtau + off[sd] : R[sd], 1, 3, 2)) return rc;
    if (int rc = fprx_halo_exchange3d_comm_x(ctx, Hout, nx, ny, nz, 0x30, nullptr, nullptr)) return rc;
#undef D3ARGS
    if (int rc = fpr_stream_wait(ctx, 2, 1)) return rc;       // the core stream takes in the shell chain
    if (int rc = fpr_stream_wait(ctx, 1, 2)) return rc;       // the comm stream goes on behind the core launch: the next chain ...
    if (sumsq3_dev) {   // ... and the triple's sums = core partials + the shell chain's sums
        if (core_nparts > 0) {
            if (int rc = fprx_finish_sum3(ctx, cpart, cpart + FPR_CORE_PARTIALS, cpart + 2 * FPR_CORE_PARTIALS, core_nparts, sumsq3_dev, 1, sqs)) return rc;
        } else {
            FPR_HIP(ctx, hipMemcpyAsync(sumsq3_dev, sqs, 3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream[1]));
        }
    }
    rejoin.armed = false;
    ctx->pair_pending = true;
    return join ? diff3_join(ctx, true) : FPR_OK;
}

extern "C" int fpr_diffusion3d_step_box(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2,
                                        double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx,
                                        double _dy, double _dz, double D_dx, double D_dy, double D_dz, const int lo[3],
                                        const int hi[3], double scale, double* sumsq_dev, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, lo && hi, "null box");
    return diff3_run(ctx, Ht, Htau, Htau2, dHdtau, nx, ny, nz, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi,
                     sumsq_dev != nullptr, scale, sumsq_dev, true, stream_sel);
}

// ------------------------------------------------------------------------------------------------
// split form: compute_flux! / compute_dHdtau! / update_H!   (part1_array_programming.jl:9-18)
// plain one-thread-per-element kernels; these are API-parity entry points, not the fast path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flux(double* __restrict__ q, const double* __restrict__ H, int nx, int ny,
                                               int qnx, int qny, int qnz, int dim, double D, double dd)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= qnx || j >= qny || k >= qnz) return;
    // @d_xi: difference along `dim`, inner (offset +1) in the other two dimensions
    const int hi = i + (dim == 0 ? 0 : 1), hj = j + (dim == 1 ? 0 : 1), hk = k + (dim == 2 ? 0 : 1);
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    const size_t id = (size_t)hi + sy * hj + sz * hk;
    const size_t st = dim == 0 ? 1 : (dim == 1 ? sy : sz);
    q[(size_t)i + (size_t)qnx * ((size_t)j + (size_t)qny * k)] = D * (H[id + st] - H[id]) / dd;
}

__global__ __launch_bounds__(256) void k_dHdtau(double* __restrict__ dH, const double* __restrict__ Htau,
                                                 const double* __restrict__ Ht, const double* __restrict__ qx,
                                                 const double* __restrict__ qy, const double* __restrict__ qz, int nx,
                                                 int ny, int nz, double dt, double dx, double dy, double dz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= nx - 2 || j >= ny - 2 || k >= nz - 2) return;
    const size_t qxi = (size_t)i + (size_t)(nx - 1) * ((size_t)j + (size_t)(ny - 2) * k);
    const size_t qyi = (size_t)i + (size_t)(nx - 2) * ((size_t)j + (size_t)(ny - 1) * k);
    const size_t qzi = (size_t)i + (size_t)(nx - 2) * ((size_t)j + (size_t)(ny - 2) * k);
    const size_t hid = (size_t)(i + 1) + (size_t)nx * ((size_t)(j + 1) + (size_t)ny * (k + 1));
    const double dqx = (qx[qxi + 1] - qx[qxi]) / dx;
    const double dqy = (qy[qyi + (size_t)(nx - 2)] - qy[qyi]) / dy;
    const double dqz = (qz[qzi + (size_t)(nx - 2) * (size_t)(ny - 2)] - qz[qzi]) / dz;
    const double tt = -(Htau[hid] - Ht[hid]) / dt;
    dH[qzi] = tt + ((dqx + dqy) + dqz);
}

__global__ __launch_bounds__(256) void k_updateH(double* __restrict__ Htau, const double* __restrict__ dH, int nx, int ny,
                                                  int nz, double dtau)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= nx - 2 || j >= ny - 2 || k >= nz - 2) return;
    const size_t di = (size_t)i + (size_t)(nx - 2) * ((size_t)j + (size_t)(ny - 2) * k);
    const size_t hid = (size_t)(i + 1) + (size_t)nx * ((size_t)(j + 1) + (size_t)ny * (k + 1));
    Htau[hid] = Htau[hid] + dH[di] * dtau;
}

static inline dim3 grid3(int a, int b, int c) { return dim3((a + 63) / 64, (b + 3) / 4, c); }

extern "C" int fpr_diffusion3d_flux(fpr_ctx* ctx, double* qx, double* qy, double* qz, const double* Htau, int nx, int ny,
                                    int nz, double D, double dx, double dy, double dz)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, qx && qy && qz && Htau, "null pointer");
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3 && nz >= 3, "grid must be at least 3^3");
    const dim3 b(64, 4, 1);
    k_flux<<<grid3(nx - 1, ny - 2, nz - 2), b, 0, ctx->stream[0]>>>(qx, Htau, nx, ny, nx - 1, ny - 2, nz - 2, 0, D, dx);
    k_flux<<<grid3(nx - 2, ny - 1, nz - 2), b, 0, ctx->stream[0]>>>(qy, Htau, nx, ny, nx - 2, ny - 1, nz - 2, 1, D, dy);
    k_flux<<<grid3(nx - 2, ny - 2, nz - 1), b, 0, ctx->stream[0]>>>(qz, Htau, nx, ny, nx - 2, ny - 2, nz - 1, 2, D, dz);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_diffusion3d_dHdtau(fpr_ctx* ctx, double* dHdtau, const double* Htau, const double* Ht,
                                      const double* qx, const double* qy, const double* qz, int nx, int ny, int nz,
                                      double dt, double dx, double dy, double dz)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, dHdtau && Htau && Ht && qx && qy && qz, "null pointer");
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3 && nz >= 3, "grid must be at least 3^3");
    k_dHdtau<<<grid3(nx - 2, ny - 2, nz - 2), dim3(64, 4, 1), 0, ctx->stream[0]>>>(dHdtau, Htau, Ht, qx, qy, qz, nx, ny,
                                                                                   nz, dt, dx, dy, dz);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_diffusion3d_update(fpr_ctx* ctx, double* Htau, const double* dHdtau, int nx, int ny, int nz,
                                      double dtau)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, Htau && dHdtau, "null pointer");
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3 && nz >= 3, "grid must be at least 3^3");
    k_updateH<<<grid3(nx - 2, ny - 2, nz - 2), dim3(64, 4, 1), 0, ctx->stream[0]>>>(Htau, dHdtau, nx, ny, nz, dtau);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// init_local_gaussian -- part1_utils.jl:1-12
__global__ __launch_bounds__(256) void k_gauss(double* __restrict__ H, int nx, int ny, int nz, double dx, double dy,
                                                double dz, double cx, double cy, double cz, long ox, long oy, long oz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= nx || j >= ny || k >= nz) return;
    const double x = (double)(ox + i) * dx, y = (double)(oy + j) * dy, z = (double)(oz + k) * dz;
    const double ax = x + dx / 2 - cx, ay = y + dy / 2 - cy, az = z + dz / 2 - cz;
    H[(size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k)] = 2 * exp(-1.0 * ((ax * ax + ay * ay) + az * az));
}

extern "C" int fpr_init_gaussian3d(fpr_ctx* ctx, double* H, int nx, int ny, int nz, double dx, double dy, double dz,
                                   double cx, double cy, double cz, int coordx, int coordy, int coordz)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, H, "null pointer");
    FPR_REQUIRE(ctx, nx >= 1 && ny >= 1 && nz >= 1, "empty grid");
    k_gauss<<<grid3(nx, ny, nz), dim3(64, 4, 1), 0, ctx->stream[0]>>>(H, nx, ny, nz, dx, dy, dz, cx, cy, cz,
                                                                     (long)coordx * (nx - 2), (long)coordy * (ny - 2),
                                                                     (long)coordz * (nz - 2));
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}
