// context.hip -- lifecycle, streams, reductions and flat-array utilities of libfpr_hip.so (gfx950).
#include <cstring>
#include "fpr_internal.hpp"

extern "C" const char* fpr_version(void) { return "fpr-hip 0.1 (gfx950, fp64, no-fma)"; }

extern "C" int fpr_ctx_create(fpr_ctx** out, int device, void* compute_stream, void* comm_stream)
{
    if (!out) return FPR_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FPR_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return FPR_ERR_INVALID;
    fpr_ctx* ctx = new fpr_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return FPR_ERR_HIP; }
    void* given[2] = {compute_stream, comm_stream};
    for (int s = 0; s < 2; ++s) {
        if (given[s]) {
            ctx->stream[s] = (hipStream_t)given[s];
        } else {
            if (hipStreamCreateWithFlags(&ctx->stream[s], hipStreamNonBlocking) != hipSuccess) { delete ctx; return FPR_ERR_HIP; }
            ctx->own_stream[s] = true;
        }
        if (hipEventCreateWithFlags(&ctx->ev[s], hipEventDisableTiming) != hipSuccess) { delete ctx; return FPR_ERR_HIP; }
    }
    ctx->stream[2] = ctx->stream[0];
    if (hipEventCreateWithFlags(&ctx->ev[2], hipEventDisableTiming) != hipSuccess) { delete ctx; return FPR_ERR_HIP; }
    bool ok = hipMalloc(&ctx->partials, (FPR_MAX_PARTIALS + 256) * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->partials2, (FPR_MAX_PARTIALS + 256) * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->scalars, 64 * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->state, sizeof(FprSolveState)) == hipSuccess &&
              hipHostMalloc(&ctx->state_h, sizeof(FprSolveState)) == hipSuccess &&
              hipHostMalloc(&ctx->host_scalars, 64 * sizeof(double)) == hipSuccess &&
              hipMalloc(&ctx->cyc, sizeof(FprCycleCtl)) == hipSuccess &&
              hipHostMalloc(&ctx->cyc_h, FPR_CYC_SLOTS * sizeof(FprCycleCtl), hipHostMallocCoherent) == hipSuccess;
    if (!ok) { fpr_ctx_destroy(ctx); return FPR_ERR_HIP; }
    hipMemset(ctx->scalars, 0, 64 * sizeof(double));
    hipMemset(ctx->state, 0, sizeof(FprSolveState));
    *out = ctx;
    return FPR_OK;
}

extern "C" int fpr_ctx_destroy(fpr_ctx* ctx)
{
    if (!ctx) return FPR_OK;
    hipSetDevice(ctx->device);
    for (int s = 0; s < 2; ++s)
        if (ctx->stream[s]) hipStreamSynchronize(ctx->stream[s]);
    fpr_comm_finalize(ctx);
    fpr_reserve_comm_cus(ctx, 0);
    for (auto& kv : ctx->arenas)
        for (auto& L : kv.second) {
            if (L.tmp && L.own_tmp) hipFree(L.tmp);
            if (L.res_c && L.own_coarse) hipFree(L.res_c);
            if (L.corr_c && L.own_coarse) hipFree(L.corr_c);
            if (L.tmp2 && L.own_tmp2) hipFree(L.tmp2);
            if (L.corr_c2 && L.own_coarse) hipFree(L.corr_c2);
            for (double* q : L.coarse_own)
                if (q) hipFree(q);
        }
    for (auto& e : ctx->ktimer_ev) hipEventDestroy(e);
    if (ctx->cg_buf) hipFree(ctx->cg_buf);
    if (ctx->partials) hipFree(ctx->partials);
    if (ctx->partials2) hipFree(ctx->partials2);
    if (ctx->core_partials) hipFree(ctx->core_partials);
    if (ctx->shell3) hipFree(ctx->shell3);
    if (ctx->tickets) hipFree(ctx->tickets);
    if (ctx->xstrips) hipFree(ctx->xstrips);
    if (ctx->ns_ev) hipEventDestroy(ctx->ns_ev);
    fprx_ns_worker_free(ctx);
    if (ctx->reserved_map) hipFree(ctx->reserved_map);
    if (ctx->scalars) hipFree(ctx->scalars);
    if (ctx->state) hipFree(ctx->state);
    if (ctx->state_h) hipHostFree(ctx->state_h);
    if (ctx->host_scalars) hipHostFree(ctx->host_scalars);
    if (ctx->cyc) hipFree(ctx->cyc);
    if (ctx->cyc_h) hipHostFree(ctx->cyc_h);
    for (int s = 0; s < 2; ++s) {
        if (ctx->ev[s]) hipEventDestroy(ctx->ev[s]);
        if (ctx->own_stream[s] && ctx->stream[s]) hipStreamDestroy(ctx->stream[s]);
    }
    if (ctx->ev[2]) hipEventDestroy(ctx->ev[2]);
    delete ctx;
    return FPR_OK;
}

extern "C" int fpr_synchronize(fpr_ctx* ctx)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[1]));
    if (ctx->stream[2] != ctx->stream[0]) FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[2]));
    // all three streams are drained: a fused pair that fpr_diffusion3d_step2_halo(join = 0) left on the core / comm streams is joined
    // (the next pair must fork from the compute stream again, behind whatever the caller enqueues there in between)
    ctx->pair_pending = false;
    return FPR_OK;
}

// Split the device between two library-owned streams (hipExtStreamCreateWithCUMask): stream 1 (comm) may use k compute
// units, stream 2 (core) all the others.  Why masks and not just a smaller grid: a workgroup of a second queue is dealt
// to a shader engine and WAITS there when that engine has no room, even with idle units elsewhere
// (tools/cu_share_probe.hip, profiles/r3_cu_share_probe.txt: beside a launch that fills 244 of 256 units three quarters
// of a 128-workgroup guest kernel started only when the launch ended; with masks every workgroup started at once).
// The low k mask bits go to the comm stream: on gfx950 bit b is unit (b / 8 / 4) of shader engine (b / 8) % 4 of XCD b % 8,
// so they spread over the XCDs first, then over the engines.  k = 0: back to the caller's streams (stream 2 = stream 0).
// marks the compute unit every workgroup runs on (key of fpr_cu_key) and stays for a few microseconds, so that a launch of many
// workgroups visits every unit of its stream's mask
__global__ __launch_bounds__(64) void k_cu_probe(unsigned* __restrict__ map)
{
    if (threadIdx.x == 0) {
        const unsigned key = fpr_cu_key();
        atomicOr(map + (key >> 5), 1u << (key & 31));
        for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(127);
    }
}

extern "C" int fpr_reserve_comm_cus(fpr_ctx* ctx, int k)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (k == ctx->comm_cus || (k > 0 && k == ctx->comm_cus_asked)) return FPR_OK;
    FPR_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->ncu <= 0) {
        int v = 0;
        ctx->ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && v > 0) ? v : 256;
    }
    FPR_REQUIRE(ctx, k >= 0 && k <= ctx->ncu / 2, "0 <= k <= half the compute units");
    FPR_REQUIRE(ctx, k % 8 == 0 || ctx->ncu != 256, "k must be a multiple of 8 (the same number of units out of every XCD)");
    if (ctx->comm_cus > 0) {   // drop the library's streams
        for (int s = 0; s < 3; ++s) FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[s]));
        ctx->stream[1] = ctx->caller_comm;
        ctx->stream[2] = ctx->stream[0];
        for (int m = 0; m < 2; ++m) {
            if (ctx->masked[m]) hipStreamDestroy(ctx->masked[m]);
            ctx->masked[m] = nullptr;
        }
        ctx->comm_cus = ctx->comm_cus_asked = 0;
        ctx->core_unmasked = false;
        ctx->pair_pending = false;   // drained above: nothing is left on the streams that have just been destroyed
    }
    if (k == 0) return FPR_OK;
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[1]));
    // Two forms of the split.  A workgroup is dealt to a shader engine and waits THERE for room, so a core stream whose mask takes
    // different numbers of units out of the engines (32 engines of 8 units on MI355X) leaves working units without a workgroup
    // of a launch that has one per unit while workgroups wait elsewhere (profiles/r3_cu_share_probe.txt).
    //  (a) k a multiple of 32: the comm stream gets the low k mask bits (bit b = unit b / 32 of engine (b / 8) % 4 of XCD b % 8:
    //      one unit of every engine per 32), the core stream all the others.
    //  (b) any other multiple of 8 (16 for a rank with z-faces only): the comm stream as in (a), the core stream EVERY unit; a
    //      probe launch on the comm stream records which units that stream really has (key XCC_ID | HW_ID[15:8]), and the
    //      workgroups of a core launch that find themselves on one of them leave at once (Diff3Args2::reserved).  If the probe
    //      does not find exactly k units, k is rounded up to the next multiple of 32 and form (a) is used.
    const int words = (ctx->ncu + 31) / 32;
    const int asked = k;
    const long force = -1;    // experiments: 0 = never (b), 1 = (b) for every k
    bool unmasked = force == 1 || (force != 0 && k % 32 != 0);
    // every failure below leaves the context unsplit and owns no stream (comm_cus stays 0)
    auto undo = [&](hipError_t e, const char* what) {
        for (int m = 0; m < 2; ++m) {
            if (ctx->masked[m]) hipStreamDestroy(ctx->masked[m]);
            ctx->masked[m] = nullptr;
        }
        return fpr_fail(ctx, FPR_ERR_HIP, "fpr_reserve_comm_cus: %s: %s", what, hipGetErrorString(e));
    };
#define FPR_SPLIT(call)                                  \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return undo(e_, #call);    \
    } while (0)
    for (int attempt = 0; attempt < 2; ++attempt) {
        std::vector<uint32_t> mc(words, 0u), mr(words, 0u);
        for (int b = 0; b < ctx->ncu; ++b) (b < k ? mc : mr)[b / 32] |= 1u << (b % 32);
        FPR_SPLIT(hipExtStreamCreateWithCUMask(&ctx->masked[0], (uint32_t)words, mc.data()));
        if (unmasked) {
            if (!ctx->reserved_map) FPR_SPLIT(hipMalloc(&ctx->reserved_map, 128 * sizeof(unsigned)));   // [64, 128): scratch of the tests
            unsigned host[64];
            int found = 0;
            for (int probe = 0; probe < 3 && found != k; ++probe) {
                FPR_SPLIT(hipMemsetAsync(ctx->reserved_map, 0, sizeof(host), ctx->masked[0]));
                k_cu_probe<<<64 * k, 64, 0, ctx->masked[0]>>>(ctx->reserved_map);
                FPR_SPLIT(hipGetLastError());
                FPR_SPLIT(hipMemcpyAsync(host, ctx->reserved_map, sizeof(host), hipMemcpyDeviceToHost, ctx->masked[0]));
                FPR_SPLIT(hipStreamSynchronize(ctx->masked[0]));
                found = 0;
                for (unsigned w : host) found += __builtin_popcount(w);
            }
            ctx->options["comm_units_found"] = found;
            if (found == k) {
                FPR_SPLIT(hipStreamCreateWithFlags(&ctx->masked[1], hipStreamNonBlocking));
                ctx->core_unmasked = true;
                break;
            }
            hipStreamDestroy(ctx->masked[0]);      // form (a) with the next multiple of 32
            ctx->masked[0] = nullptr;
            unmasked = false;
            k = (k + 31) / 32 * 32;
            if (k > ctx->ncu / 2) return fpr_fail(ctx, FPR_ERR_INVALID, "the comm stream's units could not be identified and k cannot be rounded up");
            continue;
        }
        if (!(k % 32 == 0 || ctx->ncu != 256)) {
            undo(hipSuccess, "mask");
            return fpr_fail(ctx, FPR_ERR_INVALID, "k must be a multiple of 32 for a masked core stream");
        }
        FPR_SPLIT(hipExtStreamCreateWithCUMask(&ctx->masked[1], (uint32_t)words, mr.data()));
        ctx->core_unmasked = false;
        break;
    }
#undef FPR_SPLIT
    ctx->comm_cus_asked = asked;
    ctx->caller_comm = ctx->stream[1];
    ctx->stream[1] = ctx->masked[0];
    ctx->stream[2] = ctx->masked[1];
    ctx->comm_cus = k;
    return FPR_OK;
}

extern "C" int fpr_comm_cus(fpr_ctx* ctx) { return ctx ? ctx->comm_cus : -1; }

// hipStream_t behind a selector (0 compute, 1 comm, 2 core), for host code that enqueues its own work on them
extern "C" int fpr_stream_handle(fpr_ctx* ctx, int sel, void** out)
{
    if (!ctx || !out || sel < 0 || sel > 2) return FPR_ERR_INVALID;
    *out = (void*)ctx->stream[sel];
    return FPR_OK;
}

extern "C" const char* fpr_last_error(fpr_ctx* ctx) { return ctx ? ctx->err : "null context"; }

// Every switch the library reads (DESIGN 6 lists what each one does).  Tuning knobs of earlier rounds are compile-time constants now; a key
// that is not in this list is an error, not a silent no-op.
static const char* const FPR_OPTION_KEYS[] = {
    "diff3_fuse2", "diff3_fuse3", "diff3_ahead", "diff3_lazy_residual", "diff3_xstrips", "diff3_comm_units", "fp_contract",
    "mg_multi", "mg_seam", "mg_mid", "mg_small", "mg_patch", "mg_fuse_restrict", "mg_fuse_prolong", "mg_zero_guess", "mg_zero_fuse",
    "mg_fold_finish", "mg_fold_fsq", "mg_ahead", "mg_seam_history", "mg_jacobi_persist", "cg_fused", "handoff_fences", "ns_pipeline",
    "grid_drop_faces",                                                                    // measurement aid (bench.py's neighbour legs)
    "diff3_bal_g", "diff3_reserved_test", "diff3_zc2", "mg_jacobi_persist_test_abort", "mg_seam_predict",   // test hooks
    "cg_persistent_timeouts", "mg_jacobi_persist_timeouts"};                              // diagnostics the library counts (resettable)

extern "C" int fpr_set_option(fpr_ctx* ctx, const char* key, long value)
{
    if (!ctx || !key) return FPR_ERR_INVALID;
    bool known = false;
    for (const char* k : FPR_OPTION_KEYS) known = known || std::strcmp(k, key) == 0;
    if (!known) return fpr_fail(ctx, FPR_ERR_INVALID, "unknown option '%s'", key);
    ctx->options[key] = value;
    // asking for the persistent Jacobi coarse solve again lifts the switch a timed-out hand-off left behind (multigrid2d.hip)
    if (value != 0 && std::string(key) == "mg_jacobi_persist") ctx->jacp_resident = -1;
    return FPR_OK;
}

extern "C" long fpr_get_option(fpr_ctx* ctx, const char* key)
{
    if (!ctx || !key) return 0;
    return fpr_opt(ctx, key, 0);
}

extern "C" int fpr_kernel_timer(fpr_ctx* ctx, int enable)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (enable && ctx->ktimer_ev.empty()) {
        ctx->ktimer_ev.resize(2 * 8192);
        ctx->ktimer_kind.assign(8192, -1);
        // timing only: no system-scope fence at the record (default events write back and invalidate the caches there; measured
        // on the fused kernel: no difference, EXPERIMENTS 12.4 -- the flag is what the HIP headers prescribe for timing events)
        for (auto& e : ctx->ktimer_ev) FPR_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    }
    ctx->ktimer_on = enable != 0;
    if (enable) ctx->ktimer_used = 0;
    return FPR_OK;
}

extern "C" int fpr_kernel_timer_read(fpr_ctx* ctx, int kind, double* total_ms_host, long* count_host)
{
    if (!ctx || !total_ms_host || !count_host) return FPR_ERR_INVALID;
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[1]));
    if (ctx->stream[2] != ctx->stream[0]) FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[2]));
    double tot = 0.0;
    long cnt = 0;
    for (size_t i = 0; i + 1 < ctx->ktimer_used; i += 2) {
        if (kind >= 0 && ctx->ktimer_kind[i / 2] != kind) continue;
        float ms = 0.f;
        FPR_HIP(ctx, hipEventElapsedTime(&ms, ctx->ktimer_ev[i], ctx->ktimer_ev[i + 1]));
        tot += ms;
        ++cnt;
    }
    *total_ms_host = tot;
    *count_host = cnt;
    return FPR_OK;
}

extern "C" int fpr_stream_wait(fpr_ctx* ctx, int waiter, int signaller)
{
    if (!ctx || waiter < 0 || waiter > 2 || signaller < 0 || signaller > 2) return FPR_ERR_INVALID;
    if (ctx->stream[waiter] == ctx->stream[signaller]) return FPR_OK;
    FPR_HIP(ctx, hipEventRecord(ctx->ev[signaller], ctx->stream[signaller]));
    FPR_HIP(ctx, hipStreamWaitEvent(ctx->stream[waiter], ctx->ev[signaller], 0));
    return FPR_OK;
}

// ---------------------------------------------------------------------------------------------
// flat streaming kernels: 16 B per lane, grid-stride, grid capped at 8 blocks per CU
// ---------------------------------------------------------------------------------------------
static inline int flat_grid(size_t nvec)
{
    size_t b = (nvec + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ __launch_bounds__(256) void k_copy(double* __restrict__ dst, const double* __restrict__ src, size_t n)
{
    const size_t n2 = n / 2;
    const size_t stride = (size_t)gridDim.x * 256;
    const bool aligned = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
    if (aligned) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride)
            reinterpret_cast<double2*>(dst)[i] = reinterpret_cast<const double2*>(src)[i];
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(256) void k_fill(double* __restrict__ dst, double v, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

// u .= u - corr   (multigrid.jl:139)
__global__ __launch_bounds__(256) void k_axmy(double* __restrict__ u, const double* __restrict__ corr, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) u[i] = u[i] - corr[i];
}

template <int MODE>  // 0: sum((x*scale)^2)   1: sum(x*y)   2: max(|x|)
__global__ __launch_bounds__(256) void k_reduce(const double* __restrict__ x, const double* __restrict__ y, size_t n,
                                                 double scale, double* __restrict__ partials)
{
    __shared__ double red[16];
    const size_t stride = (size_t)gridDim.x * 256;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if constexpr (MODE == 0) { const double t = x[i] * scale; acc += t * t; }
        else if constexpr (MODE == 1) acc += x[i] * y[i];
        else acc = fmax(acc, fabs(x[i]));
    }
    if constexpr (MODE == 2) acc = fpr_block_max<256>(acc, red);
    else acc = fpr_block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// Dot2 (fpr_internal.hpp): block partials as (s, e) pairs, then one block sums the pairs and rounds s + e once
__global__ __launch_bounds__(256) void k_dot2(const double* __restrict__ x, const double* __restrict__ y, size_t n,
                                               double* __restrict__ partials)
{
    __shared__ double red[32];
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0.0, e = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) fpr_s2_add_prod(s, e, x[i], y[i]);
    fpr_block_sum_s2<256>(s, e, red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s; partials[2 * blockIdx.x + 1] = e; }
}

__global__ __launch_bounds__(256) void k_finish_s2(const double* __restrict__ partials, int n, double* __restrict__ out)
{
    __shared__ double red[32];
    double s, e;
    fpr_sum_partials_256_s2(partials, n, red, s, e);
    if (threadIdx.x == 0) out[0] = s + e;
}

template <int MODE>  // 0 store sum, 1 accumulate sum, 2 store max
__global__ __launch_bounds__(256) void k_finish(const double* __restrict__ partials, int n, double* __restrict__ out)
{
    __shared__ double red[16];
    if constexpr (MODE == 2) {
        double m = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, partials[i]);
        m = fpr_block_max<256>(m, red);
        if (threadIdx.x == 0) out[0] = m;
    } else {
        const double s = fpr_sum_partials_256(partials, n, red);
        if (threadIdx.x == 0) out[0] = (MODE == 1) ? out[0] + s : s;
    }
}

// first stage for long partial lists: block b sums the slice [b*per, (b+1)*per) in a fixed order
__global__ __launch_bounds__(256) void k_fold_partials(const double* __restrict__ partials, int n, int per,
                                                        double* __restrict__ out)
{
    __shared__ double red[16];
    const int lo = blockIdx.x * per;
    int hi = lo + per;
    if (hi > n) hi = n;
    double s = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += partials[i];
    s = fpr_block_sum<256>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

int fprx_finish_sum(fpr_ctx* ctx, const double* partials, int nparts, double* out_dev, bool accumulate, int stream_sel)
{
    if (nparts > 2048) {
        // two-stage finish (deterministic): 128 slices folded in parallel, then the usual single block.
        // The folded values live behind the partial list itself (the buffers hold FPR_MAX_PARTIALS + 256).
        const int nb = 128;
        const int per = (nparts + nb - 1) / nb;
        double* fold = (stream_sel == 1 ? ctx->partials2 : ctx->partials) + FPR_MAX_PARTIALS;
        k_fold_partials<<<nb, 256, 0, ctx->stream[stream_sel]>>>(partials, nparts, per, fold);
        partials = fold;
        nparts = nb;
    }
    if (accumulate) k_finish<1><<<1, 256, 0, ctx->stream[stream_sel]>>>(partials, nparts, out_dev);
    else k_finish<0><<<1, 256, 0, ctx->stream[stream_sel]>>>(partials, nparts, out_dev);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// ---- cycles enqueued ahead (FprCycleCtl, fpr_internal.hpp) ------------------------------------------------
// start of a solve: sum(f.^2) finished exactly as k_finish<0> does it, f_rms and the loop's threshold; the cycle state reset
__global__ __launch_bounds__(256) void k_cycle_init(const double* __restrict__ partials, int n, FprCycleCtl* ctl,
                                                     FprSolveState* st, double tol, double npoints)
{
    __shared__ double red[16];
    const double s = fpr_sum_partials_256(partials, n, red);
    if (threadIdx.x == 0) {
        ctl->stop = 0; ctl->ncycles = 0; ctl->coarse_iters = 0; ctl->seq = 0;
        const double frms = sqrt(s / npoints);   // multigrid.jl:53
        ctl->frms = frms;
        ctl->tolf = tol * frms;                  // :70
        ctl->rms = 0.0;
        st->acc_iters = 0;
    }
}

// end of a V-cycle: sum(res.^2) of the last post-smoothing sweep exactly as k_finish<0> sums it, r_rms
// (multigrid.jl:252) and the loop's exit test (:70) on the device
__global__ __launch_bounds__(256) void k_cycle_finish(FprFinishArgs a)
{
    __shared__ double red[16];
    fpr_cycle_finish_body(a, red);
}

int fprx_cycle_init(fpr_ctx* ctx, const double* f, size_t n, double tol)
{
    ctx->fin = FprFinishArgs{};   // nothing is handed over from an earlier solve
    // as fprx_sumsq_scaled_dev(f) + the finishing launch, which here also sets up the cycle state (one launch less)
    int g = flat_grid(n);
    const double* partials = ctx->partials;
    k_reduce<0><<<g, 256, 0, ctx->stream[0]>>>(f, nullptr, n, 1.0, ctx->partials);
    FPR_CHECK_LAUNCH(ctx);
    if (g > 2048) {
        const int nb = 128;
        const int per = (g + nb - 1) / nb;
        double* fold = ctx->partials + FPR_MAX_PARTIALS;
        k_fold_partials<<<nb, 256, 0, ctx->stream[0]>>>(partials, g, per, fold);
        partials = fold;
        g = nb;
    }
    k_cycle_init<<<1, 256, 0, ctx->stream[0]>>>(partials, g, ctx->cyc, ctx->state, tol, (double)n);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// the same in two steps: the cycle state reset before the first launch of a solve (f_rms = 0 for the moment), f_rms and the threshold from
// block partials of sum(f.^2) that the solve's first pass over the finest grid has left (k_smooth2_march_v2<..., FSQ>)
int fprx_cycle_reset(fpr_ctx* ctx, double tol)
{
    ctx->fin = FprFinishArgs{};
    k_cycle_init<<<1, 256, 0, ctx->stream[0]>>>(ctx->partials, 0, ctx->cyc, ctx->state, tol, 1.0);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

int fprx_cycle_init_from(fpr_ctx* ctx, const double* partials, int nparts, size_t n, double tol)
{
    if (nparts > 2048) {
        const int nb = 128;
        const int per = (nparts + nb - 1) / nb;
        double* fold = ctx->partials + FPR_MAX_PARTIALS;
        k_fold_partials<<<nb, 256, 0, ctx->stream[0]>>>(partials, nparts, per, fold);
        partials = fold;
        nparts = nb;
    }
    k_cycle_init<<<1, 256, 0, ctx->stream[0]>>>(partials, nparts, ctx->cyc, ctx->state, tol, (double)n);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// Host side of the record: wait until the cycle with sequence number `seq` has reported into `slot`.  Polls the pinned
// record; every now and then it asks the stream whether it has drained or failed, so that a launch that never ran
// cannot make this spin for ever.
int fprx_cycle_wait(fpr_ctx* ctx, int slot, int seq, FprCycleCtl* out)
{
    FprCycleCtl* rec = &ctx->cyc_h[slot];
    unsigned long spins = 0;
    while (__atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xffff) == 0) {
            const hipError_t q = hipStreamQuery(ctx->stream[0]);
            if (q == hipSuccess) {
                if (__atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) == seq) break;
                return fpr_fail(ctx, FPR_ERR_HIP, "V-cycle %d never reported (stream idle)", seq);
            }
            if (q != hipErrorNotReady) return fpr_fail(ctx, FPR_ERR_HIP, "HIP error while waiting for V-cycle %d: %s", seq, hipGetErrorString(q));
        }
    }
    *out = *rec;
    return FPR_OK;
}

static FprFinishArgs finish_args(fpr_ctx* ctx, const double* partials, int nparts, double* sumsq_out_dev, double npoints, int slot)
{
    FprFinishArgs a;
    a.partials = partials; a.n = nparts; a.out = sumsq_out_dev; a.npoints = npoints;
    a.ctl = ctx->cyc; a.st = ctx->state; a.rec_host = &ctx->cyc_h[slot];
    return a;
}

int fprx_cycle_finish(fpr_ctx* ctx, const double* partials, int nparts, double* sumsq_out_dev, double npoints, int slot)
{
    if (int rc = fprx_cycle_finish_flush(ctx)) return rc;           // (none can be pending here; order kept if one ever is)
    __atomic_store_n(&ctx->cyc_h[slot].seq, 0, __ATOMIC_RELEASE);   // before the launch that will report into it
    if (nparts > 2048) {   // as fprx_finish_sum; folding stale partials in a skipped cycle only touches scratch
        const int nb = 128;
        const int per = (nparts + nb - 1) / nb;
        double* fold = ctx->partials + FPR_MAX_PARTIALS;
        k_fold_partials<<<nb, 256, 0, ctx->stream[0]>>>(partials, nparts, per, fold);
        partials = fold;
        nparts = nb;
    }
    k_cycle_finish<<<1, 256, 0, ctx->stream[0]>>>(finish_args(ctx, partials, nparts, sumsq_out_dev, npoints, slot));
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

int fprx_cycle_finish_defer(fpr_ctx* ctx, const double* partials, int nparts, double* sumsq_out_dev, double npoints, int slot)
{
    if (nparts > 2048 || !fpr_opt(ctx, "mg_fold_finish", FPR_FOLD_FINISH_DEFAULT))
        return fprx_cycle_finish(ctx, partials, nparts, sumsq_out_dev, npoints, slot);
    if (int rc = fprx_cycle_finish_flush(ctx)) return rc;
    __atomic_store_n(&ctx->cyc_h[slot].seq, 0, __ATOMIC_RELEASE);
    ctx->fin = finish_args(ctx, partials, nparts, sumsq_out_dev, npoints, slot);
    return FPR_OK;
}

int fprx_cycle_finish_flush(fpr_ctx* ctx)
{
    if (!ctx->fin.partials) return FPR_OK;
    const FprFinishArgs a = ctx->fin;
    ctx->fin = FprFinishArgs{};
    k_cycle_finish<<<1, 256, 0, ctx->stream[0]>>>(a);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// two partial lists of equal length -> out[0], out[1] in ONE launch (block b sums list b; fixed order)
template <int ACC>
__global__ __launch_bounds__(256) void k_finish2(const double* __restrict__ p0, const double* __restrict__ p1, int n,
                                                  double* __restrict__ out)
{
    __shared__ double red[16];
    const double s = fpr_sum_partials_256(blockIdx.x ? p1 : p0, n, red);
    if (threadIdx.x == 0) out[blockIdx.x] = ACC ? out[blockIdx.x] + s : s;
}

// out[b] = (sum of list b, fixed order) + add[b]: what k_finish2<0> followed by out += add gives, in one launch
__global__ __launch_bounds__(256) void k_finish2_plus(const double* __restrict__ p0, const double* __restrict__ p1, int n,
                                                       const double* __restrict__ add, double* __restrict__ out)
{
    __shared__ double red[16];
    const double s = fpr_sum_partials_256(blockIdx.x ? p1 : p0, n, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s + add[blockIdx.x];
}

int fprx_finish_sum2_plus(fpr_ctx* ctx, const double* p0, const double* p1, int nparts, const double* add2_dev, double* out2_dev,
                          int stream_sel)
{
    k_finish2_plus<<<2, 256, 0, ctx->stream[stream_sel]>>>(p0, p1, nparts, add2_dev, out2_dev);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// three partial lists of equal length -> out[0..2] in one launch (k_diff3_march3: the sums of its three iterations)
__global__ __launch_bounds__(256) void k_finish3(const double* __restrict__ p0, const double* __restrict__ p1, const double* __restrict__ p2,
                                                  int n, double* __restrict__ out, const double* __restrict__ add)
{
    __shared__ double red[16];
    const double s = fpr_sum_partials_256(blockIdx.x == 0 ? p0 : (blockIdx.x == 1 ? p1 : p2), n, red);
    if (threadIdx.x == 0) out[blockIdx.x] = add ? s + add[blockIdx.x] : s;
}

int fprx_finish_sum3(fpr_ctx* ctx, const double* p0, const double* p1, const double* p2, int nparts, double* out3_dev, int stream_sel,
                     const double* add3_dev)
{
    k_finish3<<<3, 256, 0, ctx->stream[stream_sel]>>>(p0, p1, p2, nparts, out3_dev, add3_dev);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

int fprx_finish_sum2(fpr_ctx* ctx, const double* p0, const double* p1, int nparts, double* out2_dev, bool accumulate,
                     int stream_sel)
{
    if (accumulate) k_finish2<1><<<2, 256, 0, ctx->stream[stream_sel]>>>(p0, p1, nparts, out2_dev);
    else k_finish2<0><<<2, 256, 0, ctx->stream[stream_sel]>>>(p0, p1, nparts, out2_dev);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

int fprx_sumsq_scaled_dev(fpr_ctx* ctx, const double* x, size_t n, double scale, double* out_dev, int stream_sel)
{
    const int g = flat_grid(n);
    double* part = stream_sel == 1 ? ctx->partials2 : ctx->partials;
    k_reduce<0><<<g, 256, 0, ctx->stream[stream_sel]>>>(x, nullptr, n, scale, part);
    FPR_CHECK_LAUNCH(ctx);
    return fprx_finish_sum(ctx, part, g, out_dev, false, stream_sel);
}

int fprx_dot_dev(fpr_ctx* ctx, const double* x, const double* y, size_t n, double* out_dev)
{
    const int g = flat_grid(n);
    k_reduce<1><<<g, 256, 0, ctx->stream[0]>>>(x, y, n, 1.0, ctx->partials);
    FPR_CHECK_LAUNCH(ctx);
    return fprx_finish_sum(ctx, ctx->partials, g, out_dev, false, 0);
}

int fprx_dot2_dev(fpr_ctx* ctx, const double* x, const double* y, size_t n, double* out_dev)
{
    const int g = flat_grid(n);
    k_dot2<<<g, 256, 0, ctx->stream[0]>>>(x, y, n, ctx->partials);
    FPR_CHECK_LAUNCH(ctx);
    k_finish_s2<<<1, 256, 0, ctx->stream[0]>>>(ctx->partials, g, out_dev);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

static int read_scalar(fpr_ctx* ctx, const double* dev, double* out_host)
{
    FPR_HIP(ctx, hipMemcpyAsync(ctx->host_scalars, dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream[0]));
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[0]));
    *out_host = ctx->host_scalars[0];
    return FPR_OK;
}

extern "C" int fpr_sumsq_scaled_dev(fpr_ctx* ctx, const double* x, size_t n, double scale, double* out_dev)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, x && out_dev, "null pointer");
    return fprx_sumsq_scaled_dev(ctx, x, n, scale, out_dev, 0);
}

extern "C" int fpr_sumsq_scaled(fpr_ctx* ctx, const double* x, size_t n, double scale, double* out_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, x && out_host, "null pointer");
    int rc = fprx_sumsq_scaled_dev(ctx, x, n, scale, ctx->scalars, 0);
    if (rc) return rc;
    return read_scalar(ctx, ctx->scalars, out_host);
}

extern "C" int fpr_dot(fpr_ctx* ctx, const double* x, const double* y, size_t n, double* out_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, x && y && out_host, "null pointer");
    int rc = fprx_dot_dev(ctx, x, y, n, ctx->scalars);
    if (rc) return rc;
    return read_scalar(ctx, ctx->scalars, out_host);
}

extern "C" int fpr_absmax(fpr_ctx* ctx, const double* x, size_t n, double* out_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, x && out_host, "null pointer");
    const int g = flat_grid(n);
    k_reduce<2><<<g, 256, 0, ctx->stream[0]>>>(x, nullptr, n, 1.0, ctx->partials);
    FPR_CHECK_LAUNCH(ctx);
    k_finish<2><<<1, 256, 0, ctx->stream[0]>>>(ctx->partials, g, ctx->scalars);
    FPR_CHECK_LAUNCH(ctx);
    return read_scalar(ctx, ctx->scalars, out_host);
}

extern "C" int fpr_copy(fpr_ctx* ctx, double* dst, const double* src, size_t n)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, dst && src, "null pointer");
    if (n == 0) return FPR_OK;
    k_copy<<<flat_grid((n + 1) / 2), 256, 0, ctx->stream[0]>>>(dst, src, n);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_fill(fpr_ctx* ctx, double* dst, double value, size_t n)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, dst, "null pointer");
    if (n == 0) return FPR_OK;
    k_fill<<<flat_grid(n), 256, 0, ctx->stream[0]>>>(dst, value, n);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

__global__ __launch_bounds__(256) void k_addto(double* __restrict__ dst, const double* __restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = dst[i] + src[i];
}

// fill / dst += src on a chosen stream (0 compute, 1 comm, 2 core): the norm accumulators of a decomposed run's fused
// pair live on the comm and core streams, and nothing of a pair should have to pass through the compute stream
extern "C" int fpr_fill_on(fpr_ctx* ctx, double* dst, double value, size_t n, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, dst && stream_sel >= 0 && stream_sel <= 2, "null pointer / stream_sel");
    if (n == 0) return FPR_OK;
    k_fill<<<flat_grid(n), 256, 0, ctx->stream[stream_sel]>>>(dst, value, n);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_add_on(fpr_ctx* ctx, double* dst, const double* src, size_t n, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, dst && src && stream_sel >= 0 && stream_sel <= 2, "null pointer / stream_sel");
    if (n == 0) return FPR_OK;
    k_addto<<<flat_grid(n), 256, 0, ctx->stream[stream_sel]>>>(dst, src, n);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_axmy2d(fpr_ctx* ctx, double* u, const double* corr, size_t n)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, u && corr, "null pointer");
    if (n == 0) return FPR_OK;
    k_axmy<<<flat_grid(n), 256, 0, ctx->stream[0]>>>(u, corr, n);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}
