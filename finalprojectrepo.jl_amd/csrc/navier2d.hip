// navier2d.hip -- NEXT row 8f-1: pointwise kernels of the stream-function/vorticity Navier-Stokes step
// that calls the V-cycle (reference scripts-part2/part2.jl:90-137).  One thread per interior point.
#include "fpr_internal.hpp"
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#define NS_IDX                                                                      \
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;  \
    if (i < 1 || j < 1 || i >= nx - 1 || j >= ny - 1) return;                       \
    const size_t id = (size_t)i + (size_t)nx * j;

__global__ __launch_bounds__(256) void k_velocity(const double* __restrict__ S, double hx, double hy, double* __restrict__ vx,
                                                   double* __restrict__ vy, int nx, int ny)
{
    NS_IDX
    vx[id] = (S[id + nx] - S[id - nx]) / (2 * hy);   // part2.jl:92
    vy[id] = -(S[id + 1] - S[id - 1]) / (2 * hx);    // part2.jl:93
}

__global__ __launch_bounds__(256) void k_Ra_dTdx(double Ra, double hx, const double* __restrict__ T, double* __restrict__ out,
                                                  int nx, int ny)
{
    NS_IDX
    out[id] = Ra * (T[id + 1] - T[id - 1]) / (2 * hx);  // part2.jl:101
}

__global__ __launch_bounds__(256) void k_diffusion2d(const double* __restrict__ T, double hx2, double hy2, double k,
                                                      double* __restrict__ dT2, int nx, int ny)
{
    NS_IDX
    const double t = T[id];
    dT2[id] = k * (((T[id + 1] - 2 * t) + T[id - 1]) / hx2 + ((T[id + nx] - 2 * t) + T[id - nx]) / hy2);  // part2.jl:109-110
}

template <int DIM>
__global__ __launch_bounds__(256) void k_advection(const double* __restrict__ T, double h, const double* __restrict__ v,
                                                    double* __restrict__ out, int nx, int ny)
{
    NS_IDX
    const size_t st = DIM == 0 ? 1 : (size_t)nx;
    const double vv = v[id];
    if (vv > 0) out[id] = vv * (T[id] - T[id - st]) / h;  // part2.jl:119 / :131
    else out[id] = vv * (T[id + st] - T[id]) / h;         // part2.jl:121 / :133
}

static inline dim3 g2(int nx, int ny) { return dim3((nx + 63) / 64, (ny + 3) / 4); }
#define NS_CHECK(...)                                                   \
    if (!ctx) return FPR_ERR_INVALID;                                   \
    FPR_REQUIRE(ctx, (__VA_ARGS__), "null pointer");                    \
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3, "grid must be at least 3x3");

extern "C" int fpr_compute_velocity2d(fpr_ctx* ctx, const double* S, double hx, double hy, double* vx, double* vy, int nx, int ny)
{
    NS_CHECK(S && vx && vy)
    k_velocity<<<g2(nx, ny), dim3(64, 4), 0, ctx->stream[0]>>>(S, hx, hy, vx, vy, nx, ny);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_compute_Ra_dTdx2d(fpr_ctx* ctx, double Ra, double hx, const double* T, double* out, int nx, int ny)
{
    NS_CHECK(T && out)
    k_Ra_dTdx<<<g2(nx, ny), dim3(64, 4), 0, ctx->stream[0]>>>(Ra, hx, T, out, nx, ny);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_compute_diffusion2d(fpr_ctx* ctx, const double* T, double hx, double hy, double k, double* dT2, int nx, int ny)
{
    NS_CHECK(T && dT2)
    k_diffusion2d<<<g2(nx, ny), dim3(64, 4), 0, ctx->stream[0]>>>(T, hx * hx, hy * hy, k, dT2, nx, ny);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_compute_advection2d_x(fpr_ctx* ctx, const double* T, double hx, const double* vx, double* dTx, int nx, int ny)
{
    NS_CHECK(T && vx && dTx)
    k_advection<0><<<g2(nx, ny), dim3(64, 4), 0, ctx->stream[0]>>>(T, hx, vx, dTx, nx, ny);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_compute_advection2d_y(fpr_ctx* ctx, const double* T, double hy, const double* vy, double* dTy, int nx, int ny)
{
    NS_CHECK(T && vy && dTy)
    k_advection<1><<<g2(nx, ny), dim3(64, 4), 0, ctx->stream[0]>>>(T, hy, vy, dTy, nx, ny);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The step around the V-cycle in TWO passes.  The reference runs, per time step, seven pointwise kernels,
// three whole-array maxima and four whole-array broadcasts (part2.jl:190-230): about 25 sweeps over (nx, ny) arrays.
//   pass 1  fpr_ns_velocity_max2d : velocity from the stream function (:190) folded into the three maxima compute_dt
//                                   needs (:193-196, :76-87); vx / vy are written only if the caller wants them
//   (host: dt, part2.jl:76-87)
//   pass 2  fpr_ns_rhs2d          : Ra dT/dx, both diffusion terms, the four upwind advection terms (velocity
//                                   recomputed from S) and the right-hand sides of the two semi-implicit solves
//                                   (:220, :225) or the explicit Euler update (:229-230), for every point of the arrays
// Every expression keeps the reference's operand order (and the library is built without FMA contraction), so the
// results are bit-identical to the kernel-by-kernel path; maxima are order-independent.
constexpr int NS_ROWS_PER_BLOCK = 32;
__global__ __launch_bounds__(256) void k_ns_velocity_max(const double* __restrict__ S, double hx, double hy,
                                                          double* __restrict__ vx_out, double* __restrict__ vy_out, int nx,
                                                          int ny, double* __restrict__ partials, int nblk)
{
    __shared__ double red[16];
    const int i = blockIdx.x * 64 + threadIdx.x;
    double mv = 0.0, mx = 0.0, my = 0.0;
    // a block covers NS_ROWS_PER_BLOCK rows (4 per trip): maxima are order-independent, and the finishing launch has an
    // eighth of the partials to read (2049^2: 17.9 -> ~5 us)
#pragma unroll
    for (int q = 0; q < NS_ROWS_PER_BLOCK / 4; ++q) {
        const int j = blockIdx.y * NS_ROWS_PER_BLOCK + q * 4 + threadIdx.y;
        if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
            const size_t id = (size_t)i + (size_t)nx * j;
            const double vx = (S[id + nx] - S[id - nx]) / (2 * hy);   // part2.jl:92
            const double vy = -(S[id + 1] - S[id - 1]) / (2 * hx);    // part2.jl:93
            if (vx_out) vx_out[id] = vx;
            if (vy_out) vy_out[id] = vy;
            mv = fmax(mv, sqrt(vx * vx + vy * vy));                   // part2.jl:193
            mx = fmax(mx, fabs(vx));
            my = fmax(my, fabs(vy));
        }
    }
    const int b = blockIdx.x + gridDim.x * blockIdx.y;
    mv = fpr_block_max<256>(mv, red);
    __syncthreads();
    mx = fpr_block_max<256>(mx, red);
    __syncthreads();
    my = fpr_block_max<256>(my, red);
    if (threadIdx.x == 0 && threadIdx.y == 0) { partials[b] = mv; partials[nblk + b] = mx; partials[2 * nblk + b] = my; }
}

// the three maxima and then a sequence number into pinned host memory: the host polls the number (a stream synchronisation
// wakes the host 15-25 us after the kernel has ended; tools/exp_ns_timeline.sh)
__global__ __launch_bounds__(256) void k_ns_max3_finish(const double* __restrict__ partials, int nblk, double* out_host, int seq)
{
    __shared__ double red[16];
    double m3[3];
    for (int q = 0; q < 3; ++q) {
        double m = 0.0;
        const double* p = partials + (size_t)q * nblk;
        for (int i = threadIdx.x; i < nblk; i += 256) m = fmax(m, p[i]);
        m3[q] = fpr_block_max<256>(m, red);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_host[0] = m3[0]; out_host[1] = m3[1]; out_host[2] = m3[2];
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<int*>(out_host + 3), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool IMPLICIT>
__global__ __launch_bounds__(256) void k_ns_rhs(const double* __restrict__ T, const double* __restrict__ W,
                                                 const double* __restrict__ S, double hx, double hy, int nx, int ny, double Ra,
                                                 double Pr, double k, double beta, double dt, double cT, double cW,
                                                 bool diffuse, double* __restrict__ T_out, double* __restrict__ W_out)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= nx || j >= ny) return;
    const size_t id = (size_t)i + (size_t)nx * j;
    const double t = T[id], w = W[id];
    // the reference's term arrays are @zeros and written at interior points only (part2.jl:90-137)
    double Rd = 0.0, dT2 = 0.0, dW2 = 0.0, dTx = 0.0, dTy = 0.0, dWx = 0.0, dWy = 0.0;
    if (i >= 1 && j >= 1 && i < nx - 1 && j < ny - 1) {
        const double tE = T[id + 1], tW = T[id - 1], tN = T[id + nx], tS = T[id - nx];
        const double wE = W[id + 1], wW = W[id - 1], wN = W[id + nx], wS = W[id - nx];
        const double hx2 = hx * hx, hy2 = hy * hy;
        const double vx = (S[id + nx] - S[id - nx]) / (2 * hy);   // :92
        const double vy = -(S[id + 1] - S[id - 1]) / (2 * hx);    // :93
        Rd = Ra * (tE - tW) / (2 * hx);                            // :101
        if (diffuse) {                                             // :205-208 (opt.beta not approximately 1)
            dT2 = k * (((tE - 2 * t) + tW) / hx2 + ((tN - 2 * t) + tS) / hy2);    // :109-110
            dW2 = Pr * (((wE - 2 * w) + wW) / hx2 + ((wN - 2 * w) + wS) / hy2);
        }
        dTx = vx > 0 ? vx * (t - tW) / hx : vx * (tE - t) / hx;    // :118-122
        dTy = vy > 0 ? vy * (t - tS) / hy : vy * (tN - t) / hy;    // :130-134
        dWx = vx > 0 ? vx * (w - wW) / hx : vx * (wE - w) / hx;
        dWy = vy > 0 ? vy * (w - wS) / hy : vy * (wN - w) / hy;
    }
    if constexpr (IMPLICIT) {
        T_out[id] = -cT * (t + dt * (((1.0 - beta) * dT2 - dTx) - dTy));                    // :220
        W_out[id] = -cW * (w + dt * ((((1.0 - beta) * dW2 - dWx) - dWy) - Pr * Rd));        // :225
    } else {
        T_out[id] = t + dt * ((dT2 - dTx) - dTy);                                           // :229
        W_out[id] = w + dt * (((dW2 - dWx) - dWy) - Pr * Rd);                               // :230
    }
}

// pass 1.  vmax_host[3] = maximum(v), maximum(abs.(vx)), maximum(abs.(vy)) (part2.jl:77,82); vx / vy may be NULL.
// Synchronises the compute stream (the reference's maximum() does, too).
extern "C" int fpr_ns_velocity_max2d(fpr_ctx* ctx, const double* S, double hx, double hy, double* vx, double* vy, int nx, int ny,
                                     double* vmax_host)
{
    NS_CHECK(S && vmax_host)
    const dim3 g((nx + 63) / 64, (ny + NS_ROWS_PER_BLOCK - 1) / NS_ROWS_PER_BLOCK);
    const int nblk = (int)(g.x * g.y);
    FPR_REQUIRE(ctx, 3L * nblk <= FPR_MAX_PARTIALS, "grid too large for the partial buffer");
    k_ns_velocity_max<<<g, dim3(64, 4), 0, ctx->stream[0]>>>(S, hx, hy, vx, vy, nx, ny, ctx->partials, nblk);
    FPR_CHECK_LAUNCH(ctx);
    // (the three maxima go straight to pinned host memory, a sequence number behind them: no copy launch, no stream synchronisation)
    const int seq = ++ctx->ns_seq;
    int* flag = reinterpret_cast<int*>(ctx->host_scalars + 19);
    __atomic_store_n(flag, 0, __ATOMIC_RELEASE);   // before the launch that will report into it
    k_ns_max3_finish<<<1, 256, 0, ctx->stream[0]>>>(ctx->partials, nblk, ctx->host_scalars + 16, seq);
    FPR_CHECK_LAUNCH(ctx);
    unsigned long spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xffff) == 0) {
            const hipError_t q = hipStreamQuery(ctx->stream[0]);
            if (q == hipSuccess) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
                return fpr_fail(ctx, FPR_ERR_HIP, "the velocity maxima never arrived (stream idle)");
            }
            if (q != hipErrorNotReady) return fpr_fail(ctx, FPR_ERR_HIP, "HIP error while waiting for the velocity maxima: %s", hipGetErrorString(q));
        }
    }
    for (int q = 0; q < 3; ++q) vmax_host[q] = ctx->host_scalars[16 + q];
    return FPR_OK;
}

// pass 2.  beta > 0: T_out / W_out = right-hand sides of the semi-implicit solves with c = 1/(beta dt) and c/Pr
// (part2.jl:219-225); beta == 0: the explicit Euler update (:229-230).  T must already carry its boundary conditions
// (apply_boundary_conditions!, :199).  T_out / W_out must not alias T, W, S.
extern "C" int fpr_ns_rhs2d(fpr_ctx* ctx, const double* T, const double* W, const double* S, double hx, double hy, int nx, int ny,
                            double Ra, double Pr, double k, double beta, double dt, double* T_out, double* W_out)
{
    NS_CHECK(T && W && S && T_out && W_out)
    FPR_REQUIRE(ctx, T_out != T && T_out != W && T_out != S && W_out != T && W_out != W && W_out != S && T_out != W_out,
                "outputs must be buffers of their own");
    // `opt.beta ≉ 1.0` (:205): Julia's isapprox, rtol = sqrt(eps)
    const bool diffuse = !(fabs(beta - 1.0) <= 1.4901161193847656e-08 * fmax(fabs(beta), 1.0));
    if (beta > 0.0) {
        const double c = 1.0 / (beta * dt);   // :219
        k_ns_rhs<true><<<g2(nx, ny), dim3(64, 4), 0, ctx->stream[0]>>>(T, W, S, hx, hy, nx, ny, Ra, Pr, k, beta, dt, c, c / Pr,
                                                                          diffuse, T_out, W_out);
    } else {
        k_ns_rhs<false><<<g2(nx, ny), dim3(64, 4), 0, ctx->stream[0]>>>(T, W, S, hx, hy, nx, ny, Ra, Pr, k, beta, dt, 0.0, 0.0,
                                                                           diffuse, T_out, W_out);
    }
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

// ---- time steps of navier_stokes_2D inside the library (part2.jl:182-262, semi-implicit / implicit: beta > 0) -----------------
// The loop body of the reference's driver: S solve (:187), velocities and their maxima (:190-193), compute_dt (:76-87, :196),
// boundary conditions of T (:199), all pointwise terms and both right-hand sides (:202-220,225), T solve (:221) and W solve
// (:226).  Between those pieces the host only computes dt; from Python every piece is a ctypes call, the two independent solves
// meet through a thread pool, and the loop itself costs 30-60 us per step -- 80-130 us of a 1.45 ms step at 2049^2 during which the
// device waits for the host (tools/exp_ns_timeline.sh).  Here the W solve (the longer one: it sets the step's length) runs on
// the calling thread on `ctx2` (own streams, own arena), the T solve beside it on `ctx` on a worker thread the context keeps;
// the two contexts' compute streams are ordered by events.  Same launches with the same arguments as the piecewise path:
// same T, W, S, dt bit for bit.
namespace {
struct FprWorker {   // one job at a time
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, quit = false;
    std::thread th;
    FprWorker() : th([this] { loop(); }) {}
    void loop()
    {
        std::unique_lock<std::mutex> l(m);
        for (;;) {
            cv.wait(l, [&] { return has_job || quit; });
            if (quit) return;
            std::function<void()> j = std::move(job);
            has_job = false;
            l.unlock();
            j();
            l.lock();
            done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j)
    {
        { std::lock_guard<std::mutex> l(m); job = std::move(j); has_job = true; done = false; }
        cv.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return done; }); }
    ~FprWorker()
    {
        { std::lock_guard<std::mutex> l(m); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};
}   // namespace

void fprx_ns_worker_free(fpr_ctx* ctx)
{
    if (ctx->ns_worker) { delete static_cast<FprWorker*>(ctx->ns_worker); ctx->ns_worker = nullptr; }
}

static int ns_order(fpr_ctx* ctx, fpr_ctx* from, fpr_ctx* to)   // `to`'s compute stream behind what `from`'s holds now
{
    FPR_HIP(ctx, hipEventRecord(from->ns_ev, from->stream[0]));
    FPR_HIP(ctx, hipStreamWaitEvent(to->stream[0], from->ns_ev, 0));
    return FPR_OK;
}

// The loop `while sim_time < ttot` (:182) for at most max_steps steps, software-pipelined: the S solve of step n+1 (:187) needs the W
// of step n and nothing of its T, so it runs BEHIND the W solve on the second context, beside the T solve of step n (the longest of
// the three at config 5: 6-7 cycles against 3 + 4) -- the critical path of a step falls from S + max(T, W) to max(T, W + S).  The
// S solve of a step that does not follow inside this call (last step, or sim_time reaches ttot) is left to the next call, so the
// arrays a call returns are the reference's at the same point of its loop.  Same solves on the same inputs: same results.
// info (nullable, 6 ints): cycles of the S, T, W solves of the LAST step taken, then their converged flags.
static int ns_run(fpr_ctx* ctx, fpr_ctx* ctx2, double* S, double* T, double* W, double* T_rhs, double* W_rhs, int nx, int ny, double Ra,
                  double Pr, double k, double beta, double a_adv, double dt_dif, double tol, int niters, int coarse_solve_size,
                  int coarse_solver, double ttot, int max_steps, double* sim_time_inout, int* steps_host, double* dt_host,
                  int* unconverged_host, int* info)
{
    const double h = 1.0 / (ny - 1.0);   // :163
    const bool pipeline = fpr_opt(ctx, "ns_pipeline", 1) != 0;
    FprWorker* wk = static_cast<FprWorker*>(ctx->ns_worker);
    const int dev = ctx->device;
    double t = *sim_time_inout;
    int steps = 0, bad = 0;
    int ncyc[3] = {0, 0, 0}, conv[3] = {1, 1, 1};
    double rmsS = 0.0, frmsS = 0.0, rmsT = 0.0, frmsT = 0.0, rmsW = 0.0, frmsW = 0.0;
    bool have_S = false;   // the S of the coming step has been solved already (beside the previous step's T solve)
    *steps_host = 0;
    auto solveS = [&]() -> int {   // :187, on the second context (its stream is ordered behind W's producer by the caller)
        return fpr_mgsolve2d(ctx2, S, W, h, 0.0, tol, niters, 0, coarse_solve_size, coarse_solver, nx, ny, &rmsS, &ncyc[0], nullptr, &frmsS, &conv[0]);
    };
    while (t < ttot && steps < max_steps) {
        if (!have_S) {
            if (int rc = ns_order(ctx, ctx, ctx2)) return rc;       // W (and S) as the caller / the previous step left them
            if (int rc = solveS()) return fpr_fail(ctx, rc, "S solve on the second context: %s", fpr_last_error(ctx2));
            if (int rc = ns_order(ctx, ctx2, ctx)) return rc;
            bad += !conv[0];
        }
        double vm[3];
        if (int rc = fpr_ns_velocity_max2d(ctx, S, h, h, nullptr, nullptr, nx, ny, vm)) return rc;   // :190-193
        double dt;
        if (vm[0] == 0.0) dt = dt_dif;   // compute_dt, :76-87
        else {
            const double dt_adv = a_adv * fmin(h / vm[1], h / vm[2]);
            dt = beta >= 0.5 ? dt_adv : fmin(dt_dif, dt_adv);
        }
        if (int rc = fpr_bc2d(ctx, T, nx, ny)) return rc;                                                             // :199
        if (int rc = fpr_ns_rhs2d(ctx, T, W, S, h, h, nx, ny, Ra, Pr, k, beta, dt, T_rhs, W_rhs)) return rc;         // :202-220,225
        const double c = 1.0 / (beta * dt);   // :219
        // does another step follow inside this call?  (the loop's own test, :182, with the time this step will have reached)
        const bool more = pipeline && steps + 1 < max_steps && t + dt < ttot;
        // W, W_rhs, S (and what they were computed from) are complete for the other context's stream
        if (int rc = ns_order(ctx, ctx, ctx2)) return rc;
        int rcT = FPR_OK, rcW = FPR_OK, rcS = FPR_OK;
        int cS = conv[0];
        auto solveT = [&] {   // :221
            rcT = fpr_mgsolve2d(ctx, T, T_rhs, h, c, tol, niters, 1, coarse_solve_size, coarse_solver, nx, ny, &rmsT, &ncyc[1], nullptr, &frmsT, &conv[1]);
        };
        auto solveWS = [&] {   // :226, and the next step's :187 behind it
            rcW = fpr_mgsolve2d(ctx2, W, W_rhs, h, c / Pr, tol, niters, 0, coarse_solve_size, coarse_solver, nx, ny, &rmsW, &ncyc[2], nullptr, &frmsW, &conv[2]);
            if (rcW == FPR_OK && more) rcS = solveS();
        };
        // the side that took more cycles in the previous step runs on the calling thread (the other one starts a few microseconds
        // later, when the worker has woken up): it sets the step's length
        const bool t_here = ctx->ns_t_cycles >= ctx->ns_w_cycles + (more ? ctx->ns_s_cycles : 0);
        if (t_here) {
            wk->submit([&] { if (hipSetDevice(dev) != hipSuccess) { rcW = FPR_ERR_HIP; return; } solveWS(); });   // (a thread starts on device 0)
            solveT();
        } else {
            wk->submit([&] { if (hipSetDevice(dev) != hipSuccess) { rcT = FPR_ERR_HIP; return; } solveT(); });
            solveWS();
        }
        wk->wait();
        ctx->ns_t_cycles = ncyc[1]; ctx->ns_w_cycles = ncyc[2];
        if (more || !have_S) ctx->ns_s_cycles = ncyc[0];
        // the next step's kernels (this context's compute stream) read W and S
        if (int rc = ns_order(ctx, ctx2, ctx)) return rc;
        if (rcT) return rcT == FPR_ERR_HIP && !ctx->err[0] ? fpr_fail(ctx, rcT, "hipSetDevice in the worker thread") : rcT;
        if (rcW) return fpr_fail(ctx, rcW, "W solve on the second context: %s", fpr_last_error(ctx2));
        if (rcS) return fpr_fail(ctx, rcS, "S solve on the second context: %s", fpr_last_error(ctx2));
        if (info) { info[0] = more ? 0 : ncyc[0]; info[1] = ncyc[1]; info[2] = ncyc[2]; info[3] = more ? 1 : cS; info[4] = conv[1]; info[5] = conv[2]; }
        *dt_host = dt;
        t += dt;   // :249
        ++steps;
        *sim_time_inout = t; *steps_host = steps;
        bad += !conv[1] + !conv[2] + (more ? !conv[0] : 0);
        have_S = more;
    }
    *sim_time_inout = t; *steps_host = steps;
    if (unconverged_host) *unconverged_host = bad;
    return FPR_OK;
}

static int ns_prepare(fpr_ctx* ctx, fpr_ctx* ctx2, double beta)
{
    FPR_REQUIRE(ctx, ctx2 && ctx2 != ctx && ctx2->device == ctx->device, "a second context on the same device is needed for the W solve");
    FPR_REQUIRE(ctx, beta > 0.0, "explicit steps (beta == 0) have no solves to put side by side: fpr_ns_rhs2d is the whole step");
    if (!ctx->ns_ev) FPR_HIP(ctx, hipEventCreateWithFlags(&ctx->ns_ev, hipEventDisableTiming));
    if (!ctx2->ns_ev) FPR_HIP(ctx, hipEventCreateWithFlags(&ctx2->ns_ev, hipEventDisableTiming));
    if (!ctx->ns_worker) ctx->ns_worker = new FprWorker();
    return FPR_OK;
}

// info_host (nullable, 6 ints): cycles of the S, T, W solves, then their `converged` flags (0 where the reference would @warn, :78-80).
extern "C" int fpr_ns_step2d(fpr_ctx* ctx, fpr_ctx* ctx2, double* S, double* T, double* W, double* T_rhs, double* W_rhs, int nx, int ny,
                             double Ra, double Pr, double k, double beta, double a_adv, double dt_dif, double tol, int niters,
                             int coarse_solve_size, int coarse_solver, double* dt_host, int* info_host)
{
    NS_CHECK(S && T && W && T_rhs && W_rhs && dt_host)
    if (int rc = ns_prepare(ctx, ctx2, beta)) return rc;
    double t = 0.0;
    int steps = 0;
    return ns_run(ctx, ctx2, S, T, W, T_rhs, W_rhs, nx, ny, Ra, Pr, k, beta, a_adv, dt_dif, tol, niters, coarse_solve_size, coarse_solver,
                  1.0, 1, &t, &steps, dt_host, nullptr, info_host);
}

// `while sim_time < ttot` (:182) for at most max_steps steps: *sim_time_inout advances by every step's dt (:249), *steps_host = steps
// taken, *dt_host = the last dt, *unconverged_host (nullable) = solves that did not converge (the reference's @warn, :78-80).
extern "C" int fpr_ns_run2d(fpr_ctx* ctx, fpr_ctx* ctx2, double* S, double* T, double* W, double* T_rhs, double* W_rhs, int nx, int ny,
                            double Ra, double Pr, double k, double beta, double a_adv, double dt_dif, double tol, int niters,
                            int coarse_solve_size, int coarse_solver, double ttot, int max_steps, double* sim_time_inout,
                            int* steps_host, double* dt_host, int* unconverged_host)
{
    NS_CHECK(S && T && W && T_rhs && W_rhs && dt_host && sim_time_inout && steps_host)
    if (int rc = ns_prepare(ctx, ctx2, beta)) return rc;
    return ns_run(ctx, ctx2, S, T, W, T_rhs, W_rhs, nx, ny, Ra, Pr, k, beta, a_adv, dt_dif, tol, niters, coarse_solve_size, coarse_solver,
                  ttot, max_steps, sim_time_inout, steps_host, dt_host, unconverged_host, nullptr);
}
