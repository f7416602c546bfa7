// navier2d.hip -- NEXT row 8f-1: pointwise kernels of the stream-function/vorticity Navier-Stokes step
// that calls the V-cycle (reference scripts-part2/part2.jl:90-137).  One thread per interior point.
#include "fpr_internal.hpp"

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
