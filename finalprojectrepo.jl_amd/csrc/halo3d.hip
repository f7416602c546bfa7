// halo3d.hip -- pack / unpack of halo planes for the domain-decomposed 3D diffusion path.
// Role of ImplicitGlobalGrid's update_halo! send/receive buffers (called at
// scripts-part1/part1_kernel_programming.jl:182,187).  The exchange itself is done by the host over
// RCCL (torch.distributed isend/irecv); z-faces are contiguous in memory and need no packing.
#include "fpr_internal.hpp"

// face = 2*dim + side.  plane index: send plane = 1 (low) or n-2 (high); halo plane = 0 or n-1.
template <bool PACK>
__global__ __launch_bounds__(256) void k_halo_plane(double* __restrict__ A, double* __restrict__ buf, int nx, int ny, int nz,
                                                     int dim, int plane)
{
    // (a, b) index the plane: dim 0 -> (j,k), dim 1 -> (i,k), dim 2 -> (i,j); `a` is the fast index
    const int na = dim == 0 ? ny : nx;
    const int nb = dim == 2 ? ny : nz;
    const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
    if (a >= na || b >= nb) return;
    size_t id;
    if (dim == 0) id = (size_t)plane + (size_t)nx * ((size_t)a + (size_t)ny * b);
    else if (dim == 1) id = (size_t)a + (size_t)nx * ((size_t)plane + (size_t)ny * b);
    else id = (size_t)a + (size_t)nx * ((size_t)b + (size_t)ny * plane);
    const size_t bi = (size_t)a + (size_t)na * b;
    if constexpr (PACK) buf[bi] = A[id];
    else A[id] = buf[bi];
}

static int halo_args(fpr_ctx* ctx, int nx, int ny, int nz, int face, int stream_sel)
{
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3 && nz >= 3, "grid must be at least 3^3");
    FPR_REQUIRE(ctx, face >= 0 && face < 6, "face must be 0..5");
    FPR_REQUIRE(ctx, stream_sel == 0 || stream_sel == 1, "stream_sel");
    return FPR_OK;
}

extern "C" int fpr_halo_pack3d(fpr_ctx* ctx, const double* A, int nx, int ny, int nz, int face, double* buf, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, A && buf, "null pointer");
    if (int rc = halo_args(ctx, nx, ny, nz, face, stream_sel)) return rc;
    const int dim = face >> 1, side = face & 1;
    const int n[3] = {nx, ny, nz};
    const int plane = side ? n[dim] - 2 : 1;
    const int na = dim == 0 ? ny : nx, nb = dim == 2 ? ny : nz;
    k_halo_plane<true><<<dim3((na + 63) / 64, (nb + 3) / 4), dim3(64, 4), 0, ctx->stream[stream_sel]>>>(
        const_cast<double*>(A), buf, nx, ny, nz, dim, plane);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}

extern "C" int fpr_halo_unpack3d(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face, const double* buf, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, A && buf, "null pointer");
    if (int rc = halo_args(ctx, nx, ny, nz, face, stream_sel)) return rc;
    const int dim = face >> 1, side = face & 1;
    const int n[3] = {nx, ny, nz};
    const int plane = side ? n[dim] - 1 : 0;
    const int na = dim == 0 ? ny : nx, nb = dim == 2 ? ny : nz;
    k_halo_plane<false><<<dim3((na + 63) / 64, (nb + 3) / 4), dim3(64, 4), 0, ctx->stream[stream_sel]>>>(
        A, const_cast<double*>(buf), nx, ny, nz, dim, plane);
    FPR_CHECK_LAUNCH(ctx);
    return FPR_OK;
}
