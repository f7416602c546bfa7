// comm.hip -- the process/device boundary of the domain-decomposed 3D diffusion path: RCCL communicator,
// implicit global grid, halo exchange and the all-reduce of the convergence norm.
//
// Replaces, for one process per GPU on one node (xGMI):
//   init_global_grid / finalize_global_grid / nx_g.. (ImplicitGlobalGrid, scripts-part1/part1_kernel_programming.jl:100-101,117,225)
//   update_halo!(A)                                  (part1_kernel_programming.jl:182,187; part1_array_programming.jl:67)
//   MPI.Allreduce!(sq_residual, +, comm_cart)        (part1_utils.jl:38)
//   gather!(A, A_global)                             (part1_kernel_programming.jl:223)
// Transfers are ncclSend/ncclRecv inside ONE ncclGroup per exchange on the context's comm stream; z planes are
// contiguous in the column-major layout and travel in place, x / y planes through the pack kernels of halo3d.hip.
#include <rccl/rccl.h>

#include "fpr_internal.hpp"

static_assert(sizeof(ncclUniqueId) == FPR_UNIQUE_ID_BYTES, "FPR_UNIQUE_ID_BYTES must equal sizeof(ncclUniqueId)");

#define FPR_NCCL(ctx, call)                                                                              \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return fpr_fail((ctx), FPR_ERR_RCCL, "%s:%d %s -> %s", __FILE__, __LINE__, #call,           \
                            ncclGetErrorString(r_));                                                     \
    } while (0)

static inline ncclComm_t comm_of(fpr_ctx* ctx) { return (ncclComm_t)ctx->comm; }

// ---------------------------------------------------------------------------------------------------------------------
// The transport UNDER the exchange logic: five operations (group start / end, send, receive, all-reduce).  The product
// transport is RCCL.  fpr_comm_init_hosted puts a host-staged one in its place -- the same calls, issued by the same code
// (post_group's face order, the pack / unpack kernels, the strips, gather, the norm's all-reduce), but the bytes leave the
// device through host memory and travel by whatever the host process has (the tests: torch.distributed over gloo).  RCCL
// refuses two ranks on one device and a test box has one: this is how the send-high-first / receive-low-first matching and
// every buffer address of an exchange are EXECUTED between real processes before hardware with two cards appears.
// A send / receive outside a group runs at once; inside a group at the group's end: every send first (a send callback
// must not wait for the peer's receive), then the receives in posting order -- per peer the k-th send meets the k-th
// receive, RCCL's matching rule.
// ---------------------------------------------------------------------------------------------------------------------
struct FprHosted {
    fpr_hosted_send_fn send;
    fpr_hosted_recv_fn recv;
    fpr_hosted_allreduce_fn allreduce;
    void* user;
    struct Op { bool is_send; void* ptr; size_t bytes; int peer; hipStream_t s; };
    std::vector<Op> ops;
    int depth = 0;
    std::vector<char> stage;
};
static inline FprHosted* hosted_of(fpr_ctx* ctx) { return ctx->comm_hosted ? (FprHosted*)ctx->comm : nullptr; }

static int hosted_run(fpr_ctx* ctx, FprHosted* h, const FprHosted::Op& op)
{
    if (h->stage.size() < op.bytes) h->stage.resize(op.bytes);
    if (op.is_send) {
        FPR_HIP(ctx, hipMemcpyAsync(h->stage.data(), op.ptr, op.bytes, hipMemcpyDeviceToHost, op.s));
        FPR_HIP(ctx, hipStreamSynchronize(op.s));
        if (h->send(h->user, op.peer, h->stage.data(), op.bytes) != 0) return fpr_fail(ctx, FPR_ERR_RCCL, "hosted transport: send to rank %d failed", op.peer);
    } else {
        if (h->recv(h->user, op.peer, h->stage.data(), op.bytes) != 0) return fpr_fail(ctx, FPR_ERR_RCCL, "hosted transport: receive from rank %d failed", op.peer);
        FPR_HIP(ctx, hipMemcpyAsync(op.ptr, h->stage.data(), op.bytes, hipMemcpyHostToDevice, op.s));
        FPR_HIP(ctx, hipStreamSynchronize(op.s));     // the staging buffer is reused by the next operation
    }
    return FPR_OK;
}

static int x_group_start(fpr_ctx* ctx)
{
    if (FprHosted* h = hosted_of(ctx)) { ++h->depth; return FPR_OK; }
    FPR_NCCL(ctx, ncclGroupStart());
    return FPR_OK;
}
// returns the RCCL result through *res (a group must be closed whatever happened inside it); hosted: runs the group
static int x_group_end(fpr_ctx* ctx, ncclResult_t* res)
{
    *res = ncclSuccess;
    if (FprHosted* h = hosted_of(ctx)) {
        if (--h->depth > 0) return FPR_OK;
        std::vector<FprHosted::Op> ops;
        ops.swap(h->ops);
        for (int pass = 0; pass < 2; ++pass)
            for (const auto& op : ops)
                if (op.is_send == (pass == 0))
                    if (int rc = hosted_run(ctx, h, op)) return rc;
        return FPR_OK;
    }
    *res = ncclGroupEnd();
    return FPR_OK;
}
static ncclResult_t x_send(fpr_ctx* ctx, const double* src, size_t count, int peer, hipStream_t s, int* rc_out)
{
    *rc_out = FPR_OK;
    if (FprHosted* h = hosted_of(ctx)) {
        const FprHosted::Op op{true, (void*)src, count * sizeof(double), peer, s};
        if (h->depth > 0) h->ops.push_back(op);
        else *rc_out = hosted_run(ctx, h, op);
        return ncclSuccess;
    }
    return ncclSend(src, count, ncclDouble, peer, comm_of(ctx), s);
}
static ncclResult_t x_recv(fpr_ctx* ctx, double* dst, size_t count, int peer, hipStream_t s, int* rc_out)
{
    *rc_out = FPR_OK;
    if (FprHosted* h = hosted_of(ctx)) {
        const FprHosted::Op op{false, (void*)dst, count * sizeof(double), peer, s};
        if (h->depth > 0) h->ops.push_back(op);
        else *rc_out = hosted_run(ctx, h, op);
        return ncclSuccess;
    }
    return ncclRecv(dst, count, ncclDouble, peer, comm_of(ctx), s);
}
static int x_allreduce(fpr_ctx* ctx, double* x_dev, int count, hipStream_t s)
{
    if (FprHosted* h = hosted_of(ctx)) {
        std::vector<double> v((size_t)count);
        FPR_HIP(ctx, hipMemcpyAsync(v.data(), x_dev, v.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        FPR_HIP(ctx, hipStreamSynchronize(s));
        if (h->allreduce(h->user, v.data(), count) != 0) return fpr_fail(ctx, FPR_ERR_RCCL, "hosted transport: all-reduce failed");
        FPR_HIP(ctx, hipMemcpyAsync(x_dev, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, s));
        FPR_HIP(ctx, hipStreamSynchronize(s));
        return FPR_OK;
    }
    FPR_NCCL(ctx, ncclAllReduce(x_dev, x_dev, (size_t)count, ncclDouble, ncclSum, comm_of(ctx), s));
    return FPR_OK;
}

extern "C" int fpr_comm_init_hosted(fpr_ctx* ctx, int rank, int nranks, fpr_hosted_send_fn send, fpr_hosted_recv_fn recv,
                                    fpr_hosted_allreduce_fn allreduce, void* user)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, send && recv && allreduce && nranks >= 1 && rank >= 0 && rank < nranks, "rank / nranks / callbacks");
    FPR_REQUIRE(ctx, ctx->comm == nullptr, "communicator already initialised (call fpr_comm_finalize first)");
    FprHosted* h = new FprHosted();
    h->send = send; h->recv = recv; h->allreduce = allreduce; h->user = user;
    ctx->comm = (void*)h;
    ctx->comm_hosted = true;
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    return FPR_OK;
}

extern "C" int fpr_comm_get_unique_id(void* id_out)
{
    if (!id_out) return FPR_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return FPR_ERR_RCCL;
    memcpy(id_out, &id, sizeof(id));
    return FPR_OK;
}

extern "C" int fpr_comm_init(fpr_ctx* ctx, int rank, int nranks, const void* unique_id)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "rank / nranks / unique id");
    FPR_REQUIRE(ctx, ctx->comm == nullptr, "communicator already initialised (call fpr_comm_finalize first)");
    FPR_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    FPR_NCCL(ctx, ncclCommInitRank(&c, nranks, id, rank));
    ctx->comm = (void*)c;
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    return FPR_OK;
}

static void grid_free(fpr_ctx* ctx)
{
    for (int f = 0; f < 6; ++f) {
        if (ctx->grid.sendbuf[f]) hipFree(ctx->grid.sendbuf[f]);
        if (ctx->grid.recvbuf[f]) hipFree(ctx->grid.recvbuf[f]);
        ctx->grid.sendbuf[f] = ctx->grid.recvbuf[f] = nullptr;
    }
    if (ctx->grid.stage) hipFree(ctx->grid.stage);
    ctx->grid.stage = nullptr;
    ctx->grid.on = false;
}

extern "C" int fpr_comm_finalize(fpr_ctx* ctx)
{
    if (!ctx) return FPR_ERR_INVALID;
    grid_free(ctx);
    if (ctx->comm) {
        hipStreamSynchronize(ctx->stream[0]);
        hipStreamSynchronize(ctx->stream[1]);
        if (ctx->comm_hosted) delete (FprHosted*)ctx->comm;
        else ncclCommDestroy(comm_of(ctx));
        ctx->comm = nullptr;
        ctx->comm_hosted = false;
    }
    ctx->comm_rank = 0;
    ctx->comm_size = 1;
    return FPR_OK;
}

extern "C" int fpr_comm_rank(fpr_ctx* ctx) { return ctx ? ctx->comm_rank : -1; }
extern "C" int fpr_comm_size(fpr_ctx* ctx) { return ctx ? ctx->comm_size : -1; }

// MPI.Dims_create-like balanced factorisation, non-increasing: 2 -> (2,1,1), 4 -> (2,2,1), 8 -> (2,2,2)
// (the table of part1_scaling_experiments.jl:35-41).  Entries of `dims` that are > 0 are kept.
static bool dims_create(int nprocs, int dims[3])
{
    int fixed = 1, nfree = 0;
    for (int d = 0; d < 3; ++d) {
        if (dims[d] > 0) fixed *= dims[d];
        else ++nfree;
    }
    if (fixed <= 0 || nprocs % fixed != 0) return false;
    int rest = nprocs / fixed;
    if (nfree == 0) return rest == 1;
    int f[3] = {1, 1, 1};
    std::vector<int> primes;
    for (int p = 2; p * p <= rest; ++p)
        while (rest % p == 0) { primes.push_back(p); rest /= p; }
    if (rest > 1) primes.push_back(rest);
    for (int i = (int)primes.size() - 1; i >= 0; --i) {   // largest prime first onto the smallest factor
        int m = 0;
        for (int k = 1; k < nfree; ++k)
            if (f[k] < f[m]) m = k;
        f[m] *= primes[i];
    }
    for (int a = 0; a < nfree; ++a)   // non-increasing
        for (int b = a + 1; b < nfree; ++b)
            if (f[b] > f[a]) { int t = f[a]; f[a] = f[b]; f[b] = t; }
    int k = 0;
    for (int d = 0; d < 3; ++d)
        if (dims[d] <= 0) dims[d] = f[k++];
    return true;
}

// init_global_grid(nx, ny, nz; dimx, dimy, dimz, periodx, periody, periodz) -> me, dims, nprocs, coords
// (ImplicitGlobalGrid; overlap 2, i.e. one halo cell per side).  Ranks are laid out in MPI Cartesian order (last
// dimension fastest).  Without a communicator (fpr_comm_init not called) the grid is the single-rank grid.
extern "C" int fpr_grid_init(fpr_ctx* ctx, int nx, int ny, int nz, int dimx, int dimy, int dimz, int periodx, int periody,
                             int periodz, int* me_out, int* dims_out, int* nprocs_out, int* coords_out)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, nx >= 3 && ny >= 3 && nz >= 3, "grid must be at least 3^3");
    FPR_HIP(ctx, hipSetDevice(ctx->device));
    grid_free(ctx);
    FprGrid& g = ctx->grid;
    g.n[0] = nx; g.n[1] = ny; g.n[2] = nz;
    g.dims[0] = dimx; g.dims[1] = dimy; g.dims[2] = dimz;
    g.periods[0] = periodx != 0; g.periods[1] = periody != 0; g.periods[2] = periodz != 0;
    const int np = ctx->comm_size, me = ctx->comm_rank;
    if (!dims_create(np, g.dims)) return fpr_fail(ctx, FPR_ERR_INVALID, "dims (%d,%d,%d) do not match %d ranks", dimx, dimy, dimz, np);
    g.coords[0] = me / (g.dims[1] * g.dims[2]);
    g.coords[1] = (me / g.dims[2]) % g.dims[1];
    g.coords[2] = me % g.dims[2];
    const size_t plane[3] = {(size_t)ny * nz, (size_t)nx * nz, (size_t)nx * ny};
    for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
            int c[3] = {g.coords[0], g.coords[1], g.coords[2]};
            c[d] += side ? 1 : -1;
            int nb = -1;
            if (c[d] >= 0 && c[d] < g.dims[d]) nb = (c[0] * g.dims[1] + c[1]) * g.dims[2] + c[2];
            else if (g.periods[d]) {
                c[d] = (c[d] + g.dims[d]) % g.dims[d];
                nb = (c[0] * g.dims[1] + c[1]) * g.dims[2] + c[2];
            }
            // measurement aid ("grid_drop_faces", bit 2*dim+side): a periodic single rank has two faces per dimension; dropping
            // one of each gives the face set of a corner rank of a (2,2,2) decomposition (tools/exp_step2_faces.py)
            if ((fpr_opt(ctx, "grid_drop_faces", 0) >> (2 * d + side)) & 1) nb = -1;
            g.nb[2 * d + side] = nb;
            if (nb >= 0 && d != 2) {
                FPR_HIP(ctx, hipMalloc(&g.sendbuf[2 * d + side], plane[d] * sizeof(double)));
                FPR_HIP(ctx, hipMalloc(&g.recvbuf[2 * d + side], plane[d] * sizeof(double)));
            }
        }
    for (int f = 0; f < 6; ++f)
        if (g.nb[f] >= 0 && !ctx->comm) return fpr_fail(ctx, FPR_ERR_INVALID, "neighbours need a communicator: call fpr_comm_init first");
    g.on = true;
    if (me_out) *me_out = me;
    if (nprocs_out) *nprocs_out = np;
    for (int d = 0; d < 3; ++d) {
        if (dims_out) dims_out[d] = g.dims[d];
        if (coords_out) coords_out[d] = g.coords[d];
    }
    return FPR_OK;
}

// nx_g(), ny_g(), nz_g(): dims*(n-2)+2 (non-periodic) / dims*(n-2) (periodic) -- ImplicitGlobalGrid, overlap 2;
// neighbours: rank per face (face = 2*dim + side) or -1
extern "C" int fpr_grid_info(fpr_ctx* ctx, int* n_g_out, int* neighbors_out)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, ctx->grid.on, "fpr_grid_init has not been called");
    const FprGrid& g = ctx->grid;
    for (int d = 0; d < 3; ++d)
        if (n_g_out) n_g_out[d] = g.dims[d] * (g.n[d] - 2) + (g.periods[d] ? 0 : 2);
    for (int f = 0; f < 6; ++f)
        if (neighbors_out) neighbors_out[f] = g.nb[f];
    return FPR_OK;
}

static int check_grid(fpr_ctx* ctx, const double* A, int nx, int ny, int nz)
{
    FPR_REQUIRE(ctx, A, "null pointer");
    FPR_REQUIRE(ctx, ctx->grid.on, "fpr_grid_init has not been called");
    FPR_REQUIRE(ctx, nx == ctx->grid.n[0] && ny == ctx->grid.n[1] && nz == ctx->grid.n[2],
                "array size differs from the size given to fpr_grid_init");
    return FPR_OK;
}

// One group of sends / receives for the faces in `mask` that have a neighbour.  Order inside the group: receives
// low side first, sends high side first -- when both neighbours of a dimension are the same rank (periodic with
// dims <= 2, including a rank that is its own neighbour) RCCL matches the k-th send to a peer with the k-th receive
// from it, so the high plane lands in the peer's low halo and vice versa.
static int post_group(fpr_ctx* ctx, double* A, int mask, hipStream_t s, const double* const* xsend = nullptr, double* const* xrecv = nullptr,
                      const double* const* zsend = nullptr, double* const* zrecv = nullptr)
{   // xsend / xrecv: buffers for the two x-faces instead of the grid's packed-plane buffers; zsend / zrecv: planes of the caller
    // instead of the second / last planes of A
    const FprGrid& g = ctx->grid;
    const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
    const size_t pz = (size_t)nx * ny;
    const size_t count[3] = {(size_t)ny * nz, (size_t)nx * nz, pz};
    bool any = false;
    for (int f = 0; f < 6; ++f) any |= ((mask >> f) & 1) && g.nb[f] >= 0;
    if (!any) return FPR_OK;
    if (int rc = x_group_start(ctx)) return rc;
    // an error inside the group must not leave it open (later RCCL calls of this thread would be deferred for ever):
    // remember the first failure, always close the group, then report
    ncclResult_t first = ncclSuccess;
    int hrc = FPR_OK;     // (hosted transport: operations inside a group are only recorded, this stays FPR_OK)
    const char* what = "";
    for (int d = 0; d < 3 && first == ncclSuccess; ++d)
        for (int side = 0; side < 2 && first == ncclSuccess; ++side) {
            const int f = 2 * d + side;
            if (!((mask >> f) & 1) || g.nb[f] < 0) continue;
            double* dst = d == 2 ? (zrecv ? zrecv[side] : A + (side ? (size_t)(nz - 1) * pz : 0)) : ((d == 0 && xrecv) ? xrecv[side] : g.recvbuf[f]);
            first = x_recv(ctx, dst, count[d], g.nb[f], s, &hrc);
            what = "ncclRecv";
        }
    for (int d = 0; d < 3 && first == ncclSuccess; ++d)
        for (int side = 1; side >= 0 && first == ncclSuccess; --side) {
            const int f = 2 * d + side;
            if (!((mask >> f) & 1) || g.nb[f] < 0) continue;
            const double* src = d == 2 ? (zsend ? zsend[side] : A + (side ? (size_t)(nz - 2) * pz : pz)) : ((d == 0 && xsend) ? xsend[side] : g.sendbuf[f]);
            first = x_send(ctx, src, count[d], g.nb[f], s, &hrc);
            what = "ncclSend";
        }
    ncclResult_t end = ncclSuccess;
    if (int rc = x_group_end(ctx, &end)) return rc;
    if (hrc != FPR_OK) return hrc;
    if (first != ncclSuccess)
        return fpr_fail(ctx, FPR_ERR_RCCL, "%s:%d %s -> %s (group closed: %s)", __FILE__, __LINE__, what, ncclGetErrorString(first),
                        ncclGetErrorString(end));
    if (end != ncclSuccess) return fpr_fail(ctx, FPR_ERR_RCCL, "%s:%d ncclGroupEnd -> %s", __FILE__, __LINE__, ncclGetErrorString(end));
    return FPR_OK;
}

// Split form for overlap with the interior update (role of @hide_communication, part1_kernel_programming.jl:185-188):
//   begin: [compute] pack the x / y planes -> [comm] waits for compute, one group of sends / receives
//   end  : [compute] waits for comm, unpacks the received x / y planes into the halo planes
// Whatever the caller enqueues on the compute stream between the two overlaps the transfers.  All faces of `mask`
// travel concurrently, so edge and corner halo cells are NOT refreshed (the 7-point stencil reads none of them).
extern "C" int fpr_halo_exchange3d_begin(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (int rc = fpr_diffusion3d_join(ctx)) return rc;   // a fused pair left on the core / comm streams (fpr_diffusion3d_step2_halo)
    if (int rc = check_grid(ctx, A, nx, ny, nz)) return rc;
    const FprGrid& g = ctx->grid;
    bool any = false;
    for (int f = 0; f < 6; ++f) {
        if (!((face_mask >> f) & 1) || g.nb[f] < 0) continue;
        any = true;
        if ((f >> 1) != 2)
            if (int rc = fpr_halo_pack3d(ctx, A, nx, ny, nz, f, g.sendbuf[f], 0)) return rc;
    }
    if (!any) return FPR_OK;
    if (int rc = fpr_stream_wait(ctx, 1, 0)) return rc;
    return post_group(ctx, A, face_mask, ctx->stream[1]);
}

extern "C" int fpr_halo_exchange3d_end(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (int rc = check_grid(ctx, A, nx, ny, nz)) return rc;
    const FprGrid& g = ctx->grid;
    bool any = false;
    for (int f = 0; f < 6; ++f) any |= ((face_mask >> f) & 1) && g.nb[f] >= 0;
    if (!any) return FPR_OK;
    if (int rc = fpr_stream_wait(ctx, 0, 1)) return rc;
    for (int f = 0; f < 4; ++f)
        if (((face_mask >> f) & 1) && g.nb[f] >= 0)
            if (int rc = fpr_halo_unpack3d(ctx, A, nx, ny, nz, f, g.recvbuf[f], 0)) return rc;
    return FPR_OK;
}

// The whole exchange on the COMM stream, in stream order: pack the x / y planes, one group of sends / receives, unpack.
// For callers that keep a chain of thin-box launches and exchanges on the comm stream beside one long launch on the
// compute stream (the fused pairs of a decomposed run, GlobalGrid.step2): no event between the two streams per exchange.
// As with _begin/_end all faces travel concurrently: edge and corner halo cells are not refreshed.
int fprx_halo_exchange3d_comm_x(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask, const double* const xsend[2],
                                double* const xrecv[2])
{
    if (!ctx) return FPR_ERR_INVALID;
    if (int rc = check_grid(ctx, A, nx, ny, nz)) return rc;
    const FprGrid& g = ctx->grid;
    const bool xown = xsend && xrecv;   // the x-planes are the caller's strips: sent and received as they are
    bool any = false;
    for (int f = 0; f < 6; ++f) {
        if (!((face_mask >> f) & 1) || g.nb[f] < 0) continue;
        any = true;
        if ((f >> 1) != 2 && !((f >> 1) == 0 && xown))
            if (int rc = fpr_halo_pack3d(ctx, A, nx, ny, nz, f, g.sendbuf[f], 1)) return rc;
    }
    if (!any) return FPR_OK;
    if (int rc = post_group(ctx, A, face_mask, ctx->stream[1], xown ? xsend : nullptr, xown ? xrecv : nullptr)) return rc;
    for (int f = xown ? 2 : 0; f < 4; ++f)
        if (((face_mask >> f) & 1) && g.nb[f] >= 0)
            if (int rc = fpr_halo_unpack3d(ctx, A, nx, ny, nz, f, g.recvbuf[f], 1)) return rc;
    return FPR_OK;
}

int fprx_exchange_zplanes(fpr_ctx* ctx, const double* const zsend[2], double* const zrecv[2])
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, ctx->grid.on && zsend && zrecv, "fpr_grid_init has not been called / null plane lists");
    for (int side = 0; side < 2; ++side)
        FPR_REQUIRE(ctx, ctx->grid.nb[4 + side] < 0 || (zsend[side] && zrecv[side]), "a z-face with a neighbour needs its two planes");
    return post_group(ctx, nullptr, 0x30, ctx->stream[1], nullptr, nullptr, zsend, zrecv);
}

extern "C" int fpr_halo_exchange3d_comm(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask)
{
    return fprx_halo_exchange3d_comm_x(ctx, A, nx, ny, nz, face_mask, nullptr, nullptr);
}

// update_halo!(A): dimension by dimension (x, then y, then z), each after the previous one has been unpacked, as
// ImplicitGlobalGrid does -- planes sent in a later dimension carry the halo cells received in an earlier one, so edge
// and corner halo cells end up consistent too.  Ordered on the compute stream; returns without synchronising.
extern "C" int fpr_halo_exchange3d(fpr_ctx* ctx, double* A, int nx, int ny, int nz)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (int rc = check_grid(ctx, A, nx, ny, nz)) return rc;
    for (int d = 0; d < 3; ++d) {
        const int mask = 3 << (2 * d);
        if (int rc = fpr_halo_exchange3d_begin(ctx, A, nx, ny, nz, mask)) return rc;
        if (int rc = fpr_halo_exchange3d_end(ctx, A, nx, ny, nz, mask)) return rc;
    }
    return FPR_OK;
}

// MPI.Allreduce!(x, +, comm_cart) on `count` device doubles, in place, on the chosen stream (no host sync)
extern "C" int fpr_allreduce_sum_dev(fpr_ctx* ctx, double* x_dev, int count, int stream_sel)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, x_dev && count >= 1 && (stream_sel == 0 || stream_sel == 1), "pointer / count / stream");
    if (int rc = fpr_diffusion3d_join(ctx)) return rc;
    if (!ctx->comm) return FPR_OK;   // single rank without a communicator
    return x_allreduce(ctx, x_dev, count, ctx->stream[stream_sel]);
}

// part1_utils.jl:38 as the reference calls it: one host Float64, summed over all ranks, back on the host.
// Ordered behind everything enqueued so far on the compute stream; synchronises.
extern "C" int fpr_allreduce_sum1(fpr_ctx* ctx, double* x_host_inout)
{
    if (!ctx) return FPR_ERR_INVALID;
    FPR_REQUIRE(ctx, x_host_inout, "null pointer");
    if (int rc = fpr_diffusion3d_join(ctx)) return rc;
    if (!ctx->comm) return FPR_OK;   // single rank without a communicator
    // like every RCCL operation of this communicator it runs on the comm stream, behind what the compute stream holds
    double* d = ctx->scalars + 48;
    ctx->host_scalars[48] = *x_host_inout;
    if (int rc = fpr_stream_wait(ctx, 1, 0)) return rc;
    FPR_HIP(ctx, hipMemcpyAsync(d, ctx->host_scalars + 48, sizeof(double), hipMemcpyHostToDevice, ctx->stream[1]));
    if (int rc = x_allreduce(ctx, d, 1, ctx->stream[1])) return rc;
    FPR_HIP(ctx, hipMemcpyAsync(ctx->host_scalars + 48, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream[1]));
    FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[1]));
    *x_host_inout = ctx->host_scalars[48];
    return FPR_OK;
}

// gather!(A, A_global) (part1_kernel_programming.jl:144,223): rank 0 receives every rank's local array, halos
// included, and places it as a block of the HOST array A_global (nx*dims[0], ny*dims[1], nz*dims[2]), column-major,
// in Cartesian order.  A_global_host is read on rank 0 only (NULL elsewhere).  Synchronises; not on the hot path.
extern "C" int fpr_gather3d(fpr_ctx* ctx, const double* A, int nx, int ny, int nz, double* A_global_host)
{
    if (!ctx) return FPR_ERR_INVALID;
    if (int rc = check_grid(ctx, A, nx, ny, nz)) return rc;
    if (int rc = fpr_diffusion3d_join(ctx)) return rc;
    FprGrid& g = ctx->grid;
    const size_t n = (size_t)nx * ny * nz;
    const int np = ctx->comm_size, me = ctx->comm_rank;
    if (int rc = fpr_stream_wait(ctx, 1, 0)) return rc;   // comm stream (all RCCL calls), behind the producers of A
    if (me != 0) {
        int hrc = FPR_OK;
        FPR_NCCL(ctx, x_send(ctx, A, n, 0, ctx->stream[1], &hrc));
        if (hrc != FPR_OK) return hrc;
        FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[1]));
        return FPR_OK;
    }
    FPR_REQUIRE(ctx, A_global_host, "rank 0 needs the global host array");
    if (np > 1 && !g.stage) FPR_HIP(ctx, hipMalloc(&g.stage, n * sizeof(double)));
    // every rank's block goes from device memory straight into its place in the (strided) global host array, one z-plane
    // per 2-D copy: no field-sized host staging buffer (a 512^3 block is 1 GiB)
    const size_t gx = (size_t)nx * g.dims[0], gy = (size_t)ny * g.dims[1];
    for (int r = 0; r < np; ++r) {
        const double* src = A;
        if (r != 0) {
            int hrc = FPR_OK;
            FPR_NCCL(ctx, x_recv(ctx, g.stage, n, r, ctx->stream[1], &hrc));
            if (hrc != FPR_OK) return hrc;
            src = g.stage;
        }
        const int c[3] = {r / (g.dims[1] * g.dims[2]), (r / g.dims[2]) % g.dims[1], r % g.dims[2]};
        for (int k = 0; k < nz; ++k) {
            double* dst = A_global_host + (size_t)c[0] * nx + gx * ((size_t)c[1] * ny + gy * ((size_t)c[2] * nz + k));
            FPR_HIP(ctx, hipMemcpy2DAsync(dst, gx * sizeof(double), src + (size_t)nx * ny * k, (size_t)nx * sizeof(double),
                                          (size_t)nx * sizeof(double), (size_t)ny, hipMemcpyDeviceToHost, ctx->stream[1]));
        }
        FPR_HIP(ctx, hipStreamSynchronize(ctx->stream[1]));   // the staging block is reused by the next rank
    }
    return FPR_OK;
}
