/*
 * fpr.h -- C ABI of libfpr_hip.so: the MI355X (gfx950) implementation of the stencil hot path of
 * ntselepidis/FinalProjectRepo.jl.  One symbol per row of SURVEY.md section 8(a) plus lifecycle.
 *
 * The reference has no FFI: its boundary is the ParallelStencil call convention
 *     @parallel [blocks threads shmem=n] kernel(args...)
 * on Data.Array (= device Float64 arrays).  Each entry point below replaces one such kernel /
 * wrapper and keeps its argument list (same order, same meaning); the launch-geometry arguments of
 * the macro have no counterpart (the library picks its own).  The Julia binding a maintainer adds is
 * in julia/FPRHip.jl and INTEGRATION.md.  Citations are file:line into the reference repository.
 *
 * Conventions
 *   - all field arrays: Float64, column-major (Julia layout, ix fastest), DEVICE pointers, allocated
 *     and freed by the host language (AMDGPU.jl ROCArray / torch tensor); the library never retains
 *     them past the call.
 *   - every call enqueues on the context's compute stream and returns; only calls that hand a scalar
 *     back to the host (out-parameters named *_host) synchronise that stream.
 *   - return value: 0 = FPR_OK, negative = error (fpr_last_error gives the text).
 *   - arithmetic is compiled without FMA contraction: pointwise results are bit-identical to the
 *     reference's CPU expressions; reductions differ only by summation order.
 */
#ifndef FPR_H
#define FPR_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fpr_ctx fpr_ctx;

enum {
    FPR_OK = 0,
    FPR_ERR_INVALID = -1,   /* bad argument (null pointer, size < 3, ...)                            */
    FPR_ERR_HIP = -2,       /* a HIP runtime call failed                                             */
    FPR_ERR_NOT_POW2 = -3,  /* reference: error("ERROR:not a power of 2")  multigrid.jl:95-97, :103   */
    FPR_ERR_ASSERT = -4,    /* reference: @assert on coarse_solve_size      multigrid.jl:45-46        */
    FPR_ERR_NO_DEVICE = -5, /* no HIP device visible                                                 */
    FPR_ERR_RCCL = -6       /* an RCCL call failed (fpr_last_error carries ncclGetErrorString)       */
};

/* coarse solvers -- multigrid.jl:10-13 (CoarseSolver_t) */
enum { FPR_COARSE_JACOBI = 0, FPR_COARSE_CG = 1 };

/* ---- lifecycle (replaces @init_parallel_stencil / select_device / @synchronize) ------------------ */

/* device: HIP device ordinal.  compute_stream / comm_stream: hipStream_t handles owned by the caller
 * (e.g. torch.cuda.Stream().cuda_stream, AMDGPU.jl HIPStream) or NULL to let the library create its own. */
int fpr_ctx_create(fpr_ctx** out, int device, void* compute_stream, void* comm_stream);
int fpr_ctx_destroy(fpr_ctx* ctx);
/* @synchronize()  -- ParallelStencil; multigrid.jl:65, krylov.jl:51 */
int fpr_synchronize(fpr_ctx* ctx);
const char* fpr_last_error(fpr_ctx* ctx);
const char* fpr_version(void);
/* select the diffusion kernel variant (tuning knob, 0 = library default) */
int fpr_set_option(fpr_ctx* ctx, const char* key, long value);
long fpr_get_option(fpr_ctx* ctx, const char* key);

/* Per-launch timing of the dominant kernels with hipEvents recorded on the launch stream (used by bench.py for the
 * roofline figures).  enable=1 starts recording an event pair around every launch of the kernels below (up to 8192
 * pairs); read synchronises and returns the summed duration and the number of launches of one kind (-1 = all). */
enum {
    FPR_KT_DIFF3_STEP = 0,    /* k_diff3_march: one pseudo-iteration per launch                               */
    FPR_KT_DIFF3_STEP2 = 1,   /* k_diff3_march2: two pseudo-iterations per launch                             */
    FPR_KT_MG_PRE = 2,        /* finest level of a V-cycle: 2 sweeps + residual + injection in one pass       */
    FPR_KT_MG_POST = 3,       /* finest level of a V-cycle: prolongation + correction + 2 sweeps (+ norm)     */
    FPR_KT_MG_SEAM = 4,       /* finest level between two V-cycles of fpr_mgsolve2d: post pair of cycle k + norm +
                                 pre pair + residual + injection of cycle k+1 in one pass (k_seam_march)         */
    FPR_KT_MG_CG = 5,         /* coarse solve by cg! as ONE persistent launch (k_cg_persistent): a launch = a solve      */
    FPR_KT_MG_PATCH = 6,      /* coarse solve by damped Jacobi on a large coarse grid: one launch = up to 32 groups of 8 sweeps
                                 (k_jacobi_persist_tag; options mg_jacobi_persist = 0 or handoff_fences = 1: one launch per 8 sweeps, k_jacobi_patch) */
    FPR_KT_DIFF3_STEP3 = 8,   /* k_diff3_march3: three pseudo-iterations per launch                           */
    FPR_KT_DIFF3_CORE = 7     /* fpr_diffusion3d_step2_core: the core launch of a decomposed run's fused pair (the shell
                                 launches beside it stay FPR_KT_DIFF3_STEP / _STEP2 and OVERLAP it in time)              */
};
int fpr_kernel_timer(fpr_ctx* ctx, int enable);
int fpr_kernel_timer_read(fpr_ctx* ctx, int kind, double* total_ms_host, long* count_host);

/* ---- Part 1: 3D pseudo-transient diffusion ------------------------------------------------------- */

/* A1/A2: diffusion_3D_step_tau(Ht, Htau, Htau2, dHdtau, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
 * part1_kernel_programming.jl:46-58 (shared-memory variant :75-97 has the same arithmetic).
 * Interior cells only; boundary cells of Htau2 / dHdtau are left untouched. */
int fpr_diffusion3d_step(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2, double* dHdtau,
                         int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz,
                         double D_dx, double D_dy, double D_dz);

/* Same update fused with the convergence norm of part1_kernel_programming.jl:191 /
 * part1_utils.jl:36-37: *sumsq_dev (DEVICE double) = sum over interior cells of (dHdtau*scale)^2.
 * (boundary entries of residual_H are zero in the reference, so this equals sum(abs2, residual_H*dt).) */
int fpr_diffusion3d_step_norm(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2, double* dHdtau,
                              int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz,
                              double D_dx, double D_dy, double D_dz, double scale, double* sumsq_dev);

/* Same as fpr_diffusion3d_step_norm, but the sum is handed to the HOST: the reduction writes straight into
 * pinned host memory and the call synchronises the compute stream (one round trip per pseudo-iteration,
 * as the reference's `err = dist_norm_L2(...)` at part1_kernel_programming.jl:191 requires). */
int fpr_diffusion3d_step_norm_host(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2, double* dHdtau,
                                   int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz,
                                   double D_dx, double D_dy, double D_dz, double scale, double* sumsq_host);

/* THREE iterations in one launch (k_diff3_march3; the same loop body :179-192 three times): Htau -> Hout with the two fields in
 * between never materialised -- 32 bytes per cell and launch for three iterations.  Htau and Hout are the reference's TWO ping-pong
 * buffers in either order (Hout must carry the boundary values of the buffer the reference would write the third iteration into, i.e.
 * the other one of the pair; its boundary cells are read, its interior written): no third work buffer.  dHdtau (nullable) receives the
 * residual of the third iteration; sumsq3_dev (nullable): three device doubles, sum((r*scale)^2) of each iteration.  Fields bit for bit
 * three fpr_diffusion3d_step calls.  Requirements: nx even >= 128, ny >= 24, nz >= 5, 16-byte aligned arrays, nx*ny*128 < 2^31 -- query
 * with fpr_diffusion3d_can_step3.  At 512^3 on MI355X: 0.92 ms per launch against 0.76-0.86 for two iterations (profiles/r6_march3_*).
 * Option diff3_fuse3 [1]. */
int fpr_diffusion3d_can_step3(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hout, const double* dHdtau, int nx,
                              int ny, int nz);
int fpr_diffusion3d_step3(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Hout, double* dHdtau, int nx, int ny, int nz,
                          double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx, double D_dy, double D_dz,
                          double scale, double* sumsq3_dev);
/* A1x2: TWO pseudo-transient iterations in one pass over memory (temporal blocking of two trips through the loop body
 * part1_kernel_programming.jl:179-192).  Equivalent, bit for bit, to
 *     fpr_diffusion3d_step(Ht, Htau -> Hmid, dHdtau);  fpr_diffusion3d_step(Ht, Hmid -> Hout, dHdtau)
 * except that the intermediate field is never written: of Hmid only the BOUNDARY cells are read (the reference's
 * kernel writes interior cells only, so the boundary cells of its second work buffer are what the intermediate field
 * carries there).  Hout receives interior cells only and must already hold Htau's boundary values to stand in for the
 * reference's first work buffer afterwards.  Htau, Hmid, Hout: three distinct buffers.  dHdtau = residual of the
 * second iteration; it may be NULL, then the residual is not stored (its norm still lands in sumsq2_dev: a solver loop
 * reads nothing else of it, and the launch moves 24 instead of 32 bytes per cell).  sumsq2_dev (may be NULL): two device doubles, sum((r*scale)^2) of the first and of the second
 * iteration (deterministic two-stage reductions).  "Bit for bit" is a statement about the FIELDS (Hout, dHdtau): the
 * two sums are accumulated as sum(r*r) per lane with scale^2 applied once at the end, the single-step kernel adds
 * (r*scale)^2 per cell -- both deterministic, equal to about 1e-13 relative, like any two summation orders.  Requirements: nx even and >= 128, ny >= 16, 16-byte aligned
 * arrays, nx*ny*96 < 2^31 -- query with fpr_diffusion3d_can_step2 (1 = supported, 0 = use two single steps).
 * _box: output box [lo, hi) as fpr_diffusion3d_step_box; sums are ACCUMULATED into sumsq2_dev.
 * Placement: the launch streams Ht and Htau in and Hout and dHdtau out at equal offsets; on MI355X its time depends on which
 * physical pages the four allocations received (0.775 ms at 512^3 when they all differ in their placement label, 0.85-0.91 ms when
 * they agree) -- a host that owns its arrays picks them from a pool of candidates once.
 * Lanes of a tile that hold no needed cell are switched off for the march. */
int fpr_diffusion3d_can_step2(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid, const double* Hout,
                              const double* dHdtau, int nx, int ny, int nz);
int fpr_diffusion3d_step2(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid, double* Hout,
                          double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy,
                          double _dz, double D_dx, double D_dy, double D_dz, double scale, double* sumsq2_dev);
int fpr_diffusion3d_step2_box(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid, double* Hout,
                              double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy,
                              double _dz, double D_dx, double D_dy, double D_dz, const int lo[3], const int hi[3],
                              double scale, double* sumsq2_dev, int stream_sel);
/* _box2: the box [lo, hi) plus a second z-range [zlo2, zhi2) with the same x/y extent in ONE launch (the two thin
 * slabs next to the z-halos of a rank, role of @hide_communication's boundary width); the ranges must not overlap. */
int fpr_diffusion3d_step2_box2(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid, double* Hout,
                               double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy,
                               double _dz, double D_dx, double D_dy, double D_dz, const int lo[3], const int hi[3],
                               int zlo2, int zhi2, double scale, double* sumsq2_dev, int stream_sel);
/* _core: the CORE box of a decomposed run's pair (everything inside the one-cell shell next to the neighbours), launched
 * so that `reserve_cus` compute units keep no workgroup of it: the shell launches and the RCCL send / receive kernels of
 * the pair's two exchanges run BESIDE it on the comm stream instead of draining behind a grid that fills the device --
 * the role of @hide_communication (part1_kernel_programming.jl:185-188) for two iterations at once.  The (tile, plane)
 * space of the box is cut into (device slots - reserve) balanced ranges, one workgroup each (k_diff3_march2<.., BAL>);
 * reserve_cus = 0 or a box whose plain grid leaves that many units idle anyway: exactly _step2_box.  Same results.
 * accumulate = 0: the two sums are WRITTEN to sumsq2_dev (no zeroing launch in front of the pair's longest kernel). */
int fpr_diffusion3d_step2_core(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hmid, double* Hout,
                               double* dHdtau, int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy,
                               double _dz, double D_dx, double D_dy, double D_dz, const int lo[3], const int hi[3],
                               double scale, double* sumsq2_dev, int stream_sel, int reserve_cus, int accumulate);
/* _halo: TWO iterations on a rank WITH neighbours (fpr_grid_init), halos of Hout refreshed -- what `@hide_communication` + `update_halo!`
 * (part1_kernel_programming.jl:185-190) do for one iteration, for two: the device is split (fpr_reserve_comm_cus, 16 or 24 units for
 * the comm stream by the rank's faces), the core of the local grid runs as ONE fused launch on the core stream, and beside it on the
 * comm stream run the first iteration on the one-cell shell (level 1 into Hmid), the exchange of Hmid's planes, the fused launches on
 * the shell and the exchange of Hout's planes.  Same
 * results as two fpr_diffusion3d_step calls each followed by fpr_halo_exchange3d of the written buffer; Hout must carry Htau's
 * physical-boundary values; sumsq2_dev (nullable) receives the LOCAL sums of both iterations; dHdtau may be NULL (residual not
 * stored).  A device already split with more units and a masked core stream (32: a chain of _step3_halo calls) is used as it is.
 * join = 0 leaves the pair on the core /
 * comm streams: the next _halo call continues from there; fpr_diffusion3d_join waits for it (the HOST waits for the core and comm
 * streams: a wait parked on the compute stream while pairs are in flight slows them
 * instead); call it before anything else reads OR WRITES the fields or the sums.  Without neighbours: fpr_diffusion3d_step2.
 * x-faces (csrc/diffusion3d_xstrip.hpp; option diff3_xstrips, default 1): the shell column next to an x-neighbour, the three columns
 * around it and the planes that travel live in compact strips (ny*nz doubles per column, kept in the context): both iterations of
 * the shell cells read and write contiguous memory beside the core launch, RCCL sends and receives the strips as they are (no pack /
 * unpack kernels), and the strided accesses to the fields -- the received halo column, the shell column and its residual into Hout /
 * dHdtau, the columns next to the face out of Hout for the next pair -- are ONE launch between two core launches.  While a chain of
 * pairs (join = 0) continues -- this call's Htau is the pending pair's Hout, same Ht -- the strips are reused; otherwise they are
 * gathered afresh before the core launch.  Edge and corner cells of the halo planes are not
 * refreshed (as with fpr_halo_exchange3d_begin / _end: all faces travel at once).  diff3_xstrips = 0: the x-shell in the field
 * (narrow-box kernels, pack / unpack kernels; rounds 2-3). */
int fpr_diffusion3d_step2_halo(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Hmid, double* Hout, double* dHdtau,
                               int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx,
                               double D_dy, double D_dz, double scale, double* sumsq2_dev, int join);
int fpr_diffusion3d_join(fpr_ctx* ctx);
/* _step3_halo: THREE iterations on a rank of a z-slab decomposition (process grid (1,1,N): neighbours on z-faces only), halos of Hout
 * refreshed -- three trips through part1_kernel_programming.jl:185-190.  The core (planes [3, nz-3)) is ONE launch of the three-step
 * kernel on the core stream (everything it needs of the two fields in between it computes from Htau itself, whose halo planes are
 * valid); beside it, on the comm stream's 32 units, the two planes next to each z-face with a neighbour go through three rounds of
 * single-step launches on 6-plane slabs kept in the context, each followed by the exchange of ONE plane per face (iterate 1, iterate 2,
 * Hout) -- one exchange per iteration, as in the reference.  Same results as three fpr_diffusion3d_step calls each followed by
 * fpr_halo_exchange3d of the written buffer.  Htau and Hout are the reference's two ping-pong buffers in either order (as for
 * fpr_diffusion3d_step3); sumsq3_dev (nullable) receives the LOCAL sums of the three iterations; dHdtau may be NULL.  join as for
 * _step2_halo (the two forms may alternate in one chain).  _can_step3_halo: 1 if the grid (fpr_grid_init) has no x / y neighbours,
 * nz >= 12 and fpr_diffusion3d_can_step3 holds; without any neighbour the call is fpr_diffusion3d_step3. */
int fpr_diffusion3d_can_step3_halo(fpr_ctx* ctx, const double* Ht, const double* Htau, const double* Hout, const double* dHdtau,
                                   int nx, int ny, int nz);
int fpr_diffusion3d_step3_halo(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Hout, double* dHdtau, int nx, int ny, int nz,
                               double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx, double D_dy, double D_dz,
                               double scale, double* sumsq3_dev, int join);

/* A5: the single-rank host loop of diffusion_3D_kernel_programming (part1_kernel_programming.jl:166-204) in
 * native code: for each of `nt` physical steps iterate the fused update until err <= tol (err =
 * sqrt(sum((dHdtau*dt)^2)) / sqrt(total_N), :191) or iter_max / fixed_iters, then `Ht .= Htau` (:203).
 * check_every = n evaluates the norm (one host round trip) only every n-th pseudo-iteration (1 = reference).
 * fixed_iters > 0 runs exactly that many iterations per step.  iters_host / err_host: nt entries each.
 * The two work buffers swap roles every iteration (:190): *swapped_host = 1 means the current Htau lives in the
 * memory passed as Htau2 (and the last residual is in dHdtau either way).
 * Htau3 (nullable): a THIRD field-sized work buffer owned by the caller -- the library allocates no field memory.
 * With it, pairs of iterations run as one fused launch (fpr_diffusion3d_step2; an iteration whose norm ends the loop
 * is replayed alone, so fields, iteration counts and errors are those of the plain loop; option "diff3_fuse2" = 0
 * turns this off).  Inside the loop the pairs do not store dHdtau (the loop reads only its norm); a call that ends on
 * a pair replays that pair once with the store, so dHdtau holds the reference's residual_H on return (option
 * "diff3_lazy_residual" = 0: every launch stores it).  The loop's exit test (err > tol, :179) is evaluated on the device and
 * up to "diff3_ahead" (default 2) pairs are enqueued before the host has seen the norm of the oldest; pairs behind the
 * iteration that ends the loop return at once (0: the host waits for every norm).  NULL: one iteration per launch.
 * Non-NULL for a problem the fused kernel cannot serve (fpr_diffusion3d_can_step2 == 0) is FPR_ERR_INVALID, never a
 * silent fallback. After an error return the output arrays are UNDEFINED (launches enqueued ahead of the host's view of the norm
 * are drained before the call returns, but they may have run). */
int fpr_diffusion3d_solve(fpr_ctx* ctx, double* Ht, double* Htau, double* Htau2, double* Htau3, double* dHdtau, int nx, int ny, int nz,
                          double dtau, double _dt, double _dx, double _dy, double _dz, double D_dx, double D_dy, double D_dz,
                          double dt, double total_N, int nt, double tol, long iter_max, long fixed_iters, int check_every,
                          long* iters_host, double* err_host, int* swapped_host);

/* Sub-box form used by the multi-GPU driver to split boundary slabs from the interior
 * (role of @hide_communication (8,8,8), part1_kernel_programming.jl:185-188).  Updates cells with
 * lo[d] <= index < hi[d] (0-based, clipped to the interior).  sumsq_dev may be NULL; when given,
 * the partial sum of this box is ADDED to *sumsq_dev (zero it first).  stream_sel: 0 compute, 1 comm. */
int fpr_diffusion3d_step_box(fpr_ctx* ctx, const double* Ht, const double* Htau, double* Htau2, double* dHdtau,
                             int nx, int ny, int nz, double dtau, double _dt, double _dx, double _dy, double _dz,
                             double D_dx, double D_dy, double D_dz, const int lo[3], const int hi[3],
                             double scale, double* sumsq_dev, int stream_sel);

/* A3 split form (north_star names compute_flux! / compute_dHdtau! / update_H!), clean semantics of
 * part1_array_programming.jl:9-18:  qx (nx-1,ny-2,nz-2), qy (nx-2,ny-1,nz-2), qz (nx-2,ny-2,nz-1),
 * dHdtau (nx-2,ny-2,nz-2). */
int fpr_diffusion3d_flux(fpr_ctx* ctx, double* qx, double* qy, double* qz, const double* Htau,
                         int nx, int ny, int nz, double D, double dx, double dy, double dz);          /* :10-12 */
int fpr_diffusion3d_dHdtau(fpr_ctx* ctx, double* dHdtau, const double* Htau, const double* Ht, const double* qx,
                           const double* qy, const double* qz, int nx, int ny, int nz, double dt, double dx,
                           double dy, double dz);                                                    /* :14-15 */
int fpr_diffusion3d_update(fpr_ctx* ctx, double* Htau, const double* dHdtau, int nx, int ny, int nz,
                           double dtau);                                                             /* :16    */

/* A4: local part of dist_norm_L2(x*scale) -- part1_utils.jl:36-37: sum((x*scale)^2) over n elements.
 * _dev form leaves the sum in device memory (no sync); _host form synchronises. */
int fpr_sumsq_scaled_dev(fpr_ctx* ctx, const double* x, size_t n, double scale, double* out_dev);
int fpr_sumsq_scaled(fpr_ctx* ctx, const double* x, size_t n, double scale, double* out_host);
/* sum(x .* y) -- krylov.jl:64,69,83 */
int fpr_dot(fpr_ctx* ctx, const double* x, const double* y, size_t n, double* out_host);

/* A5: `Ht .= Htau` -- part1_kernel_programming.jl:203 */
int fpr_copy(fpr_ctx* ctx, double* dst, const double* src, size_t n);
int fpr_fill(fpr_ctx* ctx, double* dst, double value, size_t n);
/* the same, and dst += src, on a chosen stream (0 compute, 1 comm, 2 core): small device vectors such as the norm sums of
 * a decomposed run's fused pair, which live on the comm and core streams (GlobalGrid.step2) */
int fpr_fill_on(fpr_ctx* ctx, double* dst, double value, size_t n, int stream_sel);
int fpr_add_on(fpr_ctx* ctx, double* dst, const double* src, size_t n, int stream_sel);

/* A6: init_local_gaussian -- part1_utils.jl:1-12; coord* = 0-based Cartesian coordinates of the shard */
int fpr_init_gaussian3d(fpr_ctx* ctx, double* H, int nx, int ny, int nz, double dx, double dy, double dz,
                        double cx, double cy, double cz, int coordx, int coordy, int coordz);

/* update_halo! building blocks (ImplicitGlobalGrid, called at part1_kernel_programming.jl:182,187):
 * face = 2*dim + side (dim 0..2, side 0 = low, 1 = high).  pack copies the plane the neighbour needs
 * (local index 1 or n-2, 0-based) into a contiguous buffer; unpack writes a received plane into the
 * halo plane (index 0 or n-1).  Plane sizes: ny*nz, nx*nz, nx*ny.  stream_sel as above. */
int fpr_halo_pack3d(fpr_ctx* ctx, const double* A, int nx, int ny, int nz, int face, double* buf, int stream_sel);
int fpr_halo_unpack3d(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face, const double* buf, int stream_sel);
/* order stream `waiter` behind everything enqueued so far on stream `signaller` (0 compute, 1 comm, 2 core) */
int fpr_stream_wait(fpr_ctx* ctx, int waiter, int signaller);
/* Split the device for a decomposed run (role of @hide_communication's boundary width, part1_kernel_programming.jl:185-188,
 * in hardware terms): after fpr_reserve_comm_cus(ctx, k) stream 1 (comm) is a library-owned stream whose kernels -- the shell
 * launches of a fused pair, the pack / unpack kernels, RCCL's send / receive kernels -- run on k compute units, and stream 2
 * (core) one whose kernels run on all the others; before it, and after k = 0, stream 1 is the caller's comm stream and
 * stream 2 is stream 0.  A second queue's workgroups wait for room in the shader engine they were dealt to even when
 * other engines have idle units, so leaving units idle is not enough: the masks make the split explicit.
 * k: a multiple of 8.  For a multiple of 32 the core stream is masked to the other units.  Otherwise (16 for a rank with z-faces
 * only) the core stream keeps every unit and the one-workgroup-per-unit core launch of a fused pair lets the workgroups that land
 * on a comm unit leave at once (the comm stream's units are found by a probe launch; if that fails k is rounded up to 32).
 * fpr_comm_cus: the current k.  fpr_stream_handle: the hipStream_t behind a selector, for host code that enqueues on it. */
int fpr_reserve_comm_cus(fpr_ctx* ctx, int k);
int fpr_comm_cus(fpr_ctx* ctx);
int fpr_stream_handle(fpr_ctx* ctx, int stream_sel, void** hip_stream_out);

/* ---- process / device boundary of the decomposed diffusion path: RCCL over xGMI, one process per GPU ------
 * Replaces ImplicitGlobalGrid + MPI in the reference:
 *   init_global_grid / finalize_global_grid     part1_kernel_programming.jl:100-101,225; part1_array_programming.jl:28-29,89
 *   nx_g() ny_g() nz_g(), x_g y_g z_g            part1_kernel_programming.jl:117; part1_utils.jl:5-7
 *   update_halo!(A)                              part1_kernel_programming.jl:182,187; part1_array_programming.jl:67
 *   MPI.Allreduce!(sq_residual, +, comm_cart)    part1_utils.jl:38
 *   gather!(Array(Ht), H_g)                      part1_kernel_programming.jl:223; part1_array_programming.jl:87
 * Bootstrap: rank 0 calls fpr_comm_get_unique_id and the host language broadcasts the FPR_UNIQUE_ID_BYTES bytes by
 * whatever it has (MPI.Bcast in Julia, a torch.distributed store in Python); every rank then calls fpr_comm_init
 * on a context created on ITS device (select_device() = fpr_ctx_create(device = local rank)). */
#define FPR_UNIQUE_ID_BYTES 128
int fpr_comm_get_unique_id(void* id_out /* FPR_UNIQUE_ID_BYTES */);
int fpr_comm_init(fpr_ctx* ctx, int rank, int nranks, const void* unique_id);
/* A host-staged transport in RCCL's place, for rehearsals and tests on ONE card (RCCL refuses two ranks on one device): the same
 * library code issues the same sends, receives and all-reduces -- face order of an exchange, pack / unpack kernels, strips, gather --
 * but the bytes leave the device through host memory and travel by the host's callbacks (each returns 0 on success):
 *   send(user, peer, buf, bytes)       must not wait for the peer's receive (buffer it); buf is the library's until it returns
 *   recv(user, peer, buf, bytes)       blocks until the next message from `peer` is in buf (per peer: k-th send meets k-th receive)
 *   allreduce(user, x, count)          sum over all ranks, in place, on `count` host doubles
 * Rates through it are NOT measurements.  Everything else (fpr_grid_init, exchanges, fpr_diffusion3d_step2_halo ...) is unchanged. */
typedef int (*fpr_hosted_send_fn)(void* user, int peer, const void* buf, size_t bytes);
typedef int (*fpr_hosted_recv_fn)(void* user, int peer, void* buf, size_t bytes);
typedef int (*fpr_hosted_allreduce_fn)(void* user, double* x, int count);
int fpr_comm_init_hosted(fpr_ctx* ctx, int rank, int nranks, fpr_hosted_send_fn send, fpr_hosted_recv_fn recv,
                         fpr_hosted_allreduce_fn allreduce, void* user);
int fpr_comm_finalize(fpr_ctx* ctx);   /* finalize_global_grid(); also done by fpr_ctx_destroy */
int fpr_comm_rank(fpr_ctx* ctx);
int fpr_comm_size(fpr_ctx* ctx);

/* init_global_grid(nx, ny, nz; dimx, dimy, dimz, periodx, periody, periodz) -> me, dims, nprocs, coords.
 * dim* = 0 lets the library factorise (MPI.Dims_create order: 2 -> (2,1,1), 4 -> (2,2,1), 8 -> (2,2,2), the table of
 * part1_scaling_experiments.jl:35-41).  Local arrays are nx*ny*nz including one halo cell on every side that has a
 * neighbour (overlap 2).  Ranks in MPI Cartesian order (last dimension fastest).  Works without a communicator
 * for the single-rank, non-periodic grid.  Out-pointers may be NULL; dims_out / coords_out: 3 ints. */
int fpr_grid_init(fpr_ctx* ctx, int nx, int ny, int nz, int dimx, int dimy, int dimz, int periodx, int periody,
                  int periodz, int* me_out, int* dims_out, int* nprocs_out, int* coords_out);
/* n_g_out[3] = nx_g(), ny_g(), nz_g() = dims*(n-2)+2 (dims*(n-2) in a periodic dimension); neighbors_out[6] = rank
 * beyond face 2*dim+side, or -1.  x_g(ix, dx, A) = (coords[0]*(nx-2) + ix-1)*dx follows from coords (host side). */
int fpr_grid_info(fpr_ctx* ctx, int* n_g_out, int* neighbors_out);

/* update_halo!(A): every halo plane of A that has a neighbour receives the neighbour's adjacent interior plane
 * (and vice versa), dimension by dimension as ImplicitGlobalGrid does, so edge / corner halo cells are consistent.
 * Enqueued on the context's streams (x / y planes: pack kernel -> ncclSend/ncclRecv -> unpack kernel; z planes in
 * place); the caller's next kernel on the compute stream is ordered behind it.  No host synchronisation. */
int fpr_halo_exchange3d(fpr_ctx* ctx, double* A, int nx, int ny, int nz);
/* Split form for overlap (role of @hide_communication (8,8,8), part1_kernel_programming.jl:185-188): begin packs on
 * the compute stream and posts ONE group of sends / receives on the comm stream; end makes the compute stream wait
 * for them and unpacks.  Kernels enqueued in between overlap the transfers.  face_mask: bit 2*dim+side (63 = all
 * faces).  All faces travel at once: edge / corner halo cells are not refreshed (a 7-point stencil reads none). */
int fpr_halo_exchange3d_begin(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask);
int fpr_halo_exchange3d_end(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask);
/* _comm: the whole exchange (pack, one group of sends / receives, unpack) on the COMM stream in stream order -- for a
 * chain of thin-box launches and exchanges kept on the comm stream beside one long launch on the compute stream
 * (fpr_diffusion3d_step2_core); order it against the compute stream with fpr_stream_wait.  Faces as for _begin/_end. */
int fpr_halo_exchange3d_comm(fpr_ctx* ctx, double* A, int nx, int ny, int nz, int face_mask);

/* MPI.Allreduce!(x, +, comm_cart) -- part1_utils.jl:38.  _dev: `count` device doubles in place on stream
 * stream_sel, no host sync (norms of several iterations can be reduced in one call).  All RCCL operations of a context
 * should run on ONE stream in the same order on every rank: the halo exchanges use the comm stream (1), so pass 1 here
 * as well and order it against the compute stream with fpr_stream_wait (grid.py _RcclTransport.allreduce_).  _sum1: the
 * reference's form, one host double in / out, comm stream, synchronises.  No-ops without a communicator. */
int fpr_allreduce_sum_dev(fpr_ctx* ctx, double* x_dev, int count, int stream_sel);
int fpr_allreduce_sum1(fpr_ctx* ctx, double* x_host_inout);

/* gather!(A, A_global): the local arrays, halos included, side by side in Cartesian order in the HOST array
 * A_global (nx*dims[0], ny*dims[1], nz*dims[2]) of rank 0 (NULL on the other ranks).  Synchronises. */
int fpr_gather3d(fpr_ctx* ctx, const double* A, int nx, int ny, int nz, double* A_global_host);

/* ---- Part 2: 2D geometric multigrid for (lap - c) u = f ------------------------------------------ */

/* B1: residual_2DPoisson!(u, f, h, c, res) -- multigrid.jl:173-188 (:191-220 shmem); wrapper :223-238 */
int fpr_residual2d(fpr_ctx* ctx, const double* u, const double* f, double h, double c, double* res, int nx, int ny);

/* B2: iteration_2DPoisson!(u, f, h, c, res, policy; alpha) -- multigrid.jl:245-258.
 * rms_host (nullable): sqrt(sum(res.^2)/(nx*ny)) measured before the update; non-NULL synchronises. */
int fpr_jacobi2d(fpr_ctx* ctx, double* u, const double* f, double h, double c, double* res, int nx, int ny,
                 double alpha, double* rms_host);

/* B3: restrict_wrapper!(fine, coarse, apply_BCs, policy) -- multigrid.jl:330-358; (nx,ny) = fine size */
int fpr_restrict2d(fpr_ctx* ctx, const double* fine, double* coarse, int nx, int ny, int apply_BCs);

/* B4: prolongate_wrapper!(coarse, fine, apply_BCs, policy) -- multigrid.jl:403-472; (nx,ny) = fine size.
 * Deterministic gather in the reference's sequential accumulation order (no atomics). */
int fpr_prolongate2d(fpr_ctx* ctx, const double* coarse, double* fine, int nx, int ny, int apply_BCs);

/* `u_f .= u_f - corr_f` -- multigrid.jl:139 */
int fpr_axmy2d(fpr_ctx* ctx, double* u, const double* corr, size_t n);

/* B5: matrix_free_matvec_prod!(T, hx, hy, c, dT2) -- krylov.jl:7-13 (:16-34 shmem); wrapper :37-52 */
int fpr_laplace_apply2d(fpr_ctx* ctx, const double* T, double hx, double hy, double c, double* dT2, int nx, int ny);

/* B6: apply_boundary_conditions{,_dirichlet,_neumann}! -- part2_utils.jl:22-39 */
int fpr_bc_dirichlet2d(fpr_ctx* ctx, double* T, int nx, int ny);
int fpr_bc_neumann2d(fpr_ctx* ctx, double* T, int nx, int ny);
int fpr_bc2d(fpr_ctx* ctx, double* T, int nx, int ny);

/* B7: Vcycle_2DPoisson!(u_f, rhs, h, c, tol, coarse_solve_size, coarse_solver, policy, apply_BCs)
 * -- multigrid.jl:91-170.  The level arena (prealloc_dict, :25-38) is owned by the context.
 * rms_host (nullable) = returned res_rms. */
int fpr_vcycle2d(fpr_ctx* ctx, double* u_f, const double* rhs, double h, double c, double tol,
                 int coarse_solve_size, int coarse_solver, int apply_BCs, int nx, int ny, double* rms_host);

/* B8: MGsolve_2DPoisson!(u, f, h, c, tol, niters, apply_BCs; opt) -- multigrid.jl:41-84.
 * rms_host = returned r_rms; ncycles_host = V-cycles executed; history_host (nullable, >= niters
 * doubles) = r_rms after each cycle; frms_host (nullable) = rms(f); converged_host (nullable) = 0 when
 * the reference would emit its @warn (:78-80) -- not an error.
 * The loop runs as one stream of launches where every launch of a cycle can honour a device-side stop flag (Jacobi coarse
 * solver on a hierarchy the marching passes take): the exit test :70 is evaluated on the device, cycles are enqueued
 * ahead of the host's view of the norm, and two consecutive cycles share their pass over the finest grid -- u, the
 * history, the cycle and coarse-iteration counts are those of the plain loop.  Tuning / A-B options
 * (fpr_set_option, defaults in brackets): mg_ahead [1] cycles enqueued ahead (0 = plain loop), mg_seam [1] shared pass
 * between cycles, mg_seam_predict [1], mg_mid [1] three launch-bound levels in two launches, mg_small_row [1], mg_zero_guess [1] coarse
 * levels do not read the zero guess their parent stored (:132), mg_seam_wg_per_cu [2] chunk height of the shared pass, mg_seam_history [1] the norms of the previous solve with the same (u, f, nx, ny) (relative to its threshold) supply the expected reduction from the last norm seen to cycle k for the guess which cycle will be the last (a guess changes which launches are enqueued, never a result). After an error return u is UNDEFINED (cycles enqueued ahead of the host are drained first, but may have run). */
int fpr_mgsolve2d(fpr_ctx* ctx, double* u, const double* f, double h, double c, double tol, int niters,
                  int apply_BCs, int coarse_solve_size, int coarse_solver, int nx, int ny, double* rms_host,
                  int* ncycles_host, double* history_host, double* frms_host, int* converged_host);

/* B9: cg!(x_in, b, hx, hy, c, tol, Nmax) -- krylov.jl:55-91 (starts from x = 0, overwrites x_in).
 * The dot products (:57, :64, :69, :72, :83, :90; the reference does not specify their summation order) are Dot2 sums: twofold
 * precision, rounded once -- x, the iteration count and the returned rms do not depend on the launch form (option cg_fused [3]:
 * 3 = one persistent launch of 64 workgroups of 256 threads where the grid fits, 2 / 1 / 0 = two / three / five launches per iteration)
 * and equal the CPU restatement's bit for bit.  On meshes with hx^2, hy^2 powers of two the persistent form multiplies by the exact
 * reciprocals instead of dividing (same bits).  Option handoff_fences [0]: 1 = its neighbour hand-offs carry agent-scope release /
 * acquire pairs (the form inside the HIP memory model: +12 % per iteration, profiles/r6_handoff_fences.txt). */
int fpr_cg2d(fpr_ctx* ctx, double* x_in, const double* b, double hx, double hy, double c, double tol, int Nmax,
             int nx, int ny, double* rms_host, int* iters_host);

/* prealloc_dict (multigrid.jl:25-38, 49-51: the reference lets the caller own the level buffers): the finest level's two
 * ping-pong partners of an (nx, ny) hierarchy, nx * ny doubles each, owned by the CALLER from now on (not freed by the library; they
 * must outlive every solve at this size or be withdrawn first).  NULL for one of them = a buffer of the library's own (again).  The
 * passes over the finest grid stream u, f and these two at equal offsets; on MI355X a host that places its field arrays
 *  places these two with them.  Waits for the context's streams; results do not depend on it. */
int fpr_mg_arena_provide(fpr_ctx* ctx, int nx, int ny, double* tmp, double* tmp2);
/* ... and the three arrays of the first coarse level that the passes over the finest grid stream beside them (injected residual and the
 * two alternating correction buffers; (1 + (nx-1)/2) x (1 + (ny-1)/2) doubles each, 16-byte aligned, distinct): all three or none
 * (NULL x 3 = the library's own again).  Contents need no initialisation; results do not depend on who owns the buffers. */
int fpr_mg_arena_provide_coarse(fpr_ctx* ctx, int nx, int ny, double* res_c, double* corr_c, double* corr_c2);

/* ---- placement of field arrays ------------------------------------------------------------
 * The role of the reference's `@zeros` allocations (part1_kernel_programming.jl:145-160, multigrid.jl:25-38 prealloc_dict): the HOST
 * owns its arrays; on MI355X which physical pages an allocation received decides how fast kernels run that stream several arrays at
 * equal offsets (fpr_diffusion3d_step2: 0.76 against 0.85-0.91 ms at 512^3).  fpr_placement_rank picks, among `k` candidate
 * allocations of `n` doubles each that the caller made (and keeps: nothing is allocated or freed here), the `count` the caller
 * should use for array positions 0 .. count-1: it times a copy between every pair of candidates, ranks the assignments by their
 * slowest streamed-together pair (`pairs`: 2 * npairs position indices; npairs = 0: every pair of positions), and -- when `trial`
 * is given -- times the caller's own kernel on the candidates as given (cand[0 .. count-1], what a host that simply allocates
 * would use), on the best-ranked assignments and through a short local search, and keeps the fastest.
 *   trial(user, chosen, count) -> ms of the caller's kernel on arrays cand[chosen[0]], ..., cand[chosen[count-1]]; <= 0: cannot judge
 *   chosen_out[count]: index into cand for every position;  report[FPR_PLACE_REPORT_LEN]: what was measured (indices below)
 * Contents of the candidates are overwritten by the copies (scratch during the search).  Measured constants: the candidates as given first,
 * then the 4 best-ranked assignments, then a local search in which a swap is kept when 0.5 % faster; report[FPR_PLACE_WANT_MORE] = 1 when
 * every pair of the pool copies below 5050 GB/s (a pool of one placement class), 2 when the trial sees less than 2.5 % between any two
 * assignments. */
typedef double (*fpr_place_trial_fn)(void* user, const int* chosen, int count);
enum {
    FPR_PLACE_POOL_FASTEST_GBS = 0,   /* fastest / median / slowest pair of the pool, GB/s (read + write)  */
    FPR_PLACE_POOL_MEDIAN_GBS = 1,
    FPR_PLACE_POOL_SLOWEST_GBS = 2,
    FPR_PLACE_CHOSEN_SLOWEST_GBS = 3, /* slowest / mean streamed-together pair of the assignment kept    */
    FPR_PLACE_CHOSEN_MEAN_GBS = 4,
    FPR_PLACE_TRIALS = 5,             /* trial() calls that returned a time                                */
    FPR_PLACE_TRIAL_BEST_MS = 6,      /* the assignment kept                                               */
    FPR_PLACE_TRIAL_FIRST_MS = 7,     /* the first trial (the candidates as given)                       */
    FPR_PLACE_TRIAL_WORST_MS = 8,
    FPR_PLACE_TRIAL_IDENTITY_MS = 9,  /* cand[0 .. count-1] as given: a host that simply allocates        */
    FPR_PLACE_TRIAL_SPREAD = 10,      /* worst / best - 1 over all trials                                  */
    FPR_PLACE_WANT_MORE = 11,         /* 0, 1 (pair copies uniform) or 2 (trials uniform)                  */
    FPR_PLACE_SEARCH_NODES = 12,      /* nodes of the assignment search; NEGATIVE when it ended at its cap     */
    FPR_PLACE_REPORT_LEN = 16
};
int fpr_placement_rank(fpr_ctx* ctx, double* const* cand, int k, size_t n, int count, const int* pairs, int npairs,
                       fpr_place_trial_fn trial, void* user, int* chosen_out, double* report);

/* coarse-solver iterations spent by the last fpr_vcycle2d / fpr_mgsolve2d call (diagnostics) */
long fpr_last_coarse_iters(fpr_ctx* ctx);

/* ---- NEXT: Navier-Stokes pointwise kernels, part2.jl:90-137 ------------------------- */
int fpr_compute_velocity2d(fpr_ctx* ctx, const double* S, double hx, double hy, double* vx, double* vy, int nx, int ny);
int fpr_compute_Ra_dTdx2d(fpr_ctx* ctx, double Ra, double hx, const double* T, double* out, int nx, int ny);
int fpr_compute_diffusion2d(fpr_ctx* ctx, const double* T, double hx, double hy, double k, double* dT2, int nx, int ny);
int fpr_compute_advection2d_x(fpr_ctx* ctx, const double* T, double hx, const double* vx, double* dTx, int nx, int ny);
int fpr_compute_advection2d_y(fpr_ctx* ctx, const double* T, double hy, const double* vy, double* dTy, int nx, int ny);
/* maximum(abs.(x)) -- part2.jl:77,82 */
int fpr_absmax(fpr_ctx* ctx, const double* x, size_t n, double* out_host);
/* The same step in two passes instead of 7 kernels + 3 maxima + 4 broadcasts (part2.jl:190-230), bit-identical:
 * pass 1: velocity from S (:190) + maximum(v), maximum(abs.(vx)), maximum(abs.(vy)) (:193-196 via :76-87) into
 * vmax_host[3]; vx / vy (nullable) are written only on request.  Synchronises like the reference's maximum().
 * pass 2: Ra dT/dx, diffusion and upwind advection terms of T and W (velocity recomputed from S) and, for beta > 0, the
 * right-hand sides of the two semi-implicit solves with c = 1/(beta dt) and c/Pr (:219-225), for beta == 0 the explicit
 * Euler update (:229-230), at every point of the arrays.  T must carry its boundary conditions (:199) already. */
int fpr_ns_velocity_max2d(fpr_ctx* ctx, const double* S, double hx, double hy, double* vx, double* vy, int nx, int ny,
                          double* vmax_host);
int fpr_ns_rhs2d(fpr_ctx* ctx, const double* T, const double* W, const double* S, double hx, double hy, int nx, int ny,
                 double Ra, double Pr, double k, double beta, double dt, double* T_out, double* W_out);
/* One time step of navier_stokes_2D (part2.jl:186-226) for beta > 0 as ONE call: S solve (:187, on ctx2: a second context on the
 * same device), pass 1, compute_dt (:76-87), boundary conditions of T (:199), pass 2, then the T solve (:221, on ctx) and the W solve
 * (:226, on ctx2) side by side -- they do not depend on each other and are bound by launch latency; the longer one of the previous
 * step on the calling thread, the other on a worker thread the context keeps.
 * The same launches with the same arguments as the piecewise calls (same T, W, S, dt bit for bit), without the host language
 * between them.  h = 1/(ny-1) as the driver (:163); dt_dif, a_adv as SimIn_t (:30-46); *dt_host receives the step's dt;
 * info_host (nullable, 6 ints): V-cycles of the S, T, W solves, then their converged flags (0 = the reference would @warn). */
int fpr_ns_step2d(fpr_ctx* ctx, fpr_ctx* ctx2, double* S, double* T, double* W, double* T_rhs, double* W_rhs, int nx, int ny,
                  double Ra, double Pr, double k, double beta, double a_adv, double dt_dif, double tol, int niters,
                  int coarse_solve_size, int coarse_solver, double* dt_host, int* info_host);
/* The driver's loop `while sim_time < ttot` (part2.jl:182) around that step for at most max_steps steps: *sim_time_inout advances by
 * every step's dt (:249), *steps_host = steps taken, *dt_host = the last step's dt, *unconverged_host (nullable) = number of solves
 * that did not converge (each one a @warn in the reference, :78-80).  The host language calls it in pieces where it has something
 * to do between steps (the reference starts its clock before the fourth step, :182-184; progress lines).
 * Inside a call the loop is software-pipelined (option ns_pipeline [1]): the S solve of step n+1 (:187) needs the W of step n and nothing of
 * its T, so it runs behind the W solve on ctx2 beside the T solve of step n; the S solve of a step that does not follow inside the call is
 * left to the next call -- the arrays a call returns are the reference's at the same point of its loop. */
int fpr_ns_run2d(fpr_ctx* ctx, fpr_ctx* ctx2, double* S, double* T, double* W, double* T_rhs, double* W_rhs, int nx, int ny,
                 double Ra, double Pr, double k, double beta, double a_adv, double dt_dif, double tol, int niters,
                 int coarse_solve_size, int coarse_solver, double ttot, int max_steps, double* sim_time_inout, int* steps_host,
                 double* dt_host, int* unconverged_host);

#ifdef __cplusplus
}
#endif
#endif /* FPR_H */
