"""Part 1 -- 3D pseudo-transient diffusion.  Host mirror of
scripts-part1/part1_kernel_programming.jl, part1_array_programming.jl and part1_utils.jl.

Kernel entry points keep the reference's argument lists; launch-geometry arguments of
`@parallel blocks threads shmem=...` have no counterpart (the library picks its own).
"""
import ctypes as C
import math
import time
from dataclasses import dataclass

from . import _lib
from ._lib import fptr, fzeros


def _ctx():
    from . import ctx

    return ctx()


@dataclass
class BenchResults:
    """part1_kernel_programming.jl:22-29"""

    Δt: float
    Work: float
    Performance: float
    Memory: float
    Intensity: float
    Throughput: float


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------
def diffusion_3D_step_τ(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz):
    """part1_kernel_programming.jl:46-58."""
    nx, ny, nz = Ht.shape
    _ctx().call("fpr_diffusion3d_step", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hτ2, 3), fptr(dHdτ, 3), nx, ny, nz,
                dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)


# part1_kernel_programming.jl:75-97 -- same arithmetic; on gfx950 both names run the same kernel
diffusion_3D_step_τ_shared_memory = diffusion_3D_step_τ


def diffusion_3D_step_τ_norm(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, sumsq_dev):
    """Fused update + local part of dist_norm_L2(residual_H*scale) (part1_kernel_programming.jl:191):
    sumsq_dev[0] (device) = sum((dHdτ*scale)^2) over the interior.  No host sync."""
    nx, ny, nz = Ht.shape
    _ctx().call("fpr_diffusion3d_step_norm", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hτ2, 3), fptr(dHdτ, 3), nx, ny, nz,
                dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, sumsq_dev.data_ptr())


def diffusion_3D_step_τ_norm_host(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale):
    """Fused update + norm returned to the host (one stream sync): sum((dHdτ*scale)^2) over the interior."""
    nx, ny, nz = Ht.shape
    out = C.c_double(0.0)
    _ctx().call("fpr_diffusion3d_step_norm_host", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hτ2, 3), fptr(dHdτ, 3), nx, ny, nz,
                dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, C.byref(out))
    return out.value


def diffusion_3D_step_τ_box(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi, scale=0.0,
                            sumsq_dev=None, stream_sel=0):
    """Sub-box form (role of @hide_communication, part1_kernel_programming.jl:185-188); 0-based [lo, hi)."""
    nx, ny, nz = Ht.shape
    lo3 = (C.c_int * 3)(*lo)
    hi3 = (C.c_int * 3)(*hi)
    _ctx().call("fpr_diffusion3d_step_box", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hτ2, 3), fptr(dHdτ, 3), nx, ny, nz,
                dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3, scale,
                sumsq_dev.data_ptr() if sumsq_dev is not None else None, stream_sel)


def can_step_τ2(Ht, Hτ, Hmid, Hout, dHdτ):
    """True if the fused two-iteration kernel serves these arrays (else use two single steps)."""
    nx, ny, nz = Ht.shape
    c = _ctx()
    return c.L.fpr_diffusion3d_can_step2(c.h, fptr(Ht, 3), fptr(Hτ, 3), fptr(Hmid, 3), fptr(Hout, 3), fptr(dHdτ, 3),
                                         nx, ny, nz) == 1


def diffusion_3D_step_τ2(Ht, Hτ, Hmid, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale=0.0, sumsq2_dev=None):
    """Two trips through the loop body of part1_kernel_programming.jl:179-192 in one pass over memory:
    step(Hτ -> Hmid); step(Hmid -> Hout) with Hmid never written (only its boundary cells are read).  Hout gets
    interior cells only and must already carry Hτ's boundary.  sumsq2_dev: 2 device doubles (norm sums of the
    first / second iteration), or None.  dHdτ may be None: the residual of the second iteration is then not stored."""
    nx, ny, nz = Ht.shape
    _ctx().call("fpr_diffusion3d_step2", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hmid, 3), fptr(Hout, 3),
                fptr(dHdτ, 3) if dHdτ is not None else None, nx, ny, nz,
                dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale,
                sumsq2_dev.data_ptr() if sumsq2_dev is not None else None)


def can_step_τ3(Ht, Hτ, Hout, dHdτ):
    """True if the fused three-iteration kernel serves these arrays (else pairs or single steps)."""
    nx, ny, nz = Ht.shape
    c = _ctx()
    return c.L.fpr_diffusion3d_can_step3(c.h, fptr(Ht, 3), fptr(Hτ, 3), fptr(Hout, 3), fptr(dHdτ, 3) if dHdτ is not None else None,
                                         nx, ny, nz) == 1


def diffusion_3D_step_τ3(Ht, Hτ, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale=0.0, sumsq3_dev=None):
    """Three trips through the loop body of part1_kernel_programming.jl:179-192 in one pass over memory: Hτ -> Hout, the two fields in
    between never written.  Hτ and Hout are the reference's two ping-pong buffers in either order (Hout carries the other buffer's
    boundary values and keeps them): no third work buffer.  sumsq3_dev: 3 device doubles (norm sums of the three iterations), or None;
    dHdτ may be None (the residual of the third iteration is then not stored)."""
    nx, ny, nz = Ht.shape
    _ctx().call("fpr_diffusion3d_step3", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hout, 3), fptr(dHdτ, 3) if dHdτ is not None else None,
                nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, sumsq3_dev.data_ptr() if sumsq3_dev is not None else None)


def diffusion_3D_step_τ2_box(Ht, Hτ, Hmid, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi, scale=0.0,
                             sumsq2_dev=None, stream_sel=0, z2=None):
    """Sub-box form of diffusion_3D_step_τ2 (0-based [lo, hi)); the two sums are accumulated into sumsq2_dev.
    z2 = (zlo2, zhi2): a second, disjoint z-range with the same x/y extent handled by the same launch."""
    nx, ny, nz = Ht.shape
    lo3 = (C.c_int * 3)(*lo)
    hi3 = (C.c_int * 3)(*hi)
    if z2 is not None:
        _ctx().call("fpr_diffusion3d_step2_box2", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hmid, 3), fptr(Hout, 3), fptr(dHdτ, 3),
                    nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3, int(z2[0]), int(z2[1]), scale,
                    sumsq2_dev.data_ptr() if sumsq2_dev is not None else None, stream_sel)
        return
    _ctx().call("fpr_diffusion3d_step2_box", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hmid, 3), fptr(Hout, 3), fptr(dHdτ, 3),
                nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3, scale,
                sumsq2_dev.data_ptr() if sumsq2_dev is not None else None, stream_sel)


def diffusion_3D_step_τ2_core(Ht, Hτ, Hmid, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi, scale=0.0,
                              sumsq2_dev=None, stream_sel=0, reserve_cus=0, accumulate=True):
    """The core box of a decomposed run's fused pair, leaving `reserve_cus` compute units to the shell launches and the
    halo exchanges that run beside it on the comm stream (fpr_diffusion3d_step2_core; role of @hide_communication,
    part1_kernel_programming.jl:185-188).  Results as diffusion_3D_step_τ2_box; accumulate=False writes the two sums."""
    nx, ny, nz = Ht.shape
    lo3 = (C.c_int * 3)(*lo)
    hi3 = (C.c_int * 3)(*hi)
    _ctx().call("fpr_diffusion3d_step2_core", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hmid, 3), fptr(Hout, 3), fptr(dHdτ, 3),
                nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3, scale,
                sumsq2_dev.data_ptr() if sumsq2_dev is not None else None, stream_sel, int(reserve_cus), int(bool(accumulate)))


def compute_flux_(qx, qy, qz, Hτ, D, dx, dy, dz):
    """part1_array_programming.jl:10-12 (north_star: compute_flux!)."""
    nx, ny, nz = Hτ.shape
    _ctx().call("fpr_diffusion3d_flux", fptr(qx, 3), fptr(qy, 3), fptr(qz, 3), fptr(Hτ, 3), nx, ny, nz, D, dx, dy, dz)


def compute_dHdτ_(dHdτ, Hτ, Ht, qx, qy, qz, dt, dx, dy, dz):
    """part1_array_programming.jl:14-15 (north_star: compute_dHdτ!)."""
    nx, ny, nz = Hτ.shape
    _ctx().call("fpr_diffusion3d_dHdtau", fptr(dHdτ, 3), fptr(Hτ, 3), fptr(Ht, 3), fptr(qx, 3), fptr(qy, 3), fptr(qz, 3),
                nx, ny, nz, dt, dx, dy, dz)


def update_H_(Hτ, dHdτ, dτ):
    """part1_array_programming.jl:16 (north_star: update_H!)."""
    nx, ny, nz = Hτ.shape
    _ctx().call("fpr_diffusion3d_update", fptr(Hτ, 3), fptr(dHdτ, 3), nx, ny, nz, dτ)


def diffusion_3D_step_τ_(Ht, Hτ, dHdτ, dt, dτ, qx, qy, qz, dx, dy, dz, D):
    """part1_array_programming.jl:9-18 with clean (barrier-separated) semantics, see SURVEY 8a-A3."""
    compute_flux_(qx, qy, qz, Hτ, D, dx, dy, dz)
    compute_dHdτ_(dHdτ, Hτ, Ht, qx, qy, qz, dt, dx, dy, dz)
    update_H_(Hτ, dHdτ, dτ)


# ------------------------------------------------------------------------------------------------
# part1_utils.jl
# ------------------------------------------------------------------------------------------------
def init_local_gaussian(center, dx, dy, dz, H, coords=(0, 0, 0)):
    """part1_utils.jl:1-12; fills and returns H."""
    nx, ny, nz = H.shape
    _ctx().call("fpr_init_gaussian3d", fptr(H, 3), nx, ny, nz, dx, dy, dz, center[0], center[1], center[2],
                coords[0], coords[1], coords[2])
    return H


def apply_boundary_conditions_(H, coords, dims):
    """part1_utils.jl:14-34 -- including the reference's 0-based-coords-vs-1 comparison."""
    if coords[0] == 1:
        H[0, :, :] = 0.0
    if coords[1] == 1:
        H[:, 0, :] = 0.0
    if coords[2] == 1:
        H[:, :, 0] = 0.0
    if coords[0] == dims[0]:
        H[-1, :, :] = 0.0
    if coords[1] == dims[1]:
        H[:, -1, :] = 0.0
    if coords[2] == dims[2]:
        H[:, :, -1] = 0.0


def local_sumsq(Rh, scale=1.0):
    """sum((Rh*scale).^2) -- part1_utils.jl:37 (host value; synchronises)."""
    out = C.c_double(0.0)
    _ctx().call("fpr_sumsq_scaled", fptr(Rh), Rh.numel(), scale, C.byref(out))
    return out.value


def dist_norm_L2(Rh, comm_cart=None, scale=1.0):
    """part1_utils.jl:36-40: sqrt(allreduce_sum(sum((Rh*scale)^2)))."""
    s = local_sumsq(Rh, scale)
    if comm_cart is not None:
        s = comm_cart.allreduce_sum(s)
    return math.sqrt(s)


LOCATION_OF_INTEREST = (4.5, 4.5, 4.5)  # part1_error_vs_*_experiments.jl:20


def linear_interpolate_3D(H, dx, location=LOCATION_OF_INTEREST):
    """part1_utils.jl:42-71 (NEXT row 8f-4).  H: host array (numpy, Julia index order).

    The reference builds an 8x8 trilinear system whose z column holds z0 eight times (:57), so the matrix is
    singular by construction, `M \\ cvec` throws, and the `catch` branch (:67-69) returns H[ix, iy, iz] with
    ix = Int(x ÷ dx) + 1.  The published `interp_val` columns confirm this (tests/test_oracle_pins.py).  The
    system is still assembled here so that the behaviour follows from the same data."""
    import numpy as np

    dy = dz = dx
    ix, iy, iz = (int(c // dx) + 1 for c in location)  # 1-based, :45
    x0, x1 = ix * dx + dx / 2, (ix + 1) * dx + dx / 2
    y0, y1 = iy * dy + dy / 2, (iy + 1) * dy + dy / 2
    z0 = iz * dz + dz / 2
    h = lambda a, b, c: float(H[a - 1, b - 1, c - 1])
    cvec = np.array([h(ix, iy, iz), h(ix + 1, iy, iz), h(ix, iy + 1, iz), h(ix + 1, iy + 1, iz), h(ix, iy, iz + 1),
                     h(ix + 1, iy, iz + 1), h(ix, iy + 1, iz + 1), h(ix + 1, iy + 1, iz + 1)])
    xvec = np.tile([x0, x1], 4)
    yvec = np.tile([y0, y0, y1, y1], 2)
    zvec = np.full(8, z0)  # :57 -- z1 is never used
    M = np.column_stack([np.ones(8), xvec, yvec, zvec, xvec * yvec, xvec * zvec, yvec * zvec, xvec * yvec * zvec])
    if np.linalg.matrix_rank(M) < 8:  # always true: the z columns are multiples of the others
        return cvec[0]
    avec = np.linalg.solve(M, cvec)
    x, y, z = location
    return float(np.array([1, x, y, z, x * y, x * z, y * z, x * z * z]) @ avec)  # :64 (sic: x*z*z)


def probe_value(H, X):
    """`val` of part1_error_vs_*_experiments.jl:31-37: H[ix,iy,iz] with ix = round(Int, 4.5/dx + 1), dx = X[2]-X[1]."""
    import numpy as np

    dx = X[1] - X[0]
    i, j, k = (int(np.round(c / dx + 1)) - 1 for c in LOCATION_OF_INTEREST)
    return float(H[i, j, k]), dx


# ------------------------------------------------------------------------------------------------
# solver host loop
# ------------------------------------------------------------------------------------------------
def diffusion_3D_kernel_programming(*, nx, ny, nz, ttot=1.0, tol=1e-8, use_shared_memory=True, verbose=False,
                                    scale_physical_size=False, iter_max=100000, fixed_iters=0, check_every=1,
                                    global_grid=None, Ht_init=None, return_device=False, native_loop=True):
    """part1_kernel_programming.jl:99-228.

    Extra keyword arguments (SURVEY section 5 "config / flags"):
      iter_max     -- the reference hard-codes 1e5 (:130)
      fixed_iters  -- run exactly this many pseudo-iterations per physical step (benchmark mode)
      check_every  -- evaluate the convergence norm on the host every n-th iteration (1 = reference)
      global_grid  -- grid.GlobalGrid for multi-GPU runs (None = single rank)
      Ht_init      -- optional initial condition (device array) instead of the Gaussian
      native_loop  -- single rank: run the host loop in libfpr_hip.so (fpr_diffusion3d_solve) instead of Python
    Returns (X_g, H, BenchResults, info) -- H is the local field (numpy, or device tensor).
    """
    import torch
    from . import grid as _grid

    ctx = _ctx()
    gg = global_grid if global_grid is not None else _grid.GlobalGrid(nx, ny, nz, dims=(1, 1, 1))
    me, dims, coords = gg.me, gg.dims, gg.coords
    D = 1.0
    if scale_physical_size:  # :110-114
        lx, ly, lz = (d * 10.0 for d in dims)
    else:
        lx, ly, lz = 10.0, 10.0, 10.0
    dx, dy, dz = lx / gg.nx_g(), ly / gg.ny_g(), lz / gg.nz_g()  # :117
    total_N = dims[0] * dims[1] * dims[2] * nx * ny * nz  # :124
    dt = 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1  # :128
    nt = int(round(ttot / dt)) if abs(ttot / dt - round(ttot / dt)) < 1e-9 else int(math.ceil(ttot / dt))
    center = [lx / 2, ly / 2, lz / 2]
    Ht = fzeros(nx, ny, nz)
    if Ht_init is not None:
        Ht.copy_(Ht_init)
    else:
        init_local_gaussian(center, dx, dy, dz, Ht, coords)  # :137-138
    if gg.nprocs == 1:
        apply_boundary_conditions_(Ht, coords, dims)  # :139 -- a no-op on one rank (0-based coords vs 1, part1_utils.jl:14-34)
    # On several ranks the same comparison would zero an INTERNAL halo plane of the rank with coordinate 1 (and
    # never the physical boundary); it is skipped so that an N-shard run equals the single-domain run (EXPERIMENTS.md section 5).
    Hτ = Ht.clone(memory_format=torch.preserve_format)  # :140
    Hτ2 = fzeros(nx, ny, nz)  # :141
    residual_H = fzeros(nx, ny, nz)  # :142
    _dt, _dx, _dy, _dz = 1.0 / dt, 1.0 / dx, 1.0 / dy, 1.0 / dz  # :146-149
    D_dx, D_dy, D_dz = D / dx, D / dy, D / dz  # :150-152
    sq = ctx.scal[:1]
    if native_loop and gg.nprocs == 1:
        # the whole host loop (:166-204) runs inside libfpr_hip.so; the timing rule of :170-176 (timer starts at
        # the 4th physical step) is reproduced by splitting the call.  The third work buffer of the fused pairs is
        # allocated HERE (field arrays belong to the host language); None = one iteration per launch.
        Hτ3 = fzeros(nx, ny, nz)
        if not can_step_τ2(Ht, Hτ, Hτ2, Hτ3, residual_H):
            Hτ3 = None

        def run_steps(n_steps):
            nonlocal Hτ, Hτ2
            its = (C.c_long * max(n_steps, 1))()
            errs = (C.c_double * max(n_steps, 1))()
            sw = C.c_int(0)
            ctx.call("fpr_diffusion3d_solve", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hτ2, 3),
                     fptr(Hτ3, 3) if Hτ3 is not None else None, fptr(residual_H, 3), nx, ny, nz,
                     dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, dt, float(total_N), n_steps, tol, int(iter_max),
                     int(fixed_iters), int(check_every), its, errs, C.byref(sw))
            if sw.value:
                Hτ, Hτ2 = Hτ2, Hτ
            return list(its)[:n_steps], list(errs)[:n_steps]

        warm = min(3, nt)
        it_w, er_w = run_steps(warm)
        ctx.synchronize()
        tic = time.time()
        it_t, er_t = run_steps(nt - warm)
        ctx.synchronize()
        Δt = time.time() - tic
        iters_per_step, err_per_step = it_w + it_t, er_w + er_t
        timed_iter_total = sum(it_t) if nt > 3 else sum(it_w)  # the reference resets its counter at step 4 only (:170-176)
        return _finish(gg, nx, ny, nz, dims, dx, lx, Δt, timed_iter_total, use_shared_memory, iters_per_step, err_per_step,
                       dτ, Ht, Hτ, residual_H, return_device)
    iter_outer = 0
    timed_iter_total = 0
    tic = time.time()
    iters_per_step, err_per_step = [], []
    sqrtN = math.sqrt(total_N)
    # Fused pairs (two iterations per launch, GlobalGrid.step2): the field alternates between Hτ and a third buffer Hτ3
    # that carries Hτ's boundary; Hτ2 keeps the role of the reference's second work buffer.  Pairs start from an "even"
    # state only (field in Hτ / Hτ3); single steps handle the odd state (field in Hτ2) and odd iteration counts.  An
    # iteration whose norm ends the loop is replayed alone, so results are those of the plain loop.
    # Fused triples between ranks (z-slab decompositions, GlobalGrid.step3): Hτ and Hτ2 alternate as in the reference, three iterations at
    # a time; a norm inside a triple that ends the loop has the iterations up to it replayed singly from the (intact) input.
    coefs = (dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
    fuse3 = gg.nprocs > 1 and gg.can_step3(Ht, Hτ, Hτ2, residual_H)
    sq3 = ctx.scal[:3]
    for _ in range(nt if fuse3 else 0):  # :166, triples
        if iter_outer == 3:  # manual warm-up, :170-176
            ctx.synchronize()
            tic = time.time()
            timed_iter_total = 0
        iter_inner = 0
        err = 2 * tol
        while (iter_inner < fixed_iters) if fixed_iters > 0 else (err > tol and iter_inner < iter_max):  # :179
            left = (fixed_iters if fixed_iters > 0 else iter_max) - iter_inner
            if left >= 3:
                wanted = [(j % check_every == 0) if fixed_iters == 0 else (j == fixed_iters) for j in range(iter_inner + 1, iter_inner + 4)]
                gg.step3(Ht, Hτ, Hτ2, residual_H, *coefs, dt, sq3 if any(wanted) else None)
                stop_at = None
                if any(wanted):
                    s3 = [float(v) for v in gg.allreduce_(sq3).tolist()]
                    for j in range(3):
                        if wanted[j]:
                            err = math.sqrt(s3[j]) / sqrtN  # :191
                            if fixed_iters == 0 and not err > tol:
                                stop_at = j
                                break
                if stop_at is not None and stop_at < 2:   # the reference stops inside the triple: replay up to there
                    for _j in range(stop_at + 1):
                        gg.step(Ht, Hτ, Hτ2, residual_H, *coefs, dt, None)
                        Hτ, Hτ2 = Hτ2, Hτ  # :190
                    iter_inner += stop_at + 1
                    continue
                Hτ, Hτ2 = Hτ2, Hτ
                iter_inner += 3
                continue
            need_norm = (iter_inner + 1) % check_every == 0 if fixed_iters == 0 else iter_inner + 1 == fixed_iters
            gg.step(Ht, Hτ, Hτ2, residual_H, *coefs, dt, sq if need_norm else None)
            Hτ, Hτ2 = Hτ2, Hτ  # :190
            if need_norm:
                err = math.sqrt(gg.allreduce_sum(sq)) / sqrtN  # :191
            iter_inner += 1
        if verbose and me == 0:
            print("Converged after %d iterations." % iter_inner if err <= tol else
                  "Couldn't converge within %d iterations." % iter_inner)
        iters_per_step.append(iter_inner)
        err_per_step.append(err)
        timed_iter_total += iter_inner
        iter_outer += 1
        gg.join()
        ctx.call("fpr_copy", fptr(Ht), fptr(Hτ), Ht.numel())  # Ht .= Hτ, :203
    if fuse3:
        ctx.synchronize()
        Δt = time.time() - tic
        return _finish(gg, nx, ny, nz, dims, dx, lx, Δt, timed_iter_total, use_shared_memory, iters_per_step, err_per_step,
                       dτ, Ht, Hτ, residual_H, return_device)
    Hτ3 = Hτ.clone(memory_format=torch.preserve_format)
    fuse = gg.can_step2(Ht, Hτ, Hτ2, Hτ3, residual_H)   # false as well when option diff3_fuse2 is 0
    if not fuse:
        Hτ3 = None
    sq2 = ctx.scal[:2]
    cur, parity = Hτ, 0

    def want_norm(j):
        return (j % check_every == 0) if fixed_iters == 0 else (j == fixed_iters)

    def reduce2(t):
        if gg.dist is not None and gg.nprocs > 1:
            gg.dist.all_reduce(t, op=gg.dist.ReduceOp.SUM, group=gg.group)
        return [float(v) for v in t.tolist()]

    for _ in range(nt if fuse else 0):  # :166, fused form
        if iter_outer == 3:  # manual warm-up, :170-176
            ctx.synchronize()
            tic = time.time()
            timed_iter_total = 0
        iter_inner = 0
        err = 2 * tol
        while (iter_inner < fixed_iters) if fixed_iters > 0 else (err > tol and iter_inner < iter_max):  # :179
            left = (fixed_iters if fixed_iters > 0 else iter_max) - iter_inner
            if parity == 0 and left >= 2:
                out = Hτ3 if cur is Hτ else Hτ
                n1, n2 = want_norm(iter_inner + 1), want_norm(iter_inner + 2)
                gg.step2(Ht, cur, Hτ2, out, residual_H, *coefs, dt, sq2 if (n1 or n2) else None)
                if n1 or n2:
                    s1, s2 = reduce2(sq2)
                if n1 and fixed_iters == 0:
                    err = math.sqrt(s1) / sqrtN  # :191 after the first iteration of the pair
                    if not err > tol:
                        # the reference stops here: replay that one iteration from the (intact) input
                        gg.step(Ht, cur, Hτ2, residual_H, *coefs, dt, None)
                        cur, parity = Hτ2, 1
                        iter_inner += 1
                        continue
                cur = out
                iter_inner += 2
                if n2:
                    err = math.sqrt(s2) / sqrtN
                continue
            out = Hτ if parity else Hτ2          # single iteration: odd -> even (into Hτ) or even -> odd
            need_norm = want_norm(iter_inner + 1)
            gg.step(Ht, cur, out, residual_H, *coefs, dt, sq if need_norm else None)
            cur, parity = out, parity ^ 1  # :190
            if need_norm:
                err = math.sqrt(gg.allreduce_sum(sq)) / sqrtN  # :191
            iter_inner += 1
        if verbose and me == 0:
            print("Converged after %d iterations." % iter_inner if err <= tol else
                  "Couldn't converge within %d iterations." % iter_inner)
        iters_per_step.append(iter_inner)
        err_per_step.append(err)
        timed_iter_total += iter_inner
        iter_outer += 1
        ctx.call("fpr_copy", fptr(Ht), fptr(cur), Ht.numel())  # Ht .= Hτ, :203
    if fuse:
        ctx.synchronize()
        Δt = time.time() - tic
        return _finish(gg, nx, ny, nz, dims, dx, lx, Δt, timed_iter_total, use_shared_memory, iters_per_step, err_per_step,
                       dτ, Ht, cur, residual_H, return_device)
    for _ in range(nt):  # :166
        if iter_outer == 3:  # manual warm-up, :170-176
            ctx.synchronize()
            tic = time.time()
            timed_iter_total = 0
        iter_inner = 0
        err = 2 * tol
        while (iter_inner < fixed_iters) if fixed_iters > 0 else (err > tol and iter_inner < iter_max):  # :179
            need_norm = (fixed_iters == 0 and (iter_inner + 1) % check_every == 0) or \
                        (fixed_iters > 0 and iter_inner + 1 == fixed_iters)
            if need_norm and gg.nprocs == 1:
                # single rank: kernel + reduction + one stream sync, the sum lands in pinned host memory
                s_loc = diffusion_3D_step_τ_norm_host(Ht, Hτ, Hτ2, residual_H, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, dt)
                Hτ, Hτ2 = Hτ2, Hτ  # :190
                err = math.sqrt(s_loc) / sqrtN  # :191
                iter_inner += 1
                continue
            gg.step(Ht, Hτ, Hτ2, residual_H, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, dt, sq if need_norm else None)
            Hτ, Hτ2 = Hτ2, Hτ  # :190
            if need_norm:
                err = math.sqrt(gg.allreduce_sum(sq)) / sqrtN  # :191
            iter_inner += 1
        if verbose and me == 0:
            print("Converged after %d iterations." % iter_inner if err <= tol else
                  "Couldn't converge within %d iterations." % iter_inner)
        iters_per_step.append(iter_inner)
        err_per_step.append(err)
        timed_iter_total += iter_inner
        iter_outer += 1
        ctx.call("fpr_copy", fptr(Ht), fptr(Hτ), Ht.numel())  # Ht .= Hτ, :203
    ctx.synchronize()
    Δt = time.time() - tic
    return _finish(gg, nx, ny, nz, dims, dx, lx, Δt, timed_iter_total, use_shared_memory, iters_per_step, err_per_step,
                   dτ, Ht, Hτ, residual_H, return_device)


def diffusion_3D_array_programming(*, nx, ny, nz, do_vis=False, verbose=True, ttot=1.0, tol=1e-8, iter_max=100000,
                                   fixed_iters=0, Ht_init=None, return_info=False):
    """part1_array_programming.jl:20-92 (BASELINE config 1) on the split kernels compute_flux! / compute_dHdτ! /
    update_H! (clean semantics of the reference's single array kernel :9-18).  Single rank: the
    reference's update_halo!(Hτ) (:67) has no neighbour to talk to there, and its multi-rank runs use the
    kernel-programming solver (part1_scaling_experiments.jl).  One host round trip per pseudo-iteration for the
    norm, as in the reference (:68).  Returns (X_g, H_g) like the reference; with return_info also a dict.

    Extra keyword arguments: ttot / tol / iter_max are hard-coded in the reference (:24,43-44); fixed_iters runs
    exactly that many pseudo-iterations per physical step (config 1: 50); Ht_init replaces the Gaussian."""
    import numpy as np
    import torch

    lx, ly, lz = 10.0, 10.0, 10.0  # :22
    D = 1.0  # :23
    dx, dy, dz = lx / nx, ly / ny, lz / nz  # :30, nx_g() == nx on one rank
    total_N = nx * ny * nz  # :37
    dt = 0.2  # :40
    dτ = min(dx, dy, dz) ** 2 / D / 8.1  # :41
    center = [lx / 2, ly / 2, lz / 2]  # :47
    qx = fzeros(nx - 1, ny - 2, nz - 2)  # :48-50
    qy = fzeros(nx - 2, ny - 1, nz - 2)
    qz = fzeros(nx - 2, ny - 2, nz - 1)
    Ht = fzeros(nx, ny, nz)  # :51
    if Ht_init is not None:
        Ht.copy_(Ht_init)
    else:
        init_local_gaussian(center, dx, dy, dz, Ht)  # :52
    apply_boundary_conditions_(Ht, (0, 0, 0), (1, 1, 1))  # :53 -- a no-op on one rank (part1_utils.jl:14-34)
    Hτ = Ht.clone(memory_format=torch.preserve_format)  # :54
    dHdt = fzeros(nx - 2, ny - 2, nz - 2)  # :55
    sqrtN = math.sqrt(total_N)
    t = 0.0
    iter_total = 0
    iters_per_step, err_per_step = [], []
    while t < ttot:  # :61
        iter_inner = 0
        err = 2 * tol  # :64
        while (iter_inner < fixed_iters) if fixed_iters > 0 else (err > tol and iter_inner < iter_max):  # :65
            diffusion_3D_step_τ_(Ht, Hτ, dHdt, dt, dτ, qx, qy, qz, dx, dy, dz, D)  # :66
            err = math.sqrt(local_sumsq(dHdt, dt)) / sqrtN  # :68  dist_norm_L2(dHdt * dt, comm_cart) / sqrt(total_N)
            iter_inner += 1
        if verbose:  # :71-77
            print("Converged after %d iterations." % iter_inner if err <= tol else
                  "Couldn't converge within %d iterations." % iter_max)
        iters_per_step.append(iter_inner)
        err_per_step.append(err)
        iter_total += iter_inner
        t += dt  # :80
        _ctx().call("fpr_copy", fptr(Ht), fptr(Hτ), Ht.numel())  # Ht .= Hτ, :81
    X_g = np.linspace(dx / 2, lx - dx / 2, nx)  # :85
    H_g = _lib.tonumpy(Ht)  # gather!(Array(Ht), H_g), :87
    if return_info:
        return X_g, H_g, {"iters": iters_per_step, "err": err_per_step, "dHdt": dHdt, "Hτ": Hτ, "dτ": dτ, "dx": dx}
    return X_g, H_g


def _finish(gg, nx, ny, nz, dims, dx, lx, Δt, timed_iter_total, use_shared_memory, iters_per_step, err_per_step, dτ,
            Ht, Hτ, residual_H, return_device):
    """Benchmark accounting and return values of diffusion_3D_kernel_programming (:209-227)."""
    nranks = gg.nprocs
    cells = (nx - 2) * (ny - 2) * (nz - 2)
    Work = nranks * timed_iter_total * (25 + 2) * cells  # :210
    Memory = nranks * timed_iter_total * ((6 + 1) if use_shared_memory else (14 + 1)) * 8 * cells  # :212-214
    bench = BenchResults(Δt, Work, Work / Δt if Δt > 0 else 0.0, Memory, Work / Memory if Memory else 0.0,
                         Memory / Δt if Δt > 0 else 0.0)
    import numpy as np

    X_g = np.linspace(dx / 2, lx - dx / 2, nx * dims[0])  # :221
    info = {"iters": iters_per_step, "err": err_per_step, "dx": dx, "dτ": dτ, "A_eff_GBs":
            (32.0 * cells * nranks * timed_iter_total / Δt / 1e9 if Δt > 0 else 0.0),
            "Hτ": Hτ, "residual_H": residual_H}
    H = Ht if return_device else _lib.tonumpy(Ht)
    return X_g, H, bench, info
