"""Part 2 -- 2D geometric multigrid for (∇² - c) u = f.  Host mirror of scripts-part2/multigrid.jl,
krylov.jl and part2_utils.jl on top of libfpr_hip.so.  Julia `f!` is spelled `f_`.

All arrays are column-major float64 device tensors of Julia shape (nx, ny) (fzeros / asdevice).
"""
import ctypes as C
import enum
import struct
import warnings

import numpy as np

from ._lib import FPR_COARSE_CG, FPR_COARSE_JACOBI, FprError, fptr


def _ctx():
    from . import ctx

    return ctx()


class ExecutionPolicy_t(enum.Enum):
    """part2_utils.jl:4-8.  On gfx950 `parallel` and `parallel_shmem` run the same kernels; `serial`
    is the reference's unfinished debug branch (multigrid.jl:233-234 error())."""

    serial = 0
    parallel = 1
    parallel_shmem = 2


serial, parallel, parallel_shmem = ExecutionPolicy_t.serial, ExecutionPolicy_t.parallel, ExecutionPolicy_t.parallel_shmem


class CoarseSolver_t(enum.Enum):
    """multigrid.jl:10-13"""

    jacobi = FPR_COARSE_JACOBI
    conjugate_gradient = FPR_COARSE_CG


jacobi, conjugate_gradient = CoarseSolver_t.jacobi, CoarseSolver_t.conjugate_gradient


class MGOpt:
    """multigrid.jl:16-22 (defaults: coarse 5, jacobi, parallel_shmem)."""

    def __init__(self):
        self.coarse_solve_size = 5
        self.coarse_solver = jacobi
        self.execution_policy = parallel_shmem


def preallocate_buffers(nx, ny):
    """multigrid.jl:25-38.  The level arena is owned by the library context (allocated on first use
    for a given finest size); the returned token only keeps the reference's call signature."""
    return {"nx": nx, "ny": ny}


def provide_arena_(nx, ny, tmp=None, tmp2=None):
    """fpr_mg_arena_provide: the finest level's two ping-pong partners of an (nx, ny) hierarchy from the caller (column-major
    float64 device arrays of that shape; None = the library's own again).  The context keeps them alive.  Used to place them
    together with u and f (placement.alloc_fields); results do not depend on it."""
    c = _ctx()
    c.call("fpr_mg_arena_provide", int(nx), int(ny), fptr(tmp, 2) if tmp is not None else None, fptr(tmp2, 2) if tmp2 is not None else None)
    if not hasattr(c, "_arena_refs"):
        c._arena_refs = {}
    c._arena_refs[(int(nx), int(ny))] = (tmp, tmp2)


def provide_arena_coarse_(nx, ny, res_c=None, corr_c=None, corr_c2=None):
    """fpr_mg_arena_provide_coarse: the three arrays of the first coarse level of an (nx, ny) hierarchy from the caller (injected
    residual and the two alternating correction buffers, shape (1 + (nx-1)//2, 1 + (ny-1)//2); all three or none).  The passes over
    the finest grid stream them beside u, f and the ping-pong partners; results do not depend on who owns them."""
    c = _ctx()
    arrs = (res_c, corr_c, corr_c2)
    c.call("fpr_mg_arena_provide_coarse", int(nx), int(ny), *(fptr(a, 2) if a is not None else None for a in arrs))
    if not hasattr(c, "_arena_refs_coarse"):
        c._arena_refs_coarse = {}
    c._arena_refs_coarse[(int(nx), int(ny))] = arrs


def _policy(p):
    if p == serial or p not in (parallel, parallel_shmem):
        raise RuntimeError("execution policy %r not implemented (reference: error(), multigrid.jl:233-236)" % (p,))


# ---- part2_utils.jl -----------------------------------------------------------------------------
def load(path):
    """part2_utils.jl:11-19: Int32 nx, Int32 ny, nx*ny Float64 column-major -> numpy (Fortran order)."""
    with open(path, "rb") as fh:
        nx, ny = struct.unpack("<ii", fh.read(8))
        a = np.frombuffer(fh.read(8 * nx * ny), dtype="<f8").reshape((nx, ny), order="F")
    return np.asfortranarray(a, dtype=np.float64)


def apply_boundary_conditions_(T):
    """part2_utils.jl:22-25"""
    _ctx().call("fpr_bc2d", fptr(T, 2), *T.shape)


def apply_boundary_conditions_dirichlet_(T):
    """part2_utils.jl:28-32"""
    _ctx().call("fpr_bc_dirichlet2d", fptr(T, 2), *T.shape)


def apply_boundary_conditions_neumann_(T):
    """part2_utils.jl:35-39"""
    _ctx().call("fpr_bc_neumann2d", fptr(T, 2), *T.shape)


# ---- kernels / wrappers ----------------------------------------------------------------------------
def residual_2DPoisson_(u, f, h, c, res):
    """multigrid.jl:173-188"""
    _ctx().call("fpr_residual2d", fptr(u, 2), fptr(f, 2), h, c, fptr(res, 2), *u.shape)


residual_2DPoisson_shmem_ = residual_2DPoisson_  # multigrid.jl:191-220, same arithmetic


def residual_2DPoisson_wrapper_(u_f, rhs, h, c, res_f, execution_policy=parallel_shmem):
    """multigrid.jl:223-238"""
    _policy(execution_policy)
    residual_2DPoisson_(u_f, rhs, h, c, res_f)


def iteration_2DPoisson_(u, f, h, c, res, execution_policy=parallel_shmem, alpha=4.0 / 5.0, want_rms=True):
    """multigrid.jl:245-258; returns r_rms (measured before the update) -- a host sync, as in the reference."""
    _policy(execution_policy)
    out = C.c_double(0.0)
    _ctx().call("fpr_jacobi2d", fptr(u, 2), fptr(f, 2), h, c, fptr(res, 2), *u.shape, alpha,
                C.byref(out) if want_rms else None)
    return out.value


def restrict_wrapper_(fine, coarse, apply_BCs, execution_policy=parallel_shmem):
    """multigrid.jl:344-358"""
    _policy(execution_policy)
    nxc, nyc = 1 + (fine.shape[0] - 1) // 2, 1 + (fine.shape[1] - 1) // 2
    if tuple(coarse.shape) != (nxc, nyc):
        raise ValueError("coarse array must be %dx%d" % (nxc, nyc))
    _ctx().call("fpr_restrict2d", fptr(fine, 2), fptr(coarse, 2), *fine.shape, int(bool(apply_BCs)))


def prolongate_wrapper_(coarse, fine, apply_BCs, execution_policy=parallel_shmem):
    """multigrid.jl:451-472 (deterministic gather instead of atomics)"""
    _policy(execution_policy)
    nxc, nyc = 1 + (fine.shape[0] - 1) // 2, 1 + (fine.shape[1] - 1) // 2
    if tuple(coarse.shape) != (nxc, nyc):
        raise ValueError("coarse array must be %dx%d" % (nxc, nyc))
    _ctx().call("fpr_prolongate2d", fptr(coarse, 2), fptr(fine, 2), *fine.shape, int(bool(apply_BCs)))


def correct_(u_f, corr_f):
    """`u_f .= u_f - corr_f` (multigrid.jl:139)"""
    _ctx().call("fpr_axmy2d", fptr(u_f), fptr(corr_f), u_f.numel())


def matrix_free_matvec_prod_(T, hx, hy, c, dT2):
    """krylov.jl:7-13"""
    _ctx().call("fpr_laplace_apply2d", fptr(T, 2), hx, hy, c, fptr(dT2, 2), *T.shape)


def matrix_free_matvec_prod_wrapper_(p, hx, hy, c, p_hat, execution_policy=parallel_shmem):
    """krylov.jl:37-52 (ends with @synchronize())"""
    _policy(execution_policy)
    matrix_free_matvec_prod_(p, hx, hy, c, p_hat)
    _ctx().synchronize()


def cg_(x_in, b, hx, hy, c, tol, Nmax, execution_policy=parallel_shmem, verbose=False, return_iters=False):
    """krylov.jl:55-91: unpreconditioned CG from x = 0; overwrites x_in; returns sqrt(sum(r.^2)/(nx*ny))."""
    _policy(execution_policy)
    rms, it = C.c_double(0.0), C.c_int(0)
    _ctx().call("fpr_cg2d", fptr(x_in, 2), fptr(b, 2), hx, hy, c, tol, int(Nmax), *b.shape, C.byref(rms), C.byref(it))
    if verbose:
        print("CG stopped after %d iterations, r_rms = %g" % (it.value, rms.value))
    return (rms.value, it.value) if return_iters else rms.value


def Vcycle_2DPoisson_(u_f, rhs, h, c, tol, coarse_solve_size, coarse_solver, execution_policy, apply_BCs,
                      prealloc_dict=None):
    """multigrid.jl:91-170; returns res_rms."""
    _policy(execution_policy)
    out = C.c_double(0.0)
    try:
        _ctx().call("fpr_vcycle2d", fptr(u_f, 2), fptr(rhs, 2), h, c, tol, int(coarse_solve_size),
                    CoarseSolver_t(coarse_solver).value, int(bool(apply_BCs)), *u_f.shape, C.byref(out))
    except FprError as e:
        if e.code == -3:
            raise RuntimeError("ERROR:not a power of 2") from e  # multigrid.jl:95-97
        raise
    return out.value


def mgsolve_raw(c_, u, f, h, c, tol, niters, apply_BCs, opt):
    """fpr_mgsolve2d on the context c_ without the wrapper's printing and warnings (callable from a worker thread: ctypes
    releases the GIL for the duration of the call).  Returns (r_rms, ncycles, f_rms, converged)."""
    nx, ny = u.shape
    rms, ncyc, frms, conv = C.c_double(0.0), C.c_int(0), C.c_double(0.0), C.c_int(0)
    c_.call("fpr_mgsolve2d", fptr(u, 2), fptr(f, 2), h, c, tol, int(niters), int(bool(apply_BCs)),
            int(opt.coarse_solve_size), CoarseSolver_t(opt.coarse_solver).value, nx, ny, C.byref(rms),
            C.byref(ncyc), None, C.byref(frms), C.byref(conv))
    return rms.value, ncyc.value, frms.value, bool(conv.value)


def MGsolve_2DPoisson_(u, f, h, c, tol, niters, apply_BCs, opt=None, verbose=False, prealloc_dict=None,
                       return_history=False):
    """multigrid.jl:41-84; returns r_rms (and, with return_history, (r_rms, history, f_rms, coarse_iters))."""
    opt = opt if opt is not None else MGOpt()
    _policy(opt.execution_policy)
    nx, ny = u.shape
    rms, ncyc, frms, conv = C.c_double(0.0), C.c_int(0), C.c_double(0.0), C.c_int(0)
    hist = (C.c_double * max(int(niters), 1))()
    c_ = _ctx()
    try:
        c_.call("fpr_mgsolve2d", fptr(u, 2), fptr(f, 2), h, c, tol, int(niters), int(bool(apply_BCs)),
                int(opt.coarse_solve_size), CoarseSolver_t(opt.coarse_solver).value, nx, ny, C.byref(rms),
                C.byref(ncyc), hist, C.byref(frms), C.byref(conv))
    except FprError as e:
        if e.code == -4:
            raise AssertionError(str(e)) from e  # multigrid.jl:45-46
        if e.code == -3:
            raise RuntimeError("ERROR:not a power of 2") from e
        raise
    history = np.array(hist[: ncyc.value])
    if verbose:
        for i, r in enumerate(history):
            print("%d %g" % (i + 1, r / frms.value if frms.value else float("nan")))
        if conv.value:
            print("V-cycle multigrid converged in %d iterations." % ncyc.value)
    if not conv.value:
        warnings.warn("V-cycle multigrid failed to converge within %d iterations." % niters)  # :78-80
    if return_history:
        return rms.value, history, frms.value, c_.L.fpr_last_coarse_iters(c_.h)
    return rms.value
