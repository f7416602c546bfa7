"""ctypes binding of the CPU ORACLE (oracle/fpr_oracle.c) -- test infrastructure, NOT the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All arrays are numpy float64 in Fortran (column-major) order, the reference's Julia layout.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    so = os.path.join(_HERE, "build", "libfpr_oracle.so")
    src = os.path.join(_HERE, "fpr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _p(a):
    assert a.dtype == np.float64 and a.flags.f_contiguous, "oracle arrays must be float64, Fortran order"
    return a.ctypes.data_as(_dp)


def farr(*shape):
    return np.zeros(shape, dtype=np.float64, order="F")


def asf(a):
    return np.asfortranarray(a, dtype=np.float64)


class Oracle:
    def __init__(self, openmp=False):
        build()
        name = "libfpr_oracle_omp.so" if openmp else "libfpr_oracle.so"
        self.lib = L = C.CDLL(os.path.join(_HERE, "build", name))
        d, i, l, z = C.c_double, C.c_int, C.c_long, C.c_size_t
        sig = {
            "orc_sumsq": (d, [_dp, z]),
            "orc_dot": (d, [_dp, _dp, z]),
            "orc_dot2": (d, [_dp, _dp, z]),
            "orc_has_openmp": (i, []),
            "orc_diffusion3d_step": (None, [_dp] * 4 + [i] * 3 + [d] * 8),
            "orc_diffusion3d_step_fma": (None, [_dp] * 4 + [i] * 3 + [d] * 8),
            "orc_diffusion3d_flux": (None, [_dp] * 4 + [i] * 3 + [d] * 4),
            "orc_diffusion3d_dHdtau": (None, [_dp] * 6 + [i] * 3 + [d] * 4),
            "orc_diffusion3d_update": (None, [_dp] * 2 + [i] * 3 + [d]),
            "orc_sumsq_scaled": (d, [_dp, z, d]),
            "orc_init_gaussian": (None, [_dp] + [i] * 3 + [d] * 6 + [i] * 3),
            "orc_apply_bc3d": (None, [_dp] + [i] * 3 + [C.POINTER(i)] * 2),
            "orc_diffusion3d_solve": (l, [_dp] + [i] * 3 + [d] * 5 + [i, d, l, l, C.POINTER(l), _dp, _dp, _dp]),
            "orc_diffusion3d_array_solve": (i, [_dp] + [i] * 3 + [d] * 7 + [l, l, i, C.POINTER(l), _dp, _dp]),
            "orc_residual2d": (None, [_dp, _dp, d, d, _dp, i, i]),
            "orc_jacobi2d": (d, [_dp, _dp, d, d, _dp, i, i, d]),
            "orc_bc_dirichlet2d": (None, [_dp, i, i]),
            "orc_bc_neumann2d": (None, [_dp, i, i]),
            "orc_bc2d": (None, [_dp, i, i]),
            "orc_restrict2d": (None, [_dp, _dp, i, i, i]),
            "orc_prolongate2d": (None, [_dp, _dp, i, i, i]),
            "orc_laplace_apply2d": (None, [_dp, d, d, d, _dp, i, i]),
            "orc_cg2d": (d, [_dp, _dp, d, d, d, d, i, i, i, C.POINTER(i)]),
            "orc_cg2d_plain": (d, [_dp, _dp, d, d, d, d, i, i, i, C.POINTER(i)]),
            "orc_last_coarse_solves": (i, [C.POINTER(i), i]),
            "orc_vcycle2d": (d, [_dp, _dp, d, d, d, i, i, i, i, i]),
            "orc_mgsolve2d": (d, [_dp, _dp, d, d, d, i, i, i, i, i, i, _dp, C.POINTER(i), _dp]),
            "orc_last_coarse_iters": (l, []),
            "orc_compute_velocity": (None, [_dp, d, d, _dp, _dp, i, i]),
            "orc_compute_Ra_dTdx": (None, [d, d, _dp, _dp, i, i]),
            "orc_compute_diffusion2d": (None, [_dp, d, d, d, _dp, i, i]),
            "orc_compute_advection2d_x": (None, [_dp, d, _dp, _dp, i, i]),
            "orc_compute_advection2d_y": (None, [_dp, d, _dp, _dp, i, i]),
        }
        for k, (res, args) in sig.items():
            f = getattr(L, k)
            f.restype, f.argtypes = res, args

    # ---- Part 1 ----
    def diffusion3d_step(self, Ht, Htau, Htau2, dHdtau, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz):
        nx, ny, nz = Ht.shape
        self.lib.orc_diffusion3d_step(_p(Ht), _p(Htau), _p(Htau2), _p(dHdtau), nx, ny, nz,
                                      dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)

    def diffusion3d_step_fma(self, Ht, Htau, Htau2, dHdtau, dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz):
        """The checker of the library's opt-in option fp_contract = 1 (explicit fma sequence; NOT the reference's arithmetic)."""
        nx, ny, nz = Ht.shape
        self.lib.orc_diffusion3d_step_fma(_p(Ht), _p(Htau), _p(Htau2), _p(dHdtau), nx, ny, nz,
                                          dtau, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)

    def diffusion3d_flux(self, qx, qy, qz, Htau, D, dx, dy, dz):
        nx, ny, nz = Htau.shape
        self.lib.orc_diffusion3d_flux(_p(qx), _p(qy), _p(qz), _p(Htau), nx, ny, nz, D, dx, dy, dz)

    def diffusion3d_dHdtau(self, dHdtau, Htau, Ht, qx, qy, qz, dt, dx, dy, dz):
        nx, ny, nz = Htau.shape
        self.lib.orc_diffusion3d_dHdtau(_p(dHdtau), _p(Htau), _p(Ht), _p(qx), _p(qy), _p(qz), nx, ny, nz, dt, dx, dy, dz)

    def diffusion3d_update(self, Htau, dHdtau, dtau):
        nx, ny, nz = Htau.shape
        self.lib.orc_diffusion3d_update(_p(Htau), _p(dHdtau), nx, ny, nz, dtau)

    def sumsq_scaled(self, x, scale=1.0):
        return self.lib.orc_sumsq_scaled(_p(x), x.size, scale)

    def init_gaussian(self, shape, dx, dy, dz, center, coords=(0, 0, 0)):
        H = farr(*shape)
        self.lib.orc_init_gaussian(_p(H), *shape, dx, dy, dz, *center, *coords)
        return H

    def apply_bc3d(self, H, coords, dims):
        ci = (C.c_int * 3)(*coords)
        di = (C.c_int * 3)(*dims)
        self.lib.orc_apply_bc3d(_p(H), *H.shape, ci, di)

    def diffusion3d_solve(self, Ht, lx=10.0, ly=10.0, lz=10.0, D=1.0, dt=0.2, nt=5, tol=1e-8,
                          iter_max=100000, fixed_iters=0):
        """Runs the single-rank host loop in place on Ht; returns (iters[nt], err[nt], Htau, dHdtau)."""
        nx, ny, nz = Ht.shape
        iters = (C.c_long * nt)()
        err = np.zeros(nt)
        Htau = farr(nx, ny, nz)
        dH = farr(nx, ny, nz)
        self.lib.orc_diffusion3d_solve(_p(Ht), nx, ny, nz, lx, ly, lz, D, dt, nt, tol, iter_max,
                                       fixed_iters, iters, err.ctypes.data_as(_dp), _p(Htau), _p(dH))
        return list(iters), err, Htau, dH

    def diffusion3d_array_solve(self, Ht, lx=10.0, ly=10.0, lz=10.0, D=1.0, dt=0.2, ttot=1.0, tol=1e-8,
                                iter_max=100000, fixed_iters=0):
        """diffusion_3D_array_programming (part1_array_programming.jl:20-92) in place on Ht;
        returns (iters[steps], err[steps], dHdt (nx-2,ny-2,nz-2))."""
        nx, ny, nz = Ht.shape
        cap = 4096
        iters = (C.c_long * cap)()
        err = np.zeros(cap)
        dH = farr(nx - 2, ny - 2, nz - 2)
        n = self.lib.orc_diffusion3d_array_solve(_p(Ht), nx, ny, nz, lx, ly, lz, D, dt, ttot, tol, iter_max,
                                                 fixed_iters, cap, iters, err.ctypes.data_as(_dp), _p(dH))
        return list(iters)[:n], err[:n].copy(), dH

    # ---- Part 2 ----
    def residual2d(self, u, f, h, c, res):
        self.lib.orc_residual2d(_p(u), _p(f), h, c, _p(res), *u.shape)

    def jacobi2d(self, u, f, h, c, res, alpha=4.0 / 5.0):
        return self.lib.orc_jacobi2d(_p(u), _p(f), h, c, _p(res), *u.shape, alpha)

    def bc_dirichlet2d(self, T):
        self.lib.orc_bc_dirichlet2d(_p(T), *T.shape)

    def bc_neumann2d(self, T):
        self.lib.orc_bc_neumann2d(_p(T), *T.shape)

    def bc2d(self, T):
        self.lib.orc_bc2d(_p(T), *T.shape)

    def restrict2d(self, fine, coarse, apply_BCs=False):
        self.lib.orc_restrict2d(_p(fine), _p(coarse), *fine.shape, int(apply_BCs))

    def prolongate2d(self, coarse, fine, apply_BCs=False):
        self.lib.orc_prolongate2d(_p(coarse), _p(fine), *fine.shape, int(apply_BCs))

    def laplace_apply2d(self, T, hx, hy, c, dT2):
        self.lib.orc_laplace_apply2d(_p(T), hx, hy, c, _p(dT2), *T.shape)

    def cg2d(self, x, b, hx, hy, c, tol, Nmax):
        it = C.c_int(0)
        r = self.lib.orc_cg2d(_p(x), _p(b), hx, hy, c, tol, Nmax, *b.shape, C.byref(it))
        return r, it.value

    def cg2d_plain(self, x, b, hx, hy, c, tol, Nmax):
        """cg! with plain pairwise sums (the reference's CPU arithmetic up to the order inside a 1024-element base case)."""
        it = C.c_int(0)
        r = self.lib.orc_cg2d_plain(_p(x), _p(b), hx, hy, c, tol, Nmax, *b.shape, C.byref(it))
        return r, it.value

    def last_coarse_solve_iters(self):
        """CG iterations of every coarse solve of the last mgsolve2d call (the first 256)."""
        buf = (C.c_int * 256)()
        n = self.lib.orc_last_coarse_solves(buf, 256)
        return list(buf)[: min(n, 256)]

    def vcycle2d(self, u, rhs, h, c, tol, coarse_solve_size=5, coarse_solver=0, apply_BCs=False):
        return self.lib.orc_vcycle2d(_p(u), _p(rhs), h, c, tol, coarse_solve_size, coarse_solver,
                                     int(apply_BCs), *u.shape)

    def mgsolve2d(self, u, f, h, c, tol, niters, apply_BCs=False, coarse_solve_size=5, coarse_solver=0):
        """Returns (r_rms, history[ncycles], f_rms)."""
        hist = np.zeros(max(niters, 1))
        n = C.c_int(0)
        frms = np.zeros(1)
        r = self.lib.orc_mgsolve2d(_p(u), _p(f), h, c, tol, niters, int(apply_BCs), coarse_solve_size,
                                   coarse_solver, *u.shape, hist.ctypes.data_as(_dp), C.byref(n),
                                   frms.ctypes.data_as(_dp))
        return r, hist[: n.value].copy(), float(frms[0])

    def last_coarse_iters(self):
        return self.lib.orc_last_coarse_iters()

    # ---- NEXT 8f-1 ----
    def compute_velocity(self, S, hx, hy, vx, vy):
        self.lib.orc_compute_velocity(_p(S), hx, hy, _p(vx), _p(vy), *S.shape)

    def compute_Ra_dTdx(self, Ra, hx, T, out):
        self.lib.orc_compute_Ra_dTdx(Ra, hx, _p(T), _p(out), *T.shape)

    def compute_diffusion2d(self, T, hx, hy, k, dT2):
        self.lib.orc_compute_diffusion2d(_p(T), hx, hy, k, _p(dT2), *T.shape)

    def compute_advection2d_x(self, T, hx, vx, dTx):
        self.lib.orc_compute_advection2d_x(_p(T), hx, _p(vx), _p(dTx), *T.shape)

    def compute_advection2d_y(self, T, hy, vy, dTy):
        self.lib.orc_compute_advection2d_y(_p(T), hy, _p(vy), _p(dTy), *T.shape)
