"""GPU parity tests, Part 2 (2D multigrid): HIP path (through the C ABI) vs the CPU oracle.
Pointwise kernels bit-exact; norms 1e-13; V-cycle residual histories 1e-10 relative (north_star)."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

from fixtures_io import load_bin, splitmix64_uniform
from oracle.oracle import asf, farr

pytestmark = pytest.mark.gpu

SHAPES2 = [(64, 64), (33, 17), (257, 65), (5, 5), (3, 3), (129, 130), (66, 7)]


def rnd(shape, seed):
    return asf(splitmix64_uniform(int(np.prod(shape)), seed).reshape(shape, order="F"))


@pytest.mark.parametrize("shape", SHAPES2, ids=str)
def test_residual_laplace_bc_bit_exact(fpr, oracle, shape):
    F, mg = fpr, fpr.multigrid
    u, f = rnd(shape, 1), rnd(shape, 2)
    h, c = 1.0 / 63, 3.1415
    ref = asf(np.full(shape, 4.0))
    oracle.residual2d(u, f, h, c, ref)
    res = F.asdevice(np.full(shape, 4.0))
    mg.residual_2DPoisson_wrapper_(F.asdevice(u), F.asdevice(f), h, c, res, mg.parallel)
    assert np.array_equal(F.tonumpy(res), ref)
    ref2 = asf(np.full(shape, 4.0))
    oracle.laplace_apply2d(u, 0.013, 0.017, 2.5, ref2)
    out = F.asdevice(np.full(shape, 4.0))
    mg.matrix_free_matvec_prod_wrapper_(F.asdevice(u), 0.013, 0.017, 2.5, out)
    assert np.array_equal(F.tonumpy(out), ref2)
    T = u.copy(order="F")
    oracle.bc2d(T)
    g = F.asdevice(u)
    mg.apply_boundary_conditions_(g)
    assert np.array_equal(F.tonumpy(g), T)
    T = u.copy(order="F"); oracle.bc_neumann2d(T)
    g = F.asdevice(u); mg.apply_boundary_conditions_neumann_(g)
    assert np.array_equal(F.tonumpy(g), T)
    T = u.copy(order="F"); oracle.bc_dirichlet2d(T)
    g = F.asdevice(u); mg.apply_boundary_conditions_dirichlet_(g)
    assert np.array_equal(F.tonumpy(g), T)


def test_residual_operator_identity(fpr):
    """test/multigrid.jl:102-138 on the device path."""
    F, mg = fpr, fpr.multigrid
    n, c = 64, 3.1415
    h = 1.0 / (n - 1)
    u = rnd((n, n), 7)
    u[0, :] = u[-1, :] = 0.0
    u[:, 0] = u[:, -1] = 0.0
    f = rnd((n, n), 8)
    res = F.fzeros(n, n)
    for pol in (mg.parallel, mg.parallel_shmem):
        mg.residual_2DPoisson_wrapper_(F.asdevice(u), F.asdevice(f), h, c, res, pol)
        m = n - 2
        dx = sp.diags([np.ones(m - 1), -2 * np.ones(m), np.ones(m - 1)], [-1, 0, 1])
        A = (sp.kron(dx, sp.identity(m)) + sp.kron(sp.identity(m), dx)) / h ** 2 - c * sp.identity(m * m)
        ref = A @ u[1:-1, 1:-1].ravel(order="F") - f[1:-1, 1:-1].ravel(order="F")
        got = F.tonumpy(res)[1:-1, 1:-1].ravel(order="F")
        assert np.linalg.norm(got - ref) <= 1.5e-8 * np.linalg.norm(ref)
    with pytest.raises(RuntimeError):
        mg.residual_2DPoisson_wrapper_(F.asdevice(u), F.asdevice(f), h, c, res, mg.serial)  # multigrid.jl:233-234


@pytest.mark.parametrize("shape", [(33, 33), (257, 65), (17, 5)], ids=str)
def test_jacobi_iteration(fpr, oracle, shape):
    F, mg = fpr, fpr.multigrid
    u, f = rnd(shape, 3), rnd(shape, 4)
    res_ref = asf(np.zeros(shape))
    res_ref[0, :] = 0.5  # a caller-owned boundary value of res takes part in the norm (multigrid.jl:252)
    u_ref = u.copy(order="F")
    h, c = 1.0 / 32, 0.7
    r_ref = oracle.jacobi2d(u_ref, f, h, c, res_ref)
    gu, gres = F.asdevice(u), F.asdevice(np.where(np.arange(shape[0])[:, None] == 0, 0.5, 0.0) * np.ones(shape))
    r = mg.iteration_2DPoisson_(gu, F.asdevice(f), h, c, gres, mg.parallel_shmem)
    assert np.array_equal(F.tonumpy(gres), res_ref) and np.array_equal(F.tonumpy(gu), u_ref)
    assert abs(r - r_ref) <= 1e-13 * r_ref


@pytest.mark.parametrize("bc", [False, True])
@pytest.mark.parametrize("shape", [(17, 9), (65, 65), (257, 65), (5, 5), (9, 33)], ids=str)
def test_restrict_prolongate_bit_exact(fpr, oracle, shape, bc):
    F, mg = fpr, fpr.multigrid
    nx, ny = shape
    cs = (1 + (nx - 1) // 2, 1 + (ny - 1) // 2)
    fine, coarse = rnd(shape, 5), rnd(cs, 6)
    cref = asf(np.full(cs, 3.0))
    oracle.restrict2d(fine, cref, bc)
    gc = F.asdevice(np.full(cs, 3.0))
    mg.restrict_wrapper_(F.asdevice(fine), gc, bc)
    assert np.array_equal(F.tonumpy(gc), cref)
    fref = asf(np.full(shape, 3.0))
    oracle.prolongate2d(coarse, fref, bc)
    gf = F.asdevice(np.full(shape, 3.0))
    mg.prolongate_wrapper_(F.asdevice(coarse), gf, bc)
    assert np.array_equal(F.tonumpy(gf), fref)
    gu = F.asdevice(fine)
    mg.correct_(gu, gf)
    assert np.array_equal(F.tonumpy(gu), fine - fref)


def test_cg_matches_oracle(fpr, oracle):
    """test/krylov.jl:19-36 + iteration-count parity with the oracle."""
    F, mg = fpr, fpr.multigrid
    n = 66
    h = 1.0 / (n - 1)
    b = asf(np.ones((n, n)))
    b[0, :] = b[-1, :] = 0.0
    b[:, 0] = b[:, -1] = 0.0
    xr = farr(n, n)
    r_ref, it_ref = oracle.cg2d(xr, b, h, h, 3.14, 1e-6, 1000)
    x = F.asdevice(np.full((n, n), 5.0))  # cg! starts from zero and overwrites x_in
    r, it = mg.cg_(x, F.asdevice(b), h, h, 3.14, 1e-6, 1000, return_iters=True)
    assert r < 1e-6 * math.sqrt((b ** 2).sum() / n ** 2)
    # cg!'s dot products are Dot2 sums on both sides (twofold precision, rounded once: order-independent), everything else is
    # pointwise: the same iterates bit for bit
    assert it == it_ref
    assert r == r_ref
    assert np.array_equal(F.tonumpy(x), xr)
    # rhs with non-zero boundary (as the Neumann rows of a restricted residual): p_hat keeps b's boundary
    b2 = rnd((33, 17), 9)
    xr = farr(33, 17)
    # (the system is inconsistent, CG does not converge and amplifies rounding: few iterations only)
    r_ref, it_ref = oracle.cg2d(xr, b2, 0.1, 0.1, 1.0, 1e-8, 12)
    x = F.fzeros(33, 17)
    r, it = mg.cg_(x, F.asdevice(b2), 0.1, 0.1, 1.0, 1e-8, 12, return_iters=True)
    assert it == it_ref == 12 and r == r_ref
    assert np.array_equal(F.tonumpy(x), xr)


@pytest.mark.parametrize("fences", [0, 1])
@pytest.mark.parametrize("shape,nmax", [((257, 257), 700), ((66, 66), 1000), ((33, 17), 12), ((130, 35), 5), ((257, 65), 64),
                                         ((65, 257), 65), ((40, 9), 129)], ids=str)
def test_cg_launch_forms_agree_bit_for_bit(fpr, oracle, shape, nmax, fences):
    """cg! as ONE persistent launch (default where the grid fits 16 workgroups), as two dependent launches per iteration (the
    direction update rides in the next matvec), three, or five (one kernel per operation): the same operations on the same
    operands, dot products as Dot2 sums (order-independent) -- x, the returned residual and the iteration count are identical
    among all four AND equal to the oracle's, whether the solve converges, stops at Nmax, or crosses a host-poll boundary (64).
    fences = 1: option handoff_fences -- the persistent kernel's hand-off of the r edges also carries an agent-scope release / acquire
    pair (the form inside the HIP memory model; the default rests on sc1 stores, drains and sc1 loads alone): same results."""
    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    c.set_option("handoff_fences", fences)
    b = rnd(shape, 77)
    if nmax > 100:   # a consistent system that converges
        b[0, :] = b[-1, :] = 0.0
        b[:, 0] = b[:, -1] = 0.0
    outs = []
    try:
        # one persistent launch (3), two dependent launches per iteration (2), three (1), five (0)
        for form in (3, 2, 1, 0):
            c.set_option("cg_fused", form)
            x = F.asdevice(np.full(shape, 3.0))
            r, it = mg.cg_(x, F.asdevice(b), 0.05, 0.07, 0.9, 1e-7, nmax, return_iters=True)
            outs.append((r, it, F.tonumpy(x)))
        assert c.get_option("cg_persistent_timeouts") == 0
    finally:
        c.set_option("cg_fused", 3)
        c.set_option("handoff_fences", 0)
    for r, it, x in outs[1:]:
        assert it == outs[0][1] and r == outs[0][0]
        assert np.array_equal(x, outs[0][2])
    assert outs[0][1] <= nmax and np.isfinite(outs[0][2]).all()
    xr = farr(*shape)
    r_ref, it_ref = oracle.cg2d(xr, b, 0.05, 0.07, 0.9, 1e-7, nmax)
    assert it_ref == outs[0][1] and r_ref == outs[0][0]
    assert np.array_equal(xr, outs[0][2])


@pytest.mark.parametrize("solver", ["jacobi", "conjugate_gradient"])
@pytest.mark.parametrize("bc,c", [(False, 0.0), (True, 78.66)])
@pytest.mark.parametrize("shape,css", [((257, 65), 5), ((129, 129), 9), ((65, 65), 65)], ids=str)
def test_single_vcycle_matches_oracle(fpr, oracle, shape, css, bc, c, solver):
    F, mg = fpr, fpr.multigrid
    u0, f = rnd(shape, 21), rnd(shape, 22)
    if solver == "conjugate_gradient" and not bc:
        # cg! keeps b's boundary inside p_hat (krylov.jl:59-61): with a non-zero boundary on the coarse rhs (random f, or the
        # Neumann rows of apply_BCs) the iteration is inconsistent and grows to 1e10..NaN in the oracle as well.  The well-posed
        # case gets a zero boundary; the Neumann case (bc) runs as it is -- through round 3 it was skipped as chaotic, but with
        # Dot2 dot products both sides walk through the same iterates bit for bit, wherever they lead.
        f[0, :] = f[-1, :] = 0.0
        f[:, 0] = f[:, -1] = 0.0
    h = 1.0 / (shape[1] - 1)
    sv = getattr(mg, solver)
    u_ref = u0.copy(order="F")
    with np.errstate(all="ignore"):
        r_ref = oracle.vcycle2d(u_ref, f, h, c, 1e-7, css, sv.value, bc)
    if solver == "conjugate_gradient" and bc and not (abs(r_ref) < 1e3 * np.abs(f).max()):
        # the inconsistent coarse system let CG run away (residual 1e12 and beyond): sums that cancel to nothing are beyond
        # twofold precision too, and what comes out is rounding noise on both sides
        pytest.skip("CG coarse solve with Neumann rows on the coarse rhs diverges in the reference algorithm (oracle residual %.3g)" % r_ref)
    gu = F.asdevice(u0)
    r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), h, c, 1e-7, css, sv, mg.parallel_shmem, bc)
    assert (math.isnan(r) and math.isnan(r_ref)) or abs(r - r_ref) <= 1e-10 * abs(r_ref)
    # same sweep / iteration counts, pointwise kernels and order-independent dot products: bit-exact with either coarse solver
    assert np.array_equal(F.tonumpy(gu), u_ref, equal_nan=True)


def test_vcycle_errors(fpr):
    F, mg = fpr, fpr.multigrid
    u, f = F.fzeros(34, 34), F.fzeros(34, 34)
    with pytest.raises(RuntimeError, match="not a power of 2"):
        mg.Vcycle_2DPoisson_(u, f, 1 / 33, 0.0, 1e-6, 5, mg.jacobi, mg.parallel, False)
    u, f = F.fzeros(33, 33), F.fzeros(33, 33)
    opt = mg.MGOpt()
    opt.coarse_solve_size = 6
    with pytest.raises(AssertionError):
        mg.MGsolve_2DPoisson_(u, f, 1 / 32, 0.0, 1e-6, 5, False, opt=opt)
    opt.coarse_solve_size = 65
    with pytest.raises(AssertionError):
        mg.MGsolve_2DPoisson_(u, f, 1 / 32, 0.0, 1e-6, 5, False, opt=opt)


def test_mgsolve_fortran_fixture(fpr, oracle):
    """test/part2.jl:8,29,36: S from Winit.bin vs S.bin (1e-8) and the oracle's 14-cycle history."""
    F, mg = fpr, fpr.multigrid
    W, Sref = load_bin("Winit.bin"), load_bin("S.bin")
    nx, ny = W.shape
    h = 1.0 / (ny - 1.0)
    S_o = farr(nx, ny)
    r_o, hist_o, frms_o = oracle.mgsolve2d(S_o, W, h, 0.0, 1e-12, 50)
    S = F.fzeros(nx, ny)
    r, hist, frms, cit = mg.MGsolve_2DPoisson_(S, F.asdevice(W), h, 0.0, 1e-12, 50, False, return_history=True)
    got = F.tonumpy(S)
    assert np.abs(got[1:-1, 1:-1] - Sref[1:-1, 1:-1]).max() < 1e-8
    assert len(hist) == len(hist_o) == 14
    assert np.allclose(hist, hist_o, rtol=1e-10, atol=0)
    assert abs(frms - frms_o) <= 1e-13 * frms_o
    assert np.array_equal(got, S_o)
    assert cit == oracle.last_coarse_iters()


@pytest.mark.parametrize("solver", ["jacobi", "conjugate_gradient"])
@pytest.mark.parametrize("l", [2, 3])
@pytest.mark.parametrize("k", [7, 8, 9])
def test_multigrid_convergence(fpr, oracle, k, l, solver):
    """test/multigrid.jl:30-58 sweep on the device path + history parity with the oracle."""
    F, mg = fpr, fpr.multigrid
    n = 2 ** k + 1
    h = 1.0 / (n - 1)
    tol = 1e-6
    xref = farr(n, n)
    xref[1:-1, 1:-1] = splitmix64_uniform((n - 2) ** 2, 3).reshape((n - 2, n - 2), order="F")
    b = farr(n, n)
    oracle.laplace_apply2d(xref, h, h, 0.0, b)
    opt = mg.MGOpt()
    opt.coarse_solve_size = 2 ** l + 1
    opt.coarse_solver = getattr(mg, solver)
    for pol in (mg.parallel, mg.parallel_shmem):
        opt.execution_policy = pol
        x = F.fzeros(n, n)
        r, hist, frms, _ = mg.MGsolve_2DPoisson_(x, F.asdevice(b), h, 0.0, tol, 20, False, opt=opt, return_history=True)
        assert r < tol * math.sqrt((b ** 2).sum() / (n * n))
    xo = farr(n, n)
    r_o, hist_o, _ = oracle.mgsolve2d(xo, b, h, 0.0, tol, 20, False, opt.coarse_solve_size, opt.coarse_solver.value)
    assert len(hist) == len(hist_o) and np.allclose(hist, hist_o, rtol=1e-10, atol=0)


def test_semi_implicit_T_solve_history(fpr, oracle):
    """apply_BCs=true, c>0 path (never exercised by the reference's tests): parity means the
    same residual history as the oracle, including non-convergence in 50 cycles (a warning, not an error)."""
    F, mg = fpr, fpr.multigrid
    T0 = load_bin("Tinit.bin")
    nx, ny = T0.shape
    h = 1.0 / (ny - 1.0)
    c = 78.6638
    rhs = asf(-c * T0)
    T_o = T0.copy(order="F")
    r_o, hist_o, _ = oracle.mgsolve2d(T_o, rhs, h, c, 1e-7, 12, True)
    T = F.asdevice(T0)
    with pytest.warns(UserWarning, match="failed to converge"):
        r, hist, _, _ = mg.MGsolve_2DPoisson_(T, F.asdevice(rhs), h, c, 1e-7, 12, True, return_history=True)
    assert len(hist) == len(hist_o) == 12
    assert np.allclose(hist, hist_o, rtol=1e-10, atol=0)
    assert np.abs(F.tonumpy(T) - T_o).max() <= 1e-12 * np.abs(T_o).max()


def test_full_size_4097_properties(fpr):
    """BASELINE config 3 size (4097^2, multigrid_bench.jl protocol): 7 V-cycles at l=2 (mesh independent,
    SURVEY 4.4); the returned r_rms is consistent with an independent residual evaluation; 5-level
    variant (l=8) with the CG coarse solver converges too."""
    F, mg = fpr, fpr.multigrid
    n = 4097
    h = 1.0 / (n - 1)
    b = F.asdevice(splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    x = F.fzeros(n, n)
    r, hist, frms, _ = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, return_history=True)
    assert len(hist) == 7 and r < 1e-6 * frms
    assert np.all(hist[1:] < 0.25 * hist[:-1])  # contraction ~0.13 per cycle
    res = F.fzeros(n, n)
    mg.residual_2DPoisson_(x, b, h, 0.0, res)
    rr = math.sqrt(F.part1.local_sumsq(res) / (n * n))
    assert rr < r  # the returned value is measured before the last Jacobi update
    assert rr > 0.2 * r
    opt = mg.MGOpt()
    opt.coarse_solve_size = 257
    opt.coarse_solver = mg.conjugate_gradient
    x.zero_()
    r5, hist5, _, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
    assert r5 < 1e-6 * frms and len(hist5) <= 12 and cit > 0


# CG coarse solver on large coarse grids: through round 3 a documented exception (the ~630 CG iterations of a coarse solve
# amplified the different summation orders of the dot products krylov.jl:64,69,83 to 4e-7 in the last residuals of a solve,
# profiles/r3_cg_parity.txt).  The dot products are now Dot2 sums on both sides (twofold precision, rounded once: the order no
# longer matters), so a V-cycle with cg! is as exact as one with the Jacobi coarse solver: fields bit for bit, history 1e-10.


def test_config3_known_answer_k10_l6_jacobi(fpr, oracle):
    """SURVEY 4.4 known answer of the multigrid_bench.jl protocol (x=0, b~U[0,1) incl. boundary, tol 1e-6):
    k=10, l=6 (coarse 65^2, Jacobi capped at 20*65 = 1300 sweeps per V-cycle) takes exactly 11 V-cycles."""
    F, mg = fpr, fpr.multigrid
    n = 1025
    h = 1.0 / (n - 1)
    b = asf(splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    opt = mg.MGOpt()
    opt.coarse_solve_size, opt.coarse_solver = 65, mg.jacobi
    xo = farr(n, n)
    r_o, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, 0.0, 1e-6, 100, False, 65, 0)
    x = F.fzeros(n, n)
    r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, F.asdevice(b), h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
    assert len(hist) == len(hist_o) == 11
    assert np.allclose(hist, hist_o, rtol=1e-10, atol=0)
    assert cit == oracle.last_coarse_iters() == 11 * 1300   # the coarse solve never reaches its tolerance
    assert np.array_equal(F.tonumpy(x), xo)


def test_bench_vcycle_config_4097_l2_full_solve_against_the_oracle(fpr, oracle):
    """The configuration bench.py's `vcycle` block times -- 4097^2, 11 grids (l = 2, coarse 5^2), Jacobi coarse solver,
    multigrid_bench.jl:27-42 protocol (x = 0, b ~ U[0,1) over the whole array, c = 0, tol 1e-6) -- as a FULL solve against the
    oracle at its own size: 7 V-cycles, residual history to 1e-10, the field bit for bit.  This runs the 11-level
    path (finest marches, seam passes between cycles, k_mid_down/up, the LDS-resident k_mg_small) under a 4097^2 finest level."""
    F, mg = fpr, fpr.multigrid
    n = 4097
    h = 1.0 / (n - 1)
    b = asf(splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    xo = farr(n, n)
    r_o, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, 0.0, 1e-6, 100, False, 5, 0)
    x = F.fzeros(n, n)
    opt = mg.MGOpt()
    opt.coarse_solve_size, opt.coarse_solver = 5, mg.jacobi
    r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, F.asdevice(b), h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
    assert len(hist) == len(hist_o) == 7
    assert abs(frms - frms_o) <= 1e-13 * frms_o
    assert np.allclose(hist, hist_o, rtol=1e-10, atol=0), (hist, hist_o)
    assert cit == oracle.last_coarse_iters()
    assert np.array_equal(F.tonumpy(x), xo)


@pytest.mark.parametrize("n,bcs", [(513, False), (513, True), (1025, True), (2049, False), (4097, True), (4097, False)])
def test_seam_pass_one_and_two_columns_per_lane(fpr, oracle, n, bcs):
    """k_seam_march_v3 (a lane owns columns g and g + 64 of a 128-column strip; the halves' x-neighbours through wave rotations and
    shifts; taken at 4097^2, where one workgroup per CU still leaves chunks of 128 rows) and k_seam_march_v2 (one column per lane: every
    smaller grid) against the oracle: full solves with and without apply_BCs (Neumann columns cross the halves' seam handling at the first
    and last strip), fields bit for bit, histories to 1e-10 (at 4097^2 the oracle's first cycles: the solve stops at niters)."""
    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    h = 1.0 / (n - 1)
    b = asf(splitmix64_uniform(n * n, 3).reshape((n, n), order="F"))
    niters = 6 if n == 4097 else 12
    xo = farr(n, n)
    r_o, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, 0.0, 1e-9, niters, bcs, 5, 0)
    x = F.fzeros(n, n)
    r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, F.asdevice(b), h, 0.0, 1e-9, niters, bcs, opt=mg.MGOpt(), return_history=True)
    assert len(hist) == len(hist_o)
    assert np.allclose(hist, hist_o, rtol=1e-10, atol=0), (hist, hist_o)
    assert np.array_equal(F.tonumpy(x), xo)


@pytest.mark.parametrize("n,bcs,css,solver,tol", [(129, False, 5, 0, 1e-9), (513, True, 5, 0, 1e-9), (1025, False, 5, 0, 1e-9),
                                                   (2049, True, 5, 0, 1e-9), (2049, False, 5, 0, 1e-3), (1025, False, 129, 1, 1e-9),
                                                   (4097, False, 5, 0, 1e-6)])
def test_cycle_finish_rides_on_the_next_pass(fpr, oracle, n, bcs, css, solver, tol):
    """The finish of cycle k (sum of the seam pass's partials, r_rms, the exit test of multigrid.jl:70, the host's record) in an extra
    workgroup row of cycle k+1's first pass below the finest level (option mg_fold_finish, default 1; k_smooth2_march_v2) against the
    launch of its own (0): field bit for bit, history and cycle count equal, coarse-iteration count equal -- with the level below the
    top a two-sweep march (2049^2, 4097^2: carried; 1025^2 with a 129^2 CG coarse level: carried), the three-level kernel (1025^2:
    launched first) and the LDS-resident hierarchy (129^2: launched first); a loose tolerance ends the loop in the middle of cycles
    enqueued ahead.  Against the oracle as well where it finishes in seconds."""
    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    h = 1.0 / (n - 1)
    b = asf(splitmix64_uniform(n * n, 7).reshape((n, n), order="F"))
    gb = F.asdevice(b)
    opt = mg.MGOpt()
    opt.coarse_solve_size, opt.coarse_solver = css, (mg.jacobi if solver == 0 else mg.conjugate_gradient)
    import warnings
    got = {}
    for fold in (1, 0, 1):
        try:
            c.set_option("mg_fold_finish", fold)
            x = F.fzeros(n, n)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, gb, h, 0.0, tol, 12, bcs, opt=opt, return_history=True)
        finally:
            c.set_option("mg_fold_finish", 1)
        if fold in got:
            assert np.array_equal(F.tonumpy(x), got[fold][0]) and list(hist) == got[fold][1] and cit == got[fold][2]
        got[fold] = (F.tonumpy(x), list(hist), cit, r)
    assert np.array_equal(got[1][0], got[0][0])
    assert got[1][1] == got[0][1], (got[1][1], got[0][1])
    assert got[1][2] == got[0][2] and got[1][3] == got[0][3]
    if n <= 2049:
        xo = farr(n, n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r_o, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, 0.0, tol, 12, bcs, css, solver)
        assert len(hist_o) == len(got[1][1])
        assert np.allclose(got[1][1], hist_o, rtol=1e-10, atol=0)
        if solver == 0:
            assert np.array_equal(got[1][0], xo)
            assert got[1][2] == oracle.last_coarse_iters()


@pytest.mark.parametrize("n,bcs,cc", [(129, False, 0.0), (257, True, 0.0), (1025, False, 2.5), (2049, True, 0.0)])
def test_two_sweeps_from_the_zero_guess_as_one_pass(fpr, oracle, n, bcs, cc):
    """Option mg_zero_fuse (default 1): a level below the top starts from the zero guess (multigrid.jl:132), so its first pre-smoothing
    sweep (:124) is a pointwise function of its right-hand side and the second (:125) is taken straight from the right-hand side
    (mid_sweep_z2 in k_mid_down, mgs_sweep_z2 in k_mg_small, which then does not load u either) -- against sweep after sweep (0) and the
    oracle: fields bit for bit, histories and coarse-iteration counts equal."""
    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    h = 1.0 / (n - 1)
    b = asf(splitmix64_uniform(n * n, 17).reshape((n, n), order="F"))
    gb = F.asdevice(b)
    import warnings
    got = {}
    for zf in (1, 0):
        try:
            c.set_option("mg_zero_fuse", zf)
            x = F.asdevice(asf(splitmix64_uniform(n * n, 18).reshape((n, n), order="F")))    # a non-zero initial guess at the top level
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, gb, h, cc, 1e-9, 6, bcs, opt=mg.MGOpt(), return_history=True)
        finally:
            c.set_option("mg_zero_fuse", 1)
        got[zf] = (F.tonumpy(x), list(hist), cit)
    assert np.array_equal(got[1][0], got[0][0]) and got[1][1] == got[0][1] and got[1][2] == got[0][2]
    xo = asf(splitmix64_uniform(n * n, 18).reshape((n, n), order="F"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r_o, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, cc, 1e-9, 6, bcs, 5, 0)
    assert np.allclose(got[1][1], hist_o, rtol=1e-10, atol=0)
    assert np.array_equal(got[1][0], xo)
    assert got[1][2] == oracle.last_coarse_iters()


@pytest.mark.parametrize("n,bcs", [(513, False), (1025, True), (2049, False), (4097, True)])
def test_f_rms_from_the_first_pass(fpr, oracle, n, bcs):
    """Option mg_fold_fsq (default 1): sum(f.^2) for f_rms (multigrid.jl:53) is left as block partials by the solve's first pass over the
    finest grid (k_smooth2_march_v2<..., FSQ>: every point of f once, boundary rows and columns included) instead of a pass of its own over f
    (0).  f_rms agrees to 1e-13 (another summation order), with each other and with the oracle; histories, cycle counts and fields are
    equal (the threshold tol * f_rms moves by an ulp at most)."""
    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    h = 1.0 / (n - 1)
    b = asf(splitmix64_uniform(n * n, 19).reshape((n, n), order="F") - 0.25)
    gb = F.asdevice(b)
    import warnings
    got = {}
    for fs in (1, 0):
        try:
            c.set_option("mg_fold_fsq", fs)
            x = F.fzeros(n, n)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, gb, h, 0.0, 1e-7, 20, bcs, opt=mg.MGOpt(), return_history=True)
        finally:
            c.set_option("mg_fold_fsq", 1)
        got[fs] = (F.tonumpy(x), list(hist), cit, frms)
    assert abs(got[1][3] - got[0][3]) <= 1e-13 * got[0][3]
    frms_ref = math.sqrt(float((b.astype(np.longdouble) ** 2).sum()) / (n * n))
    assert abs(got[1][3] - frms_ref) <= 1e-13 * frms_ref
    assert got[1][1] == got[0][1] and got[1][2] == got[0][2]
    assert np.array_equal(got[1][0], got[0][0])


def test_config3_five_levels_4097(fpr, oracle):
    """BASELINE config 3 as named: 4097^2, 5 grids (l=8, coarse 257^2), 2+2 Jacobi smooths, multigrid_bench.jl
    protocol.  Jacobi coarse solver: 44 V-cycles / 226 160 coarse sweeps to tol 1e-6 (the coarse solve is capped at
    20*257 = 5140 sweeps and never converges), first three residuals and the field after three cycles equal to the
    oracle's; CG coarse solver: 7 V-cycles and 4397 coarse iterations in every launch form of cg!, the whole residual history
    to north_star's 1e-10 per entry and the field bit for bit against the oracle's."""
    F, mg = fpr, fpr.multigrid
    n = 4097
    h = 1.0 / (n - 1)
    b = asf(splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    gb = F.asdevice(b)
    opt = mg.MGOpt()
    opt.coarse_solve_size = 257
    import warnings
    # ---- Jacobi coarse solver: three cycles against the oracle (a full oracle solve would be 44 cycles of 5140 sweeps) ----
    opt.coarse_solver = mg.jacobi
    xo = farr(n, n)
    _, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, 0.0, 1e-6, 3, False, 257, mg.jacobi.value)
    cit_o = oracle.last_coarse_iters()
    x = F.fzeros(n, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # 3 cycles do not converge: the reference's @warn (multigrid.jl:78-80)
        _, hist3, frms, cit3 = mg.MGsolve_2DPoisson_(x, gb, h, 0.0, 1e-6, 3, False, opt=opt, return_history=True)
    assert len(hist3) == len(hist_o) == 3
    assert np.allclose(hist3, hist_o, rtol=1e-10, atol=0), (hist3, hist_o)
    assert abs(frms - frms_o) <= 1e-13 * frms_o
    assert cit3 == cit_o == 3 * 5140
    assert np.array_equal(F.tonumpy(x), xo)
    x.zero_()
    r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, gb, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
    assert len(hist) == 44 and r < 1e-6 * frms
    assert np.allclose(hist[:3], hist_o, rtol=1e-10, atol=0)
    assert cit == 44 * 5140 == 226160
    # ---- CG coarse solver: the whole solve against the oracle, in every launch form of cg! ----
    opt.coarse_solver = mg.conjugate_gradient
    xo = farr(n, n)
    _, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, 0.0, 1e-6, 100, False, 257, mg.conjugate_gradient.value)
    cit_o = oracle.last_coarse_iters()
    assert len(hist_o) == 7
    c = F.ctx()
    try:
        for form in (3, 2, 1):     # persistent kernel, two launches per iteration, three
            c.set_option("cg_fused", form)
            x = F.fzeros(n, n)
            r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, gb, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
            assert len(hist) == 7 and r < 1e-6 * frms and abs(frms - frms_o) <= 1e-13 * frms_o
            assert cit == cit_o, (form, cit, cit_o)        # the stop decision (krylov.jl:71) falls in the same iteration of every solve
            d = np.abs(np.asarray(hist) - np.asarray(hist_o))
            assert (d <= 1e-10 * np.asarray(hist_o)).all(), (form, (d / np.asarray(hist_o)).tolist())     # north_star's figure, every entry
            assert np.array_equal(F.tonumpy(x), xo)
    finally:
        c.set_option("cg_fused", 3)


@pytest.mark.parametrize("tol", [0.9, 0.7, 0.5, 0.4, 0.3, 0.25, 0.2, 0.15, 1e-9])
def test_coarse_jacobi_exit_inside_a_fused_group(fpr, oracle, tol):
    """The coarse Jacobi solve runs 8 fused sweeps per launch; when the exit test (multigrid.jl:152-155)
    fires inside a group the group is recomputed with exactly the right number of sweeps."""
    F, mg = fpr, fpr.multigrid
    shape = (129, 129)  # too large for the LDS-resident path: exercises the multi-sweep launches
    u0, f = rnd(shape, 31), rnd(shape, 32)
    f[0, :] = f[-1, :] = 0.0
    f[:, 0] = f[:, -1] = 0.0
    h = 1.0 / 128
    u_ref = u0.copy(order="F")
    r_ref = oracle.vcycle2d(u_ref, f, h, 0.0, tol, 129, 0, False)
    for patch in (1, 0):  # register-patch kernel (default) and the generic LDS-tile kernel
        F.ctx().set_option("mg_patch", patch)
        gu = F.asdevice(u0)
        r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), h, 0.0, tol, 129, mg.jacobi, mg.parallel_shmem, False)
        F.ctx().set_option("mg_patch", 1)
        assert abs(r - r_ref) <= 1e-12 * abs(r_ref)
        assert np.array_equal(F.tonumpy(gu), u_ref)


@pytest.mark.parametrize("fences", [0, 1])
@pytest.mark.parametrize("tol", [0.5, 0.2, 0.05, 0.02, 1e-9])
def test_coarse_jacobi_persistent_launches_equal_the_plain_ones(fpr, oracle, tol, fences):
    """k_jacobi_persist (up to 32 groups of 8 sweeps per launch, tiles handed from neighbour to neighbour, exit test behind the
    launch and the exact number of sweeps replayed from the launch's input) against one launch per 8 sweeps and against the
    oracle: a 257 x 129 coarse grid solved directly by Vcycle_2DPoisson! (:147-159) with exits in the first launch, in a later
    one, inside a group, and none at all (cap 20 * 257 sweeps)."""
    F, mg = fpr, fpr.multigrid
    shape = (257, 129)
    u0, f = rnd(shape, 61), rnd(shape, 62)
    f[0, :] = f[-1, :] = 0.0
    f[:, 0] = f[:, -1] = 0.0
    h = 1.0 / 128
    u_ref = u0.copy(order="F")
    r_ref = oracle.vcycle2d(u_ref, f, h, 0.0, tol, 257, 0, False)
    it_ref = oracle.last_coarse_iters()
    outs = []
    c = F.ctx()
    try:
        # k_jacobi_persist_tag (every cell a {value, tag} granule, no flags); with option handoff_fences = 1 (hand-offs inside the HIP memory
        # model only) the coarse solve runs as plain launches whatever mg_jacobi_persist says
        c.set_option("handoff_fences", fences)
        for persist in (1, 0):
            c.set_option("mg_jacobi_persist", persist)
            gu = F.asdevice(u0)
            r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), h, 0.0, tol, 257, mg.jacobi, mg.parallel_shmem, False)
            outs.append((r, F.tonumpy(gu)))
    finally:
        c.set_option("mg_jacobi_persist", 1)
        c.set_option("handoff_fences", 0)
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    assert abs(outs[0][0] - r_ref) <= 1e-12 * abs(r_ref)
    assert np.array_equal(outs[0][1], u_ref)
    assert it_ref > 0


@pytest.mark.parametrize("tol,abort_launch", [(1e-9, 1), (1e-9, 2), (1e-9, 4), (0.02, 1), (0.02, 3)])
def test_coarse_jacobi_persistent_launch_that_gives_up_is_resumed_by_the_plain_ones(fpr, oracle, tol, abort_launch):
    """A neighbour hand-off of k_jacobi_persist that times out (workgroups not resident together: a shared card) must not fail the
    solve: the launch that gave up never wrote its input, so the solve resumes there with one launch per 8 sweeps, the context
    stays off the persistent form and counts the event (option mg_jacobi_persist_timeouts), and the result is the oracle's bit for
    bit.  The time-out is injected (option mg_jacobi_persist_test_abort = n: the n-th launch of the solve finds the abort flag
    set) in the first launch, in a later one (whose input is one of the rotating work buffers) and behind the exit."""
    F, mg = fpr, fpr.multigrid
    shape = (257, 129)
    u0, f = rnd(shape, 61), rnd(shape, 62)
    f[0, :] = f[-1, :] = 0.0
    f[:, 0] = f[:, -1] = 0.0
    h = 1.0 / 128
    u_ref = u0.copy(order="F")
    it0 = oracle.last_coarse_iters()           # (the oracle's counter runs on across direct V-cycle calls)
    r_ref = oracle.vcycle2d(u_ref, f, h, 0.0, tol, 257, 0, False)
    it_ref = oracle.last_coarse_iters() - it0
    c = F.ctx()
    c.set_option("mg_jacobi_persist", 1)
    before = c.L.fpr_get_option(c.h, b"mg_jacobi_persist_timeouts")
    launches_needed = -(-it_ref // 256)
    try:
        c.set_option("mg_jacobi_persist_test_abort", abort_launch)
        gu = F.asdevice(u0)
        r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), h, 0.0, tol, 257, mg.jacobi, mg.parallel_shmem, False)
        it = c.L.fpr_last_coarse_iters(c.h)
        c.set_option("mg_jacobi_persist_test_abort", 0)
        after = c.L.fpr_get_option(c.h, b"mg_jacobi_persist_timeouts")
        # a second solve on the same context: the plain form now (the switch is sticky), same result
        gu2 = F.asdevice(u0)
        r2 = mg.Vcycle_2DPoisson_(gu2, F.asdevice(f), h, 0.0, tol, 257, mg.jacobi, mg.parallel_shmem, False)
    finally:
        c.set_option("mg_jacobi_persist_test_abort", 0)
        c.set_option("mg_jacobi_persist", 1)           # lifts the switch again
    assert abs(r - r_ref) <= 1e-12 * abs(r_ref) and r2 == r
    assert np.array_equal(F.tonumpy(gu), u_ref) and np.array_equal(F.tonumpy(gu2), u_ref)
    assert it == it_ref
    # the injected launch exists only if the solve got that far (the host polls after 1, 2, 4, ... launches)
    if abort_launch <= launches_needed:
        assert after == before + 1
    else:
        assert after in (before, before + 1)


def torch_isnan(a):
    import torch

    return torch.isnan(a)


def test_provided_arena_buffers_change_nothing(fpr):
    """fpr_mg_arena_provide (prealloc_dict, multigrid.jl:25-38, 49-51): the finest level's two ping-pong partners from the caller,
    then the library's own again -- the solve is the same bit for bit either way, and the caller's buffers are really used."""
    F, mg = fpr, fpr.multigrid
    n = 513
    h = 1.0 / (n - 1)
    b = F.asdevice(rnd((n, n), 71))
    outs = []
    t1, t2 = F.fzeros(n, n), F.fzeros(n, n)
    t1.fill_(7.0); t2.fill_(7.0)
    for provided in (False, True, False):
        mg.provide_arena_(n, n, t1 if provided else None, t2 if provided else None)
        if not provided:
            t1.fill_(7.0); t2.fill_(7.0)
        x = F.fzeros(n, n)
        r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-8, 50, False, return_history=True)
        outs.append((r, tuple(hist), cit, F.tonumpy(x)))
        touched = bool((t1 != 7.0).any()) or bool((t2 != 7.0).any())
        assert touched == provided
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[1] == outs[0][1] and o[2] == outs[0][2] and np.array_equal(o[3], outs[0][3])
    with pytest.raises(F.FprError):
        F.ctx().call("fpr_mg_arena_provide", 2, 2, None, None)     # grids are at least 3 x 3
    # the three arrays of the first coarse level (fpr_mg_arena_provide_coarse), filled with NaN first: nothing is read before it is written
    nc = 1 + (n - 1) // 2
    cs = [F.fzeros(nc, nc) for _ in range(3)]
    for provided in (True, False):
        for a in cs:
            a.fill_(float("nan"))
        mg.provide_arena_coarse_(n, n, *(cs if provided else (None, None, None)))
        x = F.fzeros(n, n)
        r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-8, 50, False, return_history=True)
        assert r == outs[0][0] and tuple(hist) == outs[0][1] and cit == outs[0][2] and np.array_equal(F.tonumpy(x), outs[0][3])
        touched = any(bool((~torch_isnan(a)).any()) for a in cs)
        assert touched == provided
    with pytest.raises(F.FprError):
        mg.provide_arena_coarse_(n, n, cs[0], cs[1], None)         # all three or none
    with pytest.raises(F.FprError):
        mg.provide_arena_coarse_(n, n, cs[0], cs[0], cs[1])        # distinct


def test_multisweep_and_single_sweep_paths_agree(fpr):
    """Temporal blocking (mg_multi) and the LDS-resident coarse hierarchy (mg_small) change no bit."""
    F, mg = fpr, fpr.multigrid
    shape = (513, 257)
    u0, f = rnd(shape, 41), rnd(shape, 42)
    outs = []
    for multi, small in ((1, 1), (0, 1), (2, 1), (1, 0), (0, 0)):
        F.ctx().set_option("mg_multi", multi)
        F.ctx().set_option("mg_small", small)
        gu = F.asdevice(u0)
        r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), 1.0 / 256, 0.3, 1e-7, 5, mg.jacobi, mg.parallel, True)
        outs.append((r, F.tonumpy(gu)))
    F.ctx().set_option("mg_multi", 1)
    F.ctx().set_option("mg_small", 1)
    F.ctx().set_option("mg_fuse_prolong", 0)
    gu = F.asdevice(u0)
    r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), 1.0 / 256, 0.3, 1e-7, 5, mg.jacobi, mg.parallel, True)
    outs.append((r, F.tonumpy(gu)))
    F.ctx().set_option("mg_fuse_prolong", 1)
    F.ctx().set_option("mg_fuse_restrict", 0)
    gu = F.asdevice(u0)
    r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), 1.0 / 256, 0.3, 1e-7, 5, mg.jacobi, mg.parallel, True)
    outs.append((r, F.tonumpy(gu)))
    F.ctx().set_option("mg_fuse_restrict", 1)
    # the zero coarse guess read from memory like any field (default: not read at all, its loads are dropped by the range check)
    for fuse_r in (1, 0):
        F.ctx().set_option("mg_zero_guess", 0)
        F.ctx().set_option("mg_fuse_restrict", fuse_r)
        gu = F.asdevice(u0)
        r = mg.Vcycle_2DPoisson_(gu, F.asdevice(f), 1.0 / 256, 0.3, 1e-7, 5, mg.jacobi, mg.parallel, True)
        outs.append((r, F.tonumpy(gu)))
    F.ctx().set_option("mg_zero_guess", 1)
    F.ctx().set_option("mg_fuse_restrict", 1)
    for r, u in outs[1:]:
        assert np.array_equal(u, outs[0][1])
        assert abs(r - outs[0][0]) <= 1e-13 * abs(outs[0][0])


@pytest.mark.parametrize("bc", [False, True])
@pytest.mark.parametrize("shape,css", [((5, 5), 5), ((9, 9), 5), ((9, 33), 5), ((33, 9), 3), ((17, 5), 5), ((65, 17), 9),
                                        ((129, 33), 5), ((3, 3), 3), ((1025, 257), 5)], ids=str)
def test_small_and_rectangular_hierarchies(fpr, oracle, shape, css, bc):
    """Tiny, rectangular (lambda_x != lambda_y, multigrid.jl:27-28) and LDS-resident hierarchies: full solve parity."""
    F, mg = fpr, fpr.multigrid
    u0, f = rnd(shape, 51), rnd(shape, 52)
    h = 1.0 / (min(shape) - 1)
    c = 0.0 if not bc else 2.5
    opt = mg.MGOpt()
    opt.coarse_solve_size = css
    u_ref = u0.copy(order="F")
    r_ref, hist_ref, frms_ref = oracle.mgsolve2d(u_ref, f, h, c, 1e-9, 6, bc, css, 0)
    gu = F.asdevice(u0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r, hist, frms, cit = mg.MGsolve_2DPoisson_(gu, F.asdevice(f), h, c, 1e-9, 6, bc, opt=opt, return_history=True)
    assert len(hist) == len(hist_ref)
    assert np.allclose(hist, hist_ref, rtol=1e-10, atol=0)
    assert np.array_equal(F.tonumpy(gu), u_ref)
    assert cit == oracle.last_coarse_iters()


def test_last_cycle_guessed_from_the_previous_solve_changes_nothing(fpr):
    """A time stepper solves the same systems step after step (part2.jl:187,221,226): the context remembers how many V-cycles the
    last solve with the same arrays took and uses it as a second opinion on which cycle will be the last (mg_seam_history).  A
    guess decides which launches are enqueued, never what they compute: solves on the SAME device arrays with tolerances that
    make the remembered count right, too small and too large, against the plain loop (mg_ahead = 0) -- field bit for bit, same
    cycle count and coarse iterations, history to summation order."""
    import warnings

    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    shape = (1025, 513)
    h = 1.0 / (shape[0] - 1)
    f = F.asdevice(rnd(shape, 15))
    u, u_ref = F.fzeros(*shape), F.fzeros(*shape)
    opt = mg.MGOpt()
    counts = []
    for tol in (1e-6, 1e-6, 1e-9, 1e-9, 1e-4, 1e-4, 1e-7, 1e-6, 1e-12, 1e-6):
        res = []
        for arr, ahead in ((u, 1), (u_ref, 0)):
            arr.zero_()
            c.set_option("mg_ahead", ahead)
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res.append(mg.MGsolve_2DPoisson_(arr, f, h, 0.0, tol, 12, False, opt=opt, return_history=True))
            finally:
                c.set_option("mg_ahead", 1)
        (r, hist, frms, cit), (r0, h0, f0, c0) = res
        # (f_rms: sum(f.^2) is left by the first pass over the finest grid in the loop with cycles ahead, by a pass of its own in the plain one --
        #  two summation orders, option mg_fold_fsq)
        assert len(hist) == len(h0) and abs(frms - f0) <= 1e-13 * f0 and cit == c0 and r == pytest.approx(r0, rel=1e-12), tol
        assert np.allclose(hist, h0, rtol=1e-12, atol=0.0)
        assert np.array_equal(F.tonumpy(u), F.tonumpy(u_ref)), tol
        counts.append(len(hist))
    assert len(set(counts)) >= 4 and counts[-2] == 12      # the remembered count was wrong in both directions; 1e-12 hits niters


@pytest.mark.parametrize("shape,css,bcs,tol,niters", [
    ((257, 257), 5, False, 1e-6, 100),      # converges after several cycles
    ((257, 257), 5, False, 1e-30, 4),       # never converges: stops at niters
    ((513, 129), 9, True, 1e-5, 100),       # Neumann columns + Dirichlet rows re-applied every cycle (:60-62)
    ((1025, 1025), 5, False, 1e-1, 100),    # converges in the very first cycles: what was enqueued ahead must not run
    ((2049, 2049), 17, True, 1e-7, 50),
    ((2049, 1025), 17, False, 1e-9, 50),
    ((129, 129), 5, False, 1e-6, 100),      # one marched level above the LDS-resident sub-hierarchy
    ((1025, 65), 5, True, 1e-8, 100),       # a single row of seam chunks, 19 strips
    ((65, 1025), 3, False, 1e-8, 100),      # two strips, 3x33 coarsest grid (one interior row inside the DPP row solve)
    ((257, 129), 9, False, 1e-10, 30),
    ((4097, 4097), 5, False, 1e-6, 100),    # BASELINE config 3
], ids=str)
def test_cycles_enqueued_ahead_equal_the_plain_loop(fpr, shape, css, bcs, tol, niters):
    """MGsolve_2DPoisson! (multigrid.jl:41-84) with its exit test (:70) taken on the device, 0..3 cycles enqueued before
    the host has seen the previous norm (mg_ahead) and -- without boundary conditions -- consecutive cycles sharing one
    pass over the finest grid (k_seam_march: post-smoothing of cycle k + pre-smoothing, residual, injection of cycle k+1),
    against the plain loop that waits for every norm: the field is identical bit for bit, the cycle count and the
    coarse-solver iteration count are identical, the residual history agrees to summation order (bit for bit where the
    same kernels produce it).  Covers loops that end in their first cycles, at niters, on a seam (replay of the plain
    post-smoothing pass), on a predicted last cycle, after a wrong prediction, and a second solve right behind the first."""
    import warnings

    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    f = rnd(shape, 5)
    if bcs:
        f = f - 0.5
    opt = mg.MGOpt()
    opt.coarse_solve_size = css
    h = 1.0 / (shape[0] - 1)
    outs = []
    variants = [(0, 0, 1), (1, 0, 1), (3, 0, 1), (0, 1, 1), (1, 1, 1), (3, 1, 1), (1, 1, 0), (2, 1, 2)]
    try:
        for ahead, seam, predict in variants:
            c.set_option("mg_ahead", ahead)
            c.set_option("mg_seam", seam)
            c.set_option("mg_seam_predict", predict)
            for rep in range(2):    # the second solve starts while the skipped launches of the first are still queued
                u = F.asdevice(rnd(shape, 6) if bcs else np.zeros(shape))
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    r, hist, frms, cit = mg.MGsolve_2DPoisson_(u, F.asdevice(f), h, 0.0, tol, niters, bcs, opt=opt, return_history=True)
            outs.append((r, hist, frms, cit, F.tonumpy(u)))
    finally:
        c.set_option("mg_ahead", 1)
        c.set_option("mg_seam", 1)
        c.set_option("mg_seam_predict", 1)
    r0, h0, f0, c0, u0 = outs[0]
    assert 1 <= len(h0) <= niters and np.isfinite(u0).all()
    if tol == 1e-30:
        assert len(h0) == niters
    for (ahead, seam, predict), (r, hist, frms, cit, u) in zip(variants[1:], outs[1:]):
        assert len(hist) == len(h0) and abs(frms - f0) <= 1e-13 * f0 and cit == c0, (ahead, seam, predict)
        if seam:
            assert np.allclose(hist, h0, rtol=1e-12, atol=0.0)
        else:
            assert np.array_equal(hist, h0) and r == r0
        assert np.array_equal(u, u0), (ahead, seam, predict)


@pytest.mark.parametrize("shape,css,bcs", [
    ((1025, 1025), 5, False), ((1025, 1025), 5, True), ((513, 513), 5, False), ((2049, 2049), 17, True),
    ((1025, 513), 9, False), ((513, 257), 5, True), ((2049, 1025), 5, False), ((4097, 4097), 5, False),
    ((513, 1025), 3, True), ((1025, 129), 5, False),
], ids=str)
def test_three_levels_in_two_launches_equal_the_level_by_level_path(fpr, shape, css, bcs):
    """k_mid_down / k_mid_up (the pre-smoothing passes of the three levels above the LDS-resident sub-hierarchy in one
    launch, their post-smoothing passes in another; option mg_mid) against one marching pass per level and direction:
    a single V-cycle (multigrid.jl:91-170) and a whole solve (:41-84) give the same field bit for bit, the same norms,
    histories and coarse-solver iteration counts -- square and oblong grids, with and without boundary conditions."""
    import warnings

    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    f = rnd(shape, 15) - (0.5 if bcs else 0.0)
    u0 = rnd(shape, 16)
    opt = mg.MGOpt()
    opt.coarse_solve_size = css
    h = 1.0 / (shape[0] - 1)
    outs = []
    try:
        for mid in (0, 1):
            c.set_option("mg_mid", mid)
            u = F.asdevice(u0)
            r1 = mg.Vcycle_2DPoisson_(u, F.asdevice(f), h, 0.3, 1e-6, css, mg.jacobi, mg.parallel, bcs)
            v1 = F.tonumpy(u)
            u = F.asdevice(u0)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r, hist, frms, cit = mg.MGsolve_2DPoisson_(u, F.asdevice(f), h, 0.0, 1e-8, 12, bcs, opt=opt, return_history=True)
            outs.append((r1, v1, r, hist, cit, F.tonumpy(u)))
    finally:
        c.set_option("mg_mid", 1)
    a, b = outs
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and np.array_equal(a[3], b[3]) and a[4] == b[4]
    assert np.array_equal(a[5], b[5]) and np.isfinite(b[5]).all()


def test_stream_mode_equals_plain_loop_over_many_small_problems(fpr):
    """A sweep over shapes, coarse_solve_size, boundary conditions, niters (including 1 and 2) and tolerances: fpr_mgsolve2d
    in its default mode (device-side exit test, cycles ahead, seam pass, mid-level kernels, DPP-row coarse solve where each
    applies) against the plain loop with the level-by-level kernels -- same field bit for bit, same cycle and coarse-iteration
    counts, histories equal to summation order.  Errors the reference raises must come out the same in both modes."""
    import itertools
    import warnings

    F, mg = fpr, fpr.multigrid
    c = F.ctx()
    shapes = [(65, 65), (129, 65), (65, 129), (257, 33), (129, 129), (513, 65), (257, 257), (513, 129), (1025, 257), (33, 33), (17, 257)]
    combos = list(itertools.product(shapes, (3, 5, 9, 17), (False, True), (1, 2, 3, 9), (1e-2, 1e-7, 1e-13)))
    rng = np.random.default_rng(11)
    picks = [combos[i] for i in sorted(rng.choice(len(combos), size=90, replace=False))]
    plain = {"mg_ahead": 0, "mg_seam": 0, "mg_mid": 0, "mg_small": 0}
    checked = 0
    for shape, css, bcs, niters, tol in picks:
        f = rnd(shape, 31) - (0.5 if bcs else 0.0)
        u0 = rnd(shape, 32) * 0.1
        opt = mg.MGOpt()
        opt.coarse_solve_size = css
        h = 1.0 / (shape[0] - 1)
        res = []
        for mode in (plain, {}):
            try:
                for k in plain:
                    c.set_option(k, mode.get(k, 1))
                u = F.asdevice(u0)
                try:
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        r, hist, frms, cit = mg.MGsolve_2DPoisson_(u, F.asdevice(f), h, 0.7, tol, niters, bcs, opt=opt, return_history=True)
                    res.append(("ok", r, hist, frms, cit, F.tonumpy(u)))
                except (AssertionError, RuntimeError) as e:
                    res.append(("err", type(e).__name__))
            finally:
                for k in plain:
                    c.set_option(k, 1)
        a, b = res
        assert a[0] == b[0], (shape, css, bcs, niters, tol)
        if a[0] == "err":
            assert a[1] == b[1]
            continue
        checked += 1
        # (a[3]: f_rms -- two summation orders: the first pass over the finest grid leaves sum(f.^2) in stream mode, option mg_fold_fsq)
        assert len(a[2]) == len(b[2]) and a[4] == b[4] and abs(a[3] - b[3]) <= 1e-13 * abs(b[3]), (shape, css, bcs, niters, tol)
        assert np.allclose(a[2], b[2], rtol=1e-12, atol=0.0), (shape, css, bcs, niters, tol)
        assert np.array_equal(a[5], b[5]), (shape, css, bcs, niters, tol)
    assert checked >= 60
