"""GPU parity tests for the NEXT row 8f-1 (Navier-Stokes step around the V-cycle), pinned by the
reference's FORTRAN fixtures (test/part2.jl)."""
import math
import os

import numpy as np
import pytest

from fixtures_io import GOLDEN, load_bin
from oracle.oracle import farr

pytestmark = pytest.mark.gpu


def test_pointwise_kernels_bit_exact(fpr, oracle):
    F, p2 = fpr, fpr.part2
    S, T, W = load_bin("S.bin"), load_bin("Tinit.bin"), load_bin("Winit.bin")
    nx, ny = S.shape
    h = 1.0 / (ny - 1.0)
    vx, vy = farr(nx, ny), farr(nx, ny)
    oracle.compute_velocity(S, h, h, vx, vy)
    gvx, gvy = F.fzeros(nx, ny), F.fzeros(nx, ny)
    p2.compute_velocity_(F.asdevice(S), h, h, gvx, gvy)
    assert np.array_equal(F.tonumpy(gvx), vx) and np.array_equal(F.tonumpy(gvy), vy)
    for name, args_o, call in (
        ("Ra", (1e6, h, T), lambda o: p2.compute_Ra_dTdx_(1e6, h, F.asdevice(T), o)),
        ("dif", (W, h, h, 1e-3), lambda o: p2.compute_diffusion2d_(F.asdevice(W), h, h, 1e-3, o)),
        ("ax", (T, h, vx), lambda o: p2.compute_advection2d_x_(F.asdevice(T), h, gvx, o)),
        ("ay", (W, h, vy), lambda o: p2.compute_advection2d_y_(F.asdevice(W), h, gvy, o)),
    ):
        ref = farr(nx, ny)
        {"Ra": oracle.compute_Ra_dTdx, "dif": oracle.compute_diffusion2d, "ax": oracle.compute_advection2d_x,
         "ay": oracle.compute_advection2d_y}[name](*args_o, ref)
        out = F.fzeros(nx, ny)
        call(out)
        assert np.array_equal(F.tonumpy(out), ref), name


def test_one_explicit_step_matches_fortran(fpr):
    """test/part2.jl:4-38: testmode, 257x65, MG tol 1e-12, W from Winit.bin; T, W, S interior within 1e-8."""
    F, p2 = fpr, fpr.part2
    opt = p2.SimIn_t()
    opt.nx, opt.ny = 257, 65
    opt.tol = 1.0e-12
    opt.W_init_strategy = p2.W_from_file
    opt.W_init_file = os.path.join(GOLDEN, "fortran", "Winit.bin")
    out = p2.navier_stokes_2D(opt=opt, verbose=False, testmode=True)
    for name in ("T", "W", "S"):
        ref = load_bin(name + ".bin")
        got = getattr(out, name)
        assert got.shape == ref.shape
        assert np.abs(got[1:-1, 1:-1] - ref[1:-1, 1:-1]).max() < 1e-8, name
    assert out.dt_last == 3.662109375e-05


def _oracle_ns_steps(oracle, T, W, nx, ny, Ra, Pr, k, beta, tol, niters, nsteps, trace=None):
    """part2.jl:181-250 restated with the oracle's kernels (semi-implicit and explicit branches)."""
    h = 1.0 / (ny - 1.0)
    dt_dif = 0.15 * h * h / max(k, Pr)
    S = farr(nx, ny)
    out = []
    for _ in range(nsteps):
        rec = {}
        rec["S"] = oracle.mgsolve2d(S, W, h, 0.0, tol, niters)
        vx, vy = farr(nx, ny), farr(nx, ny)
        oracle.compute_velocity(S, h, h, vx, vy)
        v = np.sqrt(vx * vx + vy * vy)
        if v.max() == 0:
            dt = dt_dif
        else:
            dt_adv = 0.4 * min(h / np.abs(vx).max(), h / np.abs(vy).max())
            dt = dt_adv if beta >= 0.5 else min(dt_dif, dt_adv)
        oracle.bc2d(T)
        R, dT2, dW2 = farr(nx, ny), farr(nx, ny), farr(nx, ny)
        oracle.compute_Ra_dTdx(Ra, h, T, R)
        oracle.compute_diffusion2d(T, h, h, k, dT2)
        oracle.compute_diffusion2d(W, h, h, Pr, dW2)
        dTx, dTy, dWx, dWy = (farr(nx, ny) for _ in range(4))
        oracle.compute_advection2d_x(T, h, vx, dTx)
        oracle.compute_advection2d_y(T, h, vy, dTy)
        oracle.compute_advection2d_x(W, h, vx, dWx)
        oracle.compute_advection2d_y(W, h, vy, dWy)
        if beta > 0.0:
            c = 1.0 / (beta * dt)
            T_rhs = np.asfortranarray(-c * (T + dt * ((1.0 - beta) * dT2 - dTx - dTy)))
            rec["T"] = oracle.mgsolve2d(T, T_rhs, h, c, tol, niters, True)
            c = c / Pr
            W_rhs = np.asfortranarray(-c * (W + dt * ((1.0 - beta) * dW2 - dWx - dWy - Pr * R)))
            rec["W"] = oracle.mgsolve2d(W, W_rhs, h, c, tol, niters, False)
        else:
            T[:] = T + dt * (dT2 - dTx - dTy)
            W[:] = W + dt * (dW2 - dWx - dWy - Pr * R)
        out.append(dt)
        if trace is not None:
            trace.append(rec)
    return S, out


@pytest.mark.parametrize("beta", [0.0, 0.5])
def test_driver_steps_match_oracle(fpr, oracle, beta):
    """Three steps of navier_stokes_2D (explicit and semi-implicit with its apply_BCs / c>0 MG solves) vs the
    same loop restated with the oracle: identical time steps, fields to 1e-10."""
    import warnings

    F, p2 = fpr, fpr.part2
    opt = p2.SimIn_t()
    opt.nx, opt.ny, opt.beta, opt.tol, opt.Pr, opt.niters, opt.ttot = 129, 33, beta, 1.0e-7, 1.0e-1, 50, 1e9
    opt.W_init_strategy = p2.random
    nx, ny = opt.nx, opt.ny
    h = 1.0 / (ny - 1.0)
    width = (nx - 1.0) / (ny - 1.0)
    T = np.asfortranarray(np.repeat((0.5 * (1.0 + np.cos((3.0 * np.pi * np.arange(nx) * h) / width)))[:, None], ny, axis=1))
    W = np.asfortranarray(p2.splitmix64_uniform(nx * ny, opt.seed).reshape((nx, ny), order="F"))
    S_o, dts = _oracle_ns_steps(oracle, T, W, nx, ny, opt.Ra, opt.Pr, opt.k, beta, opt.tol, opt.niters, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=3)
    assert out.steps == 3 and abs(out.dt_last - dts[-1]) <= 1e-12 * dts[-1]
    for name, ref in (("T", T), ("W", W), ("S", S_o)):
        got = getattr(out, name)
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.abs(got - ref).max() <= 1e-10 * scale, (name, np.abs(got - ref).max(), scale)


def _config5_inputs(p2, nx, ny, seed=1):
    h = 1.0 / (ny - 1.0)
    width = (nx - 1.0) / (ny - 1.0)
    T = np.asfortranarray(np.repeat((0.5 * (1.0 + np.cos((3.0 * np.pi * np.arange(nx) * h) / width)))[:, None], ny, axis=1))
    W = np.asfortranarray(p2.splitmix64_uniform(nx * ny, seed).reshape((nx, ny), order="F"))
    return T, W


def _config5_opt(p2, nx, ny, niters):
    """BASELINE config 5 as the reference can run it: buoyancy-driven convection (the reference has
    no lid-driven cavity), semi-implicit beta = 0.5, MG tol 1e-7 (part2_semi_implicit_vs_explicit_experiments.jl:35-44),
    Pr = 1, Ra = 1e6, cosine T, counter-based random W."""
    opt = p2.SimIn_t()
    opt.nx, opt.ny, opt.beta, opt.tol, opt.Pr, opt.niters, opt.ttot = nx, ny, 0.5, 1.0e-7, 1.0, niters, 1e9
    opt.W_init_strategy = p2.random
    return opt


def test_config5_semi_implicit_513x129_three_steps_match_oracle(fpr, oracle):
    """Config 5 at a size the oracle runs whole: three semi-implicit steps (3 MG solves each, the T solve with
    apply_BCs and c > 0 running into niters = 50 as in the reference) -- identical time steps, residual histories of
    all nine solves to 1e-10, fields to 1e-10."""
    import warnings

    F, p2 = fpr, fpr.part2
    nx, ny = 513, 129
    opt = _config5_opt(p2, nx, ny, 50)
    T, W = _config5_inputs(p2, nx, ny)
    tr_o, tr = [], []
    S_o, dts = _oracle_ns_steps(oracle, T, W, nx, ny, opt.Ra, opt.Pr, opt.k, opt.beta, opt.tol, opt.niters, 3, trace=tr_o)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=3, trace=tr)
    assert out.steps == 3 and len(tr) == 3
    for step in range(3):
        assert abs(tr[step]["dt"] - dts[step]) <= 1e-12 * dts[step]
        for name in ("S", "T", "W"):
            r_o, hist_o, frms_o = tr_o[step][name]
            g = tr[step][name]
            assert len(g["history"]) == len(hist_o), (step, name, len(g["history"]), len(hist_o))
            assert np.allclose(g["history"], hist_o, rtol=1e-10, atol=0), (step, name)
    # the first T solve does not converge in niters (the reference emits its @warn); with the tiny time
    # steps that follow c = 1/(beta dt) dominates the operator and the later ones converge in a few cycles
    assert len(tr[0]["T"]["history"]) == 50 and len(tr[1]["T"]["history"]) < 50
    for name, ref in (("T", T), ("W", W), ("S", S_o)):
        got = getattr(out, name)
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-10 * scale, (name, np.abs(got - ref).max(), scale)


def test_config5_semi_implicit_2049sq(fpr, oracle):
    """BASELINE config 5 at its full size (2049^2 = nearest valid grid to 2048^2), the reference's niters = 50.
    The first step against the oracle: residual histories of the S solve (converges), of the T solve (apply_BCs,
    c > 0: runs into niters, the reference's @warn path) and of the W solve to 1e-10, the time step, and the fields
    after the step to 1e-10.  Steps 2 and 3 (the oracle needs ~20 s per step at this size): size-independent
    properties -- Dirichlet columns exact, fields finite, time steps positive, S and W solves converged."""
    import warnings

    F, p2 = fpr, fpr.part2
    nx = ny = 2049
    T, W = _config5_inputs(p2, nx, ny)
    opt = _config5_opt(p2, nx, ny, 50)
    tr_o, tr, snap = [], [], {}
    S_o, dts = _oracle_ns_steps(oracle, T, W, nx, ny, opt.Ra, opt.Pr, opt.k, opt.beta, opt.tol, opt.niters, 1, trace=tr_o)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out1 = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=1, trace=tr)
    assert abs(tr[0]["dt"] - dts[0]) <= 1e-12 * dts[0]
    for name in ("S", "T", "W"):
        r_o, hist_o, frms_o = tr_o[0][name]
        g = tr[0][name]
        assert len(g["history"]) == len(hist_o), (name, len(g["history"]), len(hist_o))
        assert np.allclose(g["history"], hist_o, rtol=1e-10, atol=0), (name, g["history"], hist_o)
        assert abs(g["f_rms"] - frms_o) <= 1e-12 * frms_o
    assert len(tr[0]["T"]["history"]) == 50 and len(tr[0]["S"]["history"]) < 50 and len(tr[0]["W"]["history"]) < 50
    for name, ref in (("T", T), ("W", W), ("S", S_o)):
        got = getattr(out1, name)
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-10 * scale, (name, np.abs(got - ref).max(), scale)
    tr = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=3, trace=tr)
    assert out.steps == 3
    assert tr[0]["dt"] == out1.dt_last                      # deterministic: the same first step again
    assert len(tr[0]["T"]["history"]) == 50 and tr[0]["T"]["r_rms"] > opt.tol * tr[0]["T"]["f_rms"]   # @warn path
    for rec in tr:
        assert rec["dt"] > 0 and math.isfinite(rec["dt"])
        assert rec["S"]["r_rms"] < opt.tol * rec["S"]["f_rms"] and rec["W"]["r_rms"] < opt.tol * rec["W"]["f_rms"]
    for name in ("T", "W", "S"):
        assert np.isfinite(getattr(out, name)).all(), name
    # Dirichlet columns of T (part2_utils.jl:28-29) are re-imposed before every V-cycle and never touched by it
    assert np.all(out.T[:, 0] == 1.0) and np.all(out.T[:, -1] == 0.0)
    assert np.all(out.S[0, :] == 0.0) and np.all(out.S[:, 0] == 0.0)   # S keeps its zero boundary (x = 0 start)


@pytest.mark.parametrize("beta", [0.0, 0.5, 1.0])
def test_fused_step_equals_kernel_by_kernel_step(fpr, oracle, beta):
    """The two-pass step (fpr_ns_velocity_max2d + fpr_ns_rhs2d) against the reference's kernel-by-kernel sequence
    (part2.jl:190-230 on the seven pointwise kernels): time steps and fields bit for bit over three steps, for the
    explicit, the semi-implicit and the fully implicit (no diffusion terms, :205) branch."""
    import warnings

    F, p2 = fpr, fpr.part2
    outs = []
    for fused in (True, False):
        opt = p2.SimIn_t()
        opt.nx, opt.ny, opt.beta, opt.tol, opt.Pr, opt.niters, opt.ttot = 257, 65, beta, 1.0e-7, 1.0e-1, 20, 1e9
        opt.W_init_strategy = p2.random
        tr = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=3, trace=tr, fused=fused)
        outs.append((out, [r["dt"] for r in tr]))
    (a, dta), (b, dtb) = outs
    assert dta == dtb
    for name in ("T", "W", "S"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    # pass 1 alone: maxima and (optional) velocity arrays against the oracle
    S = load_bin("S.bin")
    nx, ny = S.shape
    h = 1.0 / (ny - 1.0)
    vx, vy = farr(nx, ny), farr(nx, ny)
    oracle.compute_velocity(S, h, h, vx, vy)
    gvx, gvy = F.fzeros(nx, ny), F.fzeros(nx, ny)
    m = p2.velocity_and_maxima(F.asdevice(S), h, h, gvx, gvy)
    assert np.array_equal(F.tonumpy(gvx), vx) and np.array_equal(F.tonumpy(gvy), vy)
    assert m == (np.sqrt(vx * vx + vy * vy).max(), np.abs(vx).max(), np.abs(vy).max())
    assert p2.velocity_and_maxima(F.asdevice(S), h, h) == m


@pytest.mark.parametrize("shape", [(513, 129), (1025, 1025)], ids=str)
def test_t_and_w_solves_side_by_side_equal_the_sequence(fpr, shape):
    """The T solve (part2.jl:221) and the W solve (:226) of a time step do not depend on each other: run side by side (W on a
    second context from a worker thread, ordered against the default stream by events) they leave the same T, W, S and time
    step as one after the other -- bit for bit, over several steps, so that each step consumes what the previous one's two
    solves produced.  The same for the loop body as ONE library call (fpr_ns_step2d: the default), twice."""
    import warnings

    p2 = fpr.part2
    outs = []
    for conc, native, pipeline in ((False, False, 1), (True, False, 1), (True, True, 1), (True, True, 1), (True, True, 0)):
        fpr.ctx().set_option("ns_pipeline", pipeline)     # 0: the library's loop in the reference's order (S solve at the top of a step)
        opt = p2.SimIn_t()
        opt.nx, opt.ny, opt.beta, opt.tol, opt.Pr, opt.niters, opt.ttot = shape[0], shape[1], 0.5, 1.0e-7, 1.0, 30, 1e9
        opt.W_init_strategy = p2.random
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outs.append(p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=5, fused=True, concurrent_solves=conc, native_step=native))
    fpr.ctx().set_option("ns_pipeline", 1)
    a = outs[0]
    for b in outs[1:]:
        assert a.dt_last == b.dt_last and a.steps == b.steps
        for name in ("T", "W", "S"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.isfinite(a.T).all() and np.isfinite(a.W).all()


def test_ns_steps_inside_the_library_one_by_one_and_in_one_call(fpr):
    """fpr_ns_step2d (one loop body of part2.jl:186-226 per call) twice against fpr_ns_run2d (the loop :182 around it) for two steps:
    same T, W, S, dt and simulated time, bit for bit; run2d stops at ttot like `while sim_time < ttot`."""
    import ctypes as C

    F = fpr
    p2, mg = F.part2, F.multigrid
    fptr = F._lib.fptr

    nx, ny = 129, 65
    h = 1.0 / (ny - 1.0)
    opt = p2.SimIn_t()
    opt.nx, opt.ny, opt.beta, opt.tol, opt.Pr, opt.niters = nx, ny, 0.5, 1.0e-7, 1.0, 30
    dt_dif = (opt.a_dif * h ** 2) / max(opt.k, opt.Pr)
    c, c2 = F.ctx(), F.second_ctx()

    def fields():
        A = {n: F.fzeros(nx, ny) for n in "S T T_rhs W W_rhs".split()}
        p2.init_array_(A["T"], opt.T_init_strategy, h, (nx - 1.0) / (ny - 1.0), opt)
        p2.init_array_(A["W"], p2.random, h, (nx - 1.0) / (ny - 1.0), opt)
        return A

    args = (nx, ny, opt.Ra, opt.Pr, opt.k, opt.beta, opt.a_adv, dt_dif, opt.tol, int(opt.niters), 5, 0)
    A = fields()
    dts, t, unconverged = [], 0.0, 0
    for _ in range(2):
        dt, info = C.c_double(0.0), (C.c_int * 6)()
        c.call("fpr_ns_step2d", c2.h, *(fptr(A[n], 2) for n in ("S", "T", "W", "T_rhs", "W_rhs")), *args, C.byref(dt), info)
        assert all(1 <= v <= opt.niters for v in info[0:3])
        unconverged += sum(1 for v in info[3:6] if not v)     # (the first T solve hits niters, as in the reference)
        dts.append(dt.value)
        t += dt.value
    B = fields()
    t_c, n_c, dt_c, bad = C.c_double(0.0), C.c_int(0), C.c_double(0.0), C.c_int(-1)
    c.call("fpr_ns_run2d", c2.h, *(fptr(B[n], 2) for n in ("S", "T", "W", "T_rhs", "W_rhs")), *args, 1.0e9, 2, C.byref(t_c), C.byref(n_c),
           C.byref(dt_c), C.byref(bad))
    F.synchronize()
    assert (n_c.value, bad.value, dt_c.value, t_c.value) == (2, unconverged, dts[1], t)
    for n in ("S", "T", "W"):
        assert np.array_equal(F.tonumpy(A[n]), F.tonumpy(B[n])), n
    # `while sim_time < ttot`: one more step is taken from t, none from beyond ttot
    t_c2, n_c2 = C.c_double(t), C.c_int(0)
    c.call("fpr_ns_run2d", c2.h, *(fptr(B[n], 2) for n in ("S", "T", "W", "T_rhs", "W_rhs")), *args, t + 0.5 * dts[1], 100, C.byref(t_c2),
           C.byref(n_c2), C.byref(dt_c), C.byref(bad))
    assert n_c2.value == 1 and t_c2.value > t
    n_c3 = C.c_int(7)
    c.call("fpr_ns_run2d", c2.h, *(fptr(B[n], 2) for n in ("S", "T", "W", "T_rhs", "W_rhs")), *args, t, 100, C.byref(t_c2), C.byref(n_c3),
           C.byref(dt_c), C.byref(bad))
    assert n_c3.value == 0
