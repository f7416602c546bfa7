"""GPU parity tests for the NEXT row 8f-1 (Navier-Stokes step around the V-cycle), pinned by the
reference's FORTRAN fixtures (test/part2.jl)."""
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
