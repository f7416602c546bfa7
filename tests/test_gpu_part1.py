"""GPU parity tests, Part 1 (3D diffusion): HIP path (through the C ABI) vs the CPU oracle.
Pointwise kernels are bit-exact (both sides are compiled without FMA contraction); reductions agree
to 1e-13 relative (summation order)."""
import math

import numpy as np
import pytest

from fixtures_io import part1_reference, splitmix64_uniform
from oracle.oracle import asf, farr

pytestmark = pytest.mark.gpu

SHAPES = [(32, 32, 32), (45, 45, 45), (64, 40, 24), (91, 33, 17), (130, 6, 5), (3, 3, 3), (4, 5, 6), (128, 9, 70),
          (258, 11, 7)]
COEF = dict(dτ=0.0031, _dt=5.0, _dx=3.2, _dy=2.9, _dz=3.7, D_dx=3.2, D_dy=2.9, D_dz=3.7)
VARIANTS = [dict(diff3_variant=1), dict(diff3_variant=2, diff3_ry=1), dict(diff3_variant=2, diff3_ry=2),
            dict(diff3_variant=2, diff3_ry=4), dict(diff3_variant=2, diff3_ry=4, diff3_vx=1),
            dict(diff3_variant=2, diff3_ry=4, diff3_nt=1, diff3_zc=5), dict(diff3_variant=3, diff3_ry=1),
            dict(diff3_variant=3, diff3_ry=2, diff3_zc=7), dict(diff3_variant=3, diff3_ry=4, diff3_xcd_remap=1),
            dict(diff3_variant=4, diff3_ry=4), dict(diff3_variant=4, diff3_ry=2, diff3_zc=3),
            dict(diff3_variant=5, diff3_ry=2, diff3_zc=7), dict(diff3_variant=5, diff3_ry=4, diff3_nt=0),
            dict()]
ALL_OPTS = ["diff3_variant", "diff3_ry", "diff3_vx", "diff3_nt", "diff3_zc", "diff3_xcd_remap"]


def rnd(shape, seed):
    return asf(splitmix64_uniform(int(np.prod(shape)), seed).reshape(shape, order="F"))


def set_opts(F, opts):
    c = F.ctx()
    defaults = dict(diff3_variant=0, diff3_ry=0, diff3_vx=0, diff3_nt=-1, diff3_zc=0, diff3_xcd_remap=-1)
    for k in ALL_OPTS:
        c.set_option(k, opts.get(k, defaults[k]))


@pytest.mark.parametrize("opts", VARIANTS, ids=lambda o: "-".join("%s%s" % (k[6:], v) for k, v in o.items()) or "default")
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_fused_step_bit_exact(fpr, oracle, shape, opts):
    F = fpr
    set_opts(F, opts)
    try:
        Ht, Hτ = rnd(shape, 11), rnd(shape, 12)
        H2_ref, dH_ref = asf(np.full(shape, -7.0)), asf(np.full(shape, -9.0))  # sentinels: boundary untouched
        oracle.diffusion3d_step(Ht, Hτ, H2_ref, dH_ref, *COEF.values())
        H2, dH = F.asdevice(np.full(shape, -7.0)), F.asdevice(np.full(shape, -9.0))
        F.part1.diffusion_3D_step_τ(F.asdevice(Ht), F.asdevice(Hτ), H2, dH, *COEF.values())
        assert np.array_equal(F.tonumpy(dH), dH_ref)
        assert np.array_equal(F.tonumpy(H2), H2_ref)
        # fused norm
        sq = F.ctx().scal[:1]
        H2.fill_(-7.0); dH.fill_(-9.0)
        F.part1.diffusion_3D_step_τ_norm(F.asdevice(Ht), F.asdevice(Hτ), H2, dH, *COEF.values(), 0.2, sq)
        assert np.array_equal(F.tonumpy(dH), dH_ref) and np.array_equal(F.tonumpy(H2), H2_ref)
        inner = dH_ref[1:-1, 1:-1, 1:-1].copy(order="F")
        ref = oracle.sumsq_scaled(inner, 0.2)
        assert abs(sq.item() - ref) <= 1e-13 * ref
    finally:
        set_opts(F, {})


@pytest.mark.parametrize("opts", [dict(), dict(diff3_variant=3, diff3_ry=2), dict(diff3_variant=1)])
def test_box_form_covers_interior(fpr, oracle, opts):
    F = fpr
    set_opts(F, opts)
    try:
        shape = (40, 36, 30)
        Ht, Hτ = rnd(shape, 1), rnd(shape, 2)
        H2_ref, dH_ref = farr(*shape), farr(*shape)
        oracle.diffusion3d_step(Ht, Hτ, H2_ref, dH_ref, *COEF.values())
        gg = F.grid.GlobalGrid(*shape, dims=(1, 1, 1), use_dist=False)
        gg.neighbors = {0: None, 1: None, 2: None, 3: None, 4: None, 5: None}  # pretend all faces have neighbours
        boxes, inner = gg.boundary_boxes()
        assert len(boxes) == 6
        cells = sum((h[0] - l[0]) * (h[1] - l[1]) * (h[2] - l[2]) for l, h in boxes + [inner])
        assert cells == (shape[0] - 2) * (shape[1] - 2) * (shape[2] - 2)
        H2, dH = F.fzeros(*shape), F.fzeros(*shape)
        sq = F.ctx().scal[:1]
        sq.zero_()
        dHt, dHτ = F.asdevice(Ht), F.asdevice(Hτ)
        for lo, hi in boxes + [inner]:
            F.part1.diffusion_3D_step_τ_box(dHt, dHτ, H2, dH, *COEF.values(), lo, hi, 0.2, sq, 0)
        assert np.array_equal(F.tonumpy(dH), dH_ref) and np.array_equal(F.tonumpy(H2), H2_ref)
        ref = oracle.sumsq_scaled(dH_ref, 0.2)
        assert abs(sq.item() - ref) <= 1e-13 * ref
    finally:
        set_opts(F, {})


@pytest.mark.parametrize("shape", [(24, 24, 24), (33, 18, 9), (3, 3, 3)], ids=str)
def test_split_kernels_bit_exact(fpr, oracle, shape):
    F = fpr
    nx, ny, nz = shape
    D, dx, dy, dz, dt, dτ = 1.0, 0.31, 0.29, 0.37, 0.2, 0.004
    Ht, Hτ = rnd(shape, 3), rnd(shape, 4)
    qx, qy, qz = farr(nx - 1, ny - 2, nz - 2), farr(nx - 2, ny - 1, nz - 2), farr(nx - 2, ny - 2, nz - 1)
    dH = farr(nx - 2, ny - 2, nz - 2)
    Hτ_ref = Hτ.copy(order="F")
    oracle.diffusion3d_flux(qx, qy, qz, Hτ_ref, D, dx, dy, dz)
    oracle.diffusion3d_dHdtau(dH, Hτ_ref, Ht, qx, qy, qz, dt, dx, dy, dz)
    oracle.diffusion3d_update(Hτ_ref, dH, dτ)
    gqx, gqy, gqz = F.fzeros(*qx.shape), F.fzeros(*qy.shape), F.fzeros(*qz.shape)
    gdH, gHτ = F.fzeros(*dH.shape), F.asdevice(Hτ)
    F.part1.diffusion_3D_step_τ_(F.asdevice(Ht), gHτ, gdH, dt, dτ, gqx, gqy, gqz, dx, dy, dz, D)
    for g, r in ((gqx, qx), (gqy, qy), (gqz, qz), (gdH, dH), (gHτ, Hτ_ref)):
        assert np.array_equal(F.tonumpy(g), r)


def test_init_gaussian_and_norm(fpr, oracle):
    F = fpr
    n = (40, 31, 22)
    dx, dy, dz = 10.0 / 78, 10.0 / 31, 10.0 / 42
    ref = oracle.init_gaussian(n, dx, dy, dz, (5.0, 5.0, 5.0), (1, 0, 1))
    H = F.fzeros(*n)
    F.part1.init_local_gaussian((5.0, 5.0, 5.0), dx, dy, dz, H, (1, 0, 1))
    got = F.tonumpy(H)
    assert np.all(np.abs(got - ref) <= 4e-16 * np.abs(ref))  # exp() differs by <= 2 ulp between libms
    s = F.part1.local_sumsq(F.asdevice(ref), 0.2)
    r = oracle.sumsq_scaled(ref, 0.2)
    assert abs(s - r) <= 1e-13 * r
    assert abs(F.part1.dist_norm_L2(F.asdevice(ref), None, 0.2) - math.sqrt(r)) <= 1e-13 * math.sqrt(r)


def test_solver_matches_reference_fixture_and_oracle(fpr, oracle):
    """test/part1.jl:24-40 protocol: 32^3, ttot=1 (5 steps), tol 1e-8 vs test_1.bson (atol 1e-5); and
    the whole run equals the oracle bit for bit (same iteration counts => same fields)."""
    F = fpr
    ref = part1_reference()
    n = 32
    dx = 10.0 / n
    Ht0 = oracle.init_gaussian((n, n, n), dx, dx, dx, (5.0, 5.0, 5.0))
    Href = Ht0.copy(order="F")
    it_ref, err_ref, _, _ = oracle.diffusion3d_solve(Href, nt=5, tol=1e-8)
    X, H, bench, info = F.part1.diffusion_3D_kernel_programming(nx=n, ny=n, nz=n, ttot=1.0, tol=1e-8,
                                                                Ht_init=F.asdevice(Ht0))
    assert info["iters"] == it_ref == [188, 187, 185, 184, 183]
    assert np.array_equal(H, Href)
    assert np.allclose(info["err"], err_ref, rtol=1e-12, atol=0)
    inds = np.ceil(np.linspace(1, n, 12)).astype(int) - 1
    assert np.abs(H[:, :, 14][np.ix_(inds, inds)] - ref["H"]).max() < 1e-5
    assert np.allclose(X[inds], ref["X"], atol=1e-5, rtol=0)
    assert bench.Work == 0 or bench.Work > 0  # BenchResults fields populated
    # Gaussian initialised on the device instead of uploaded: same result to 1e-12
    X2, H2, _, info2 = F.part1.diffusion_3D_kernel_programming(nx=n, ny=n, nz=n, ttot=1.0, tol=1e-8)
    assert abs(sum(info2["iters"]) - sum(it_ref)) <= 1
    assert np.abs(H2[1:-1, 1:-1, 1:-1] - Href[1:-1, 1:-1, 1:-1]).max() < 1e-7


def test_config1_64cubed_50_iterations(fpr, oracle):
    """BASELINE config 1: 64^3, exactly 50 pseudo-iterations of the first step."""
    F = fpr
    n = 64
    dx = 10.0 / n
    Ht0 = oracle.init_gaussian((n, n, n), dx, dx, dx, (5.0, 5.0, 5.0))
    Href = Ht0.copy(order="F")
    _, err_ref, Hτ_ref, dH_ref = oracle.diffusion3d_solve(Href, nt=1, fixed_iters=50)
    _, H, _, info = F.part1.diffusion_3D_kernel_programming(nx=n, ny=n, nz=n, ttot=0.2, fixed_iters=50,
                                                            Ht_init=F.asdevice(Ht0))
    assert np.array_equal(H, Href)
    assert np.array_equal(F.tonumpy(info["residual_H"]), dH_ref)
    assert abs(info["err"][0] - err_ref[0]) <= 1e-13 * err_ref[0]


def test_full_size_512_properties(fpr):
    """BASELINE config 2 size (512^3): size-independent properties instead of the (slow) oracle:
    constant fields are fixed points; every kernel variant agrees bit for bit; mirror symmetry."""
    import torch

    F = fpr
    n = 512
    dx = 10.0 / n
    coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
    Ht, Hτ = F.fzeros(n, n, n), F.fzeros(n, n, n)
    H2, dH = F.fzeros(n, n, n), F.fzeros(n, n, n)
    Ht.fill_(1.25); Hτ.fill_(1.25)
    F.part1.diffusion_3D_step_τ(Ht, Hτ, H2, dH, *coef)
    assert float(dH.abs().max()) == 0.0
    assert float((H2[1:-1, 1:-1, 1:-1] - 1.25).abs().max()) == 0.0
    F.part1.init_local_gaussian((5.0, 5.0, 5.0), dx, dx, dx, Ht)
    Hτ.copy_(Ht)
    Hτ.mul_(1.0 + 0.001 * torch.arange(n, device=Hτ.device, dtype=torch.float64).reshape(n, 1, 1))  # break symmetry in x
    results = []
    for opts in (dict(diff3_variant=1), dict(diff3_variant=2), dict(diff3_variant=3), dict(diff3_variant=4),
                 dict(diff3_variant=5, diff3_zc=16), dict()):
        set_opts(F, opts)
        H2.zero_(); dH.zero_()
        F.part1.diffusion_3D_step_τ(Ht, Hτ, H2, dH, *coef)
        results.append((H2.clone(), dH.clone()))
    set_opts(F, {})
    for a, b in results[1:]:
        assert torch.equal(a, results[0][0]) and torch.equal(b, results[0][1])
    # mirror symmetry j -> n-1-j (and k -> n-1-k) of the input is preserved bit for bit
    # (dx = 10/512 is exact in binary, so the Gaussian is exactly mirror-symmetric)
    d = results[0][1]
    assert torch.equal(d, d.flip(1)) and torch.equal(d, d.flip(2))
    assert float(d.abs().max()) > 0


# ---------------------------------------------------------------- outputs of the reference itself
def _published(name):
    import csv
    import os

    from fixtures_io import GOLDEN

    with open(os.path.join(GOLDEN, "published", name)) as fh:
        return list(csv.DictReader(fh))


def _probe_value(F, oracle, n, tol):
    dx = 10.0 / n
    Ht0 = oracle.init_gaussian((n, n, n), dx, dx, dx, (5.0, 5.0, 5.0))  # exp() of the host libm, as the reference
    X, H, _, info = F.part1.diffusion_3D_kernel_programming(nx=n, ny=n, nz=n, ttot=2.0, tol=tol, Ht_init=F.asdevice(Ht0),
                                                            return_device=True)
    i = int(np.round(4.5 / (X[1] - X[0]) + 1)) - 1
    info["interp"] = F.part1.linear_interpolate_3D(F.tonumpy(H), X[1] - X[0])
    return float(H[i, i, i].item()), info


@pytest.mark.parametrize("n", [16, 23, 32, 45, 64, 91, 128])
def test_published_grid_size_values(fpr, oracle, n):
    """benchmark-results/error_vs_grid_size_experiment_results.csv (written by the reference itself, ttot=2,
    tol=1e-6): the HIP path reproduces H[ix,iy,iz] to the last printed digit for every grid size."""
    row = [r for r in _published("error_vs_grid_size_experiment_results.csv") if int(r["nx"]) == n][0]
    v, info = _probe_value(fpr, oracle, n, 1e-6)
    assert v == float(row["val"]), (v, row["val"], info["iters"])
    iv = float(row["interp_val"])
    assert abs(info["interp"] - iv) <= 2.5e-16 * abs(iv), (info["interp"], iv)


@pytest.mark.parametrize("k", range(8))
def test_published_tolerance_sweep_128(fpr, oracle, k):
    """benchmark-results/error_vs_tolerance_experiment_results.csv: 128^3, ttot=2, tol = 1e-3 ... 1e-10."""
    row = _published("error_vs_tolerance_experiment_results.csv")[k]
    tol = float(row["tol"])
    v, info = _probe_value(fpr, oracle, 128, tol)
    ref = float(row["val"])
    # every one of the up to 41 208 convergence decisions agrees with the reference's run: the value is
    # identical to the last printed digit (the device reductions are deterministic, so this is stable)
    assert v == ref, (v, ref, info["iters"])
    # the second probe cell differs by 1 ulp in ONE of the 8 runs: Julia's exp() (initial condition) and the
    # host libm's are both < 1 ulp but not identical; a 1-ulp seed survives the (contractive) iteration
    iv = float(row["interp_val"])
    assert abs(info["interp"] - iv) <= 2.5e-16 * abs(iv), (info["interp"], iv)
    print("tol %g: value %.17g published %.17g exact=%s iters=%d" % (tol, v, ref, v == ref, sum(info["iters"])))
