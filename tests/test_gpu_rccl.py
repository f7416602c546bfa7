"""The RCCL transport of libfpr_hip.so (csrc/comm.hip) on ONE GPU: a single rank that is its own neighbour through
periodic boundaries, so every plane really travels through ncclSend / ncclRecv on the comm stream (RCCL refuses two
ranks on one device, and a box has one GPU; N > 1 runs only in the driver's scaling bench).  Checks update_halo!
semantics (dimension by dimension, corners consistent), the split begin / end form, the all-reduce, gather!, and
the step / fused-pair choreographies of GlobalGrid over this transport against the oracle with wrapped halos."""
import ctypes as C

import numpy as np
import pytest

from fixtures_io import splitmix64_uniform
from oracle.oracle import asf, farr

pytestmark = pytest.mark.gpu


def rnd(shape, seed):
    return asf(splitmix64_uniform(int(np.prod(shape)), seed).reshape(shape, order="F"))


def wrap(A, dims=(0, 1, 2)):
    """Periodic halo update of a single rank, dimension by dimension (what update_halo! does there)."""
    for d in dims:
        lo, hi = [slice(None)] * 3, [slice(None)] * 3
        src_hi, src_lo = [slice(None)] * 3, [slice(None)] * 3
        lo[d], hi[d], src_hi[d], src_lo[d] = 0, -1, -2, 1
        A[tuple(lo)] = A[tuple(src_hi)]
        A[tuple(hi)] = A[tuple(src_lo)]
    return A


@pytest.fixture()
def periodic_grid(fpr):
    grids = []

    def make(n, periods):
        gg = fpr.grid.GlobalGrid(*n, dims=(1, 1, 1), periods=periods, transport="rccl", use_dist=False)
        grids.append(gg)
        return gg

    yield make
    fpr.grid.finalize_global_grid()


@pytest.mark.parametrize("n", [(10, 9, 8), (64, 5, 7), (33, 34, 35)], ids=str)
def test_update_halo_self_neighbour(fpr, periodic_grid, n):
    F = fpr
    gg = periodic_grid(n, (1, 1, 1))
    assert sorted(gg.neighbors) == [0, 1, 2, 3, 4, 5] and gg.nx_g() == n[0] - 2
    c = F.ctx()
    assert c.L.fpr_comm_size(c.h) == 1 and c.L.fpr_comm_rank(c.h) == 0
    A = rnd(n, 3)
    gA = F.asdevice(A)
    gg.update_halo_(gA)
    assert np.array_equal(F.tonumpy(gA), wrap(A.copy(order="F")))   # corners / edges included
    # split form, z faces only: x / y halos stay, z planes wrap (in place, no pack buffers)
    B = rnd(n, 4)
    gB = F.asdevice(B)
    tr = gg.transport()
    tok = tr.begin(gB, F.grid.ZFACES)
    tr.end(gB, F.grid.ZFACES, tok)
    assert np.array_equal(F.tonumpy(gB), wrap(B.copy(order="F"), dims=(2,)))
    # all faces at once: face interiors as a wrap of the ORIGINAL array (edges / corners are not refreshed)
    D = rnd(n, 5)
    gD = F.asdevice(D)
    tok = tr.begin(gD, F.grid.ALLFACES)
    tr.end(gD, F.grid.ALLFACES, tok)
    got = F.tonumpy(gD)
    for d in range(3):
        idx = [slice(1, -1)] * 3
        src = [slice(1, -1)] * 3
        idx[d], src[d] = 0, -2
        assert np.array_equal(got[tuple(idx)], D[tuple(src)])
        idx[d], src[d] = -1, 1
        assert np.array_equal(got[tuple(idx)], D[tuple(src)])
    assert np.array_equal(got[1:-1, 1:-1, 1:-1], D[1:-1, 1:-1, 1:-1])


def test_allreduce_and_gather_single_rank(fpr, periodic_grid):
    F = fpr
    n = (12, 10, 9)
    gg = periodic_grid(n, (0, 0, 1))
    c = F.ctx()
    t = F.asdevice(np.array([1.5, -2.25, 3.0]))
    c.call("fpr_allreduce_sum_dev", t.data_ptr(), 3, 0)      # a real ncclAllReduce on a 1-rank communicator
    assert t.cpu().tolist() == [1.5, -2.25, 3.0]
    x = C.c_double(0.625)
    c.call("fpr_allreduce_sum1", C.byref(x))
    assert x.value == 0.625
    A = rnd(n, 9)
    G = F.grid.gather_global_(gg, F.asdevice(A))
    assert G.shape == n and np.array_equal(G, A)


@pytest.mark.parametrize("n", [(20, 12, 10), (128, 24, 16)], ids=str)
def test_step_and_fused_pair_over_rccl_periodic_z(fpr, oracle, periodic_grid, n):
    """GlobalGrid.step / step2 with the library's transport: periodic z on one rank = the oracle's step followed by a
    wrap of the z halo planes of the NEW buffer.  Fields, residuals and norms bit-exact / 1e-13."""
    F = fpr
    gg = periodic_grid(n, (0, 0, 1))
    nx, ny, nz = n
    dx, dy, dz = 10.0 / nx, 10.0 / ny, 10.0 / (nz - 2)
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    Ht = wrap(rnd(n, 11), dims=(2,))
    A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
    gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(A), F.fzeros(*n), F.fzeros(*n)
    sq = F.fzeros(1)
    for it in range(4):
        oracle.diffusion3d_step(Ht, A, B, R, *coef)
        wrap(B, dims=(2,))
        A, B = B, A
        gg.step(gHt, gA, gB, gR, *coef, dt, sq)
        gA, gB = gB, gA
        ref = oracle.sumsq_scaled(R, dt)
        assert abs(float(sq.item()) - ref) <= 1e-13 * ref
    assert np.array_equal(F.tonumpy(gA), A) and np.array_equal(F.tonumpy(gR), R)
    gC = gA.clone()
    if not gg.can_step2(gHt, gA, gB, gC, gR):
        assert nx < 128
        return
    # fused pairs: B plays the reference's second buffer (its z halo planes receive level 1), C carries A's boundary
    sq2 = F.fzeros(2)
    for it in range(3):
        refs = []
        for k in range(2):
            oracle.diffusion3d_step(Ht, A, B, R, *coef)
            wrap(B, dims=(2,))
            A, B = B, A
            refs.append(oracle.sumsq_scaled(R, dt))
        gg.step2(gHt, gA, gB, gC, gR, *coef, dt, sq2)
        gA, gC = gC, gA
        got = sq2.cpu().tolist()
        assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs)), (it, got, refs)
        assert np.array_equal(F.tonumpy(gA)[1:-1, 1:-1, :], A[1:-1, 1:-1, :])
        assert np.array_equal(F.tonumpy(gR), R)


def test_fused_pair_over_rccl_periodic_xyz(fpr, oracle, periodic_grid):
    """The general fused-pair choreography (shell boxes on all six faces, x / y planes through the library's pack
    kernels, one-cell x-slabs in the narrow-box kernel) over the library's RCCL transport: one rank, periodic in
    x, y and z.  Interior cells, residuals and both norms against the oracle with wrapped halos."""
    F = fpr
    n = (128, 24, 16)
    gg = periodic_grid(n, (1, 1, 1))
    nx, ny, nz = n
    dx, dy, dz = 10.0 / (nx - 2), 10.0 / (ny - 2), 10.0 / (nz - 2)
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    Ht = wrap(rnd(n, 21))
    A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
    gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(A), F.fzeros(*n), F.fzeros(*n)
    gC = gA.clone()
    assert gg.can_step2(gHt, gA, gB, gC, gR)
    sq2 = F.fzeros(2)
    for it in range(3):
        refs = []
        for k in range(2):
            oracle.diffusion3d_step(Ht, A, B, R, *coef)
            wrap(B)
            A, B = B, A
            refs.append(oracle.sumsq_scaled(R, dt))
        gg.step2(gHt, gA, gB, gC, gR, *coef, dt, sq2)
        gA, gC = gC, gA
        got = sq2.cpu().tolist()
        assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs)), (it, got, refs)
        loc = F.tonumpy(gA)
        assert np.array_equal(loc[1:-1, 1:-1, 1:-1], A[1:-1, 1:-1, 1:-1])
        for d in range(3):   # halo faces (interiors of the faces; edges / corners are not refreshed by one exchange)
            for side in (0, -1):
                idx = [slice(1, -1)] * 3
                idx[d] = side
                assert np.array_equal(loc[tuple(idx)], A[tuple(idx)])
        assert np.array_equal(F.tonumpy(gR)[1:-1, 1:-1, 1:-1], R[1:-1, 1:-1, 1:-1])


@pytest.mark.parametrize("n,zc", [((128, 40, 36), 0), ((130, 30, 12), 0), ((256, 50, 23), 0), ((128, 64, 40), 9)], ids=str)
def test_triples_chained_over_rccl_periodic_z(fpr, oracle, periodic_grid, n, zc):
    """GlobalGrid.step3 between ranks (fpr_diffusion3d_step3_halo): a rank that is its own z-neighbour over the library's transport.  Core
    planes [3, nz-3) as one launch of the three-step kernel, the two planes next to each z-face in three rounds of single-step launches on
    6-plane slabs with one-plane exchanges in between.  Eight triples chained on the core / comm streams (join=False), then one left pending
    and joined by a single step: fields (halo planes and all), residual bit for bit and every norm to 1e-13 against the oracle's single
    steps with wrapped z halos.  The second buffer starts as zeros, so its boundary ring differs from the first one's (the reference's
    Hτ2 = @zeros, :141): iterates 1 and 3 carry one ring, iterate 2 the other."""
    F = fpr
    gg = periodic_grid(n, (0, 0, 1))
    nx, ny, nz = n
    dx, dy, dz = 10.0 / nx, 10.0 / ny, 10.0 / (nz - 2)
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    if zc:      # test hook: the core [3, nz-3) in chunks of zc planes (a small grid otherwise runs as one chunk)
        F.ctx().set_option("diff3_zc2", zc)
        try:
            _triples_periodic_z(F, oracle, periodic_grid, gg, n, coef, dt)
        finally:
            F.ctx().set_option("diff3_zc2", 0)
        return
    _triples_periodic_z(F, oracle, periodic_grid, gg, n, coef, dt)


def _triples_periodic_z(F, oracle, periodic_grid, gg, n, coef, dt):
    nx, ny, nz = n
    Ht = wrap(rnd(n, 77), dims=(2,))
    A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
    gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(A), F.fzeros(*n), F.fzeros(*n)
    assert gg.can_step3(gHt, gA, gB, gR)
    ntr = 8
    sq = F.fzeros(3 * ntr + 3)
    refs = []

    def oracle_steps(k):
        nonlocal A, B
        for _ in range(k):
            oracle.diffusion3d_step(Ht, A, B, R, *coef)
            wrap(B, dims=(2,))
            A, B = B, A
            refs.append(oracle.sumsq_scaled(R, dt))

    for t in range(ntr):
        oracle_steps(3)
        gg.step3(gHt, gA, gB, gR, *coef, dt, sq[3 * t:3 * t + 3], join=False)
        gA, gB = gB, gA
    assert gg.pending
    gg.allreduce_(sq)
    assert not gg.pending
    got = sq.cpu().tolist()[:3 * ntr]
    assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs)), (got, refs)
    assert np.array_equal(F.tonumpy(gA), A) and np.array_equal(F.tonumpy(gR)[1:-1, 1:-1, 1:-1], R[1:-1, 1:-1, 1:-1])
    # without sums and without a residual array; then a triple left pending and a single step that joins it
    oracle_steps(3)
    gg.step3(gHt, gA, gB, None, *coef, dt, None)
    gA, gB = gB, gA
    assert np.array_equal(F.tonumpy(gA), A)
    oracle_steps(3)
    gg.step3(gHt, gA, gB, gR, *coef, dt, sq[0:3], join=False)
    gA, gB = gB, gA
    oracle_steps(1)
    gg.step(gHt, gA, gB, gR, *coef, dt, sq[3:4])
    gA, gB = gB, gA
    assert not gg.pending
    assert abs(float(sq[3].item()) - refs[-1]) <= 1e-13 * refs[-1]
    assert np.array_equal(F.tonumpy(gA), A) and np.array_equal(F.tonumpy(gR)[1:-1, 1:-1, 1:-1], R[1:-1, 1:-1, 1:-1])
    # a chain of triples that ends in a PAIR (bench.py's remainder): the pair runs on the triples' split of the device (32 units)
    gC = gB.clone()          # the pair's third buffer carries the boundary ring of the triple's OUTPUT buffer (nothing is pending here)
    oracle_steps(3)
    gg.step3(gHt, gA, gB, gR, *coef, dt, sq[0:3], join=False)
    gA, gB = gB, gA
    if gg.can_step2(gHt, gA, gB, gC, gR):
        oracle_steps(2)
        gg.step2(gHt, gA, gB, gC, gR, *coef, dt, sq[3:5], join=False)
        gA, gC = gC, gA
        gg.join()
        assert F.ctx().L.fpr_comm_cus(F.ctx().h) == 32
        got = sq.cpu().tolist()[3:5]
        assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs[-2:])), (got, refs[-2:])
        assert np.array_equal(F.tonumpy(gA)[1:-1, 1:-1, :], A[1:-1, 1:-1, :]) and np.array_equal(F.tonumpy(gR)[1:-1, 1:-1, 1:-1], R[1:-1, 1:-1, 1:-1])
    # x / y neighbours or too few planes: the caller is told (and runs pairs)
    F.grid.finalize_global_grid()
    g2 = periodic_grid((128, 24, 16), (0, 1, 1))
    z = [F.fzeros(128, 24, 16) for _ in range(4)]
    assert not g2.can_step3(*z)
    F.grid.finalize_global_grid()
    g3 = periodic_grid((128, 24, 10), (0, 0, 1))
    z = [F.fzeros(128, 24, 10) for _ in range(4)]
    assert not g3.can_step3(*z)
    with pytest.raises(Exception, match="three fused iterations between ranks"):
        g3.step3(*z, *coef, dt, None)


@pytest.mark.parametrize("periods,n", [((0, 0, 1), (128, 40, 36)), ((1, 1, 1), (256, 36, 20)), ((1, 0, 1), (128, 24, 16))], ids=["z", "xyz", "xz"])
def test_pairs_chained_on_core_and_comm_streams(fpr, oracle, periodic_grid, periods, n):
    """step2(join=False): consecutive fused pairs chain on the core / comm streams of the split device without passing through
    the compute stream (what bench.py runs between ranks); join() -- or allreduce_ / step / update_halo_ -- orders the compute
    stream behind them.  Twelve pairs, every pair's two norms kept in their own slots; then a pair left pending and two single
    steps (GlobalGrid.step on the split device): fields and residual bit for bit, all norms to 1e-13, against the oracle with
    wrapped halos."""
    F = fpr
    dims = tuple(d for d in range(3) if periods[d])
    gg = periodic_grid(n, periods)
    ext = [m - 2 if p else m for m, p in zip(n, periods)]
    dx, dy, dz = 10.0 / ext[0], 10.0 / ext[1], 10.0 / ext[2]
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    Ht = wrap(rnd(n, 31), dims=dims)
    A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
    gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(A), F.fzeros(*n), F.fzeros(*n)
    gC = gA.clone()
    assert gg.can_step2(gHt, gA, gB, gC, gR)
    npairs = 12
    sq = F.fzeros(2 * npairs + 2)
    refs = []
    for p in range(npairs):
        for k in range(2):
            oracle.diffusion3d_step(Ht, A, B, R, *coef)
            wrap(B, dims=dims)
            A, B = B, A
            refs.append(oracle.sumsq_scaled(R, dt))
        gg.step2(gHt, gA, gB, gC, gR, *coef, dt, sq[2 * p:2 * p + 2], join=False)
        gA, gC = gC, gA
    assert gg.pending
    gg.allreduce_(sq)          # joins (a single rank: nothing else happens)
    assert not gg.pending
    got = sq.cpu().tolist()[:2 * npairs]
    assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs)), (got, refs)
    inner = (slice(1, -1),) * 3
    assert np.array_equal(F.tonumpy(gA)[inner], A[inner]) and np.array_equal(F.tonumpy(gR)[inner], R[inner])
    # one more pair left pending, then two single steps on the split device (the first one joins): the field is back in the
    # first work buffer after an even number of steps, as a fused pair needs it
    gg.step2(gHt, gA, gB, gC, gR, *coef, dt, sq[0:2], join=False)
    gA, gC = gC, gA
    for k in range(2):
        oracle.diffusion3d_step(Ht, A, B, R, *coef)
        wrap(B, dims=dims)
        A, B = B, A
    assert gg.pending
    for it in range(2):
        oracle.diffusion3d_step(Ht, A, B, R, *coef)
        wrap(B, dims=dims)
        A, B = B, A
        gg.step(gHt, gA, gB, gR, *coef, dt, sq[2:3])
        gA, gB = gB, gA
        assert not gg.pending
        ref = oracle.sumsq_scaled(R, dt)
        assert abs(float(sq[2].item()) - ref) <= 1e-13 * ref
        assert np.array_equal(F.tonumpy(gA)[inner], A[inner]) and np.array_equal(F.tonumpy(gR)[inner], R[inner])


@pytest.mark.parametrize("periods,n", [((1, 0, 0), (128, 24, 16)), ((1, 1, 0), (130, 30, 21)), ((1, 1, 1), (132, 70, 13))], ids=["x", "xy", "xyz"])
@pytest.mark.parametrize("form", ["strips", "strips-no-residual", "field"])
def test_x_shell_in_compact_strips(fpr, oracle, periodic_grid, periods, n, form):
    """The shell next to an x-neighbour in compact strips (csrc/diffusion3d_xstrip.hpp; part1_kernel_programming.jl:182-188 for a
    fused pair): seven chained pairs (the strips of a pair's level 0 come from the pair before it: `turn` between two core
    launches), then a joined pair (plain gather) -- fields, halo planes, residual bit for bit and norms to 1e-13 against the
    oracle with wrapped halos; the same with the strips gathered afresh for every pair, without a residual array, and through
    the field (option diff3_xstrips = 0: the narrow-box kernels and pack / unpack kernels of rounds 2-3).  70 rows / 13 planes:
    two row tiles, two plane chunks with a short last one."""
    F = fpr
    c = F.ctx()
    dims = tuple(d for d in range(3) if periods[d])
    gg = periodic_grid(n, periods)
    ext = [m - 2 if p else m for m, p in zip(n, periods)]
    dx, dy, dz = 10.0 / ext[0], 10.0 / ext[1], 10.0 / ext[2]
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    Ht = wrap(rnd(n, 77), dims=dims)
    A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
    gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(A), F.fzeros(*n), F.fzeros(*n)
    gC = gA.clone()
    if not gg.can_step2(gHt, gA, gB, gC, gR):
        pytest.skip("no fused pairs at this size")
    res = None if form == "strips-no-residual" else gR
    c.set_option("diff3_xstrips", 0 if form == "field" else 1)
    try:
        npairs = 8
        sq = F.fzeros(2 * npairs)
        refs = []
        for p in range(npairs):
            for k in range(2):
                oracle.diffusion3d_step(Ht, A, B, R, *coef)
                wrap(B, dims=dims)
                A, B = B, A
                refs.append(oracle.sumsq_scaled(R, dt))
            if p == npairs - 1:
                gg.join()
            gg.step2(gHt, gA, gB, gC, res, *coef, dt, sq[2 * p:2 * p + 2], join=(p == npairs - 1))
            gA, gC = gC, gA
        assert not gg.pending
        got = sq.cpu().tolist()
        assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs)), (got, refs)
        loc = F.tonumpy(gA)
        inner = (slice(1, -1),) * 3
        assert np.array_equal(loc[inner], A[inner])
        for d in dims:   # halo planes (their interiors: edges and corners are not refreshed by one exchange of all faces)
            for side in (0, -1):
                idx = [slice(1, -1)] * 3
                idx[d] = side
                assert np.array_equal(loc[tuple(idx)], A[tuple(idx)])
        if res is not None:
            assert np.array_equal(F.tonumpy(gR)[inner], R[inner])
    finally:
        c.set_option("diff3_xstrips", 1)


def test_x_strips_are_gathered_afresh_after_a_join(fpr, oracle, periodic_grid):
    """The strips of a pending pair are reused by the next pair of the chain; once the pair has been joined the caller may change
    Htau and Ht (the solver does: `Ht .= Hτ`, part1_kernel_programming.jl:203) -- the next pair must not see the old columns.
    Same array addresses before and after, new contents; also a pair in the field form between two strip pairs."""
    F = fpr
    c = F.ctx()
    n, periods = (128, 24, 16), (1, 0, 1)
    dims = (0, 2)
    gg = periodic_grid(n, periods)
    ext = [m - 2 if p else m for m, p in zip(n, periods)]
    dx, dy, dz = 10.0 / ext[0], 10.0 / ext[1], 10.0 / ext[2]
    dt = 0.2
    dτ = min(dx, dy, dz) ** 2 / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, 1 / dx, 1 / dy, 1 / dz)
    Ht = wrap(rnd(n, 79), dims=dims)
    A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
    gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(A), F.fzeros(*n), F.fzeros(*n)
    gC = gA.clone()
    sq = F.fzeros(2)
    inner = (slice(1, -1),) * 3

    def pair(join, strips=1):
        nonlocal A, B, gA, gC
        for k in range(2):
            oracle.diffusion3d_step(Ht, A, B, R, *coef)
            wrap(B, dims=dims)
            A, B = B, A
        c.set_option("diff3_xstrips", strips)
        try:
            gg.step2(gHt, gA, gB, gC, gR, *coef, dt, sq, join=join)
        finally:
            c.set_option("diff3_xstrips", 1)
        gA, gC = gC, gA

    pair(False); pair(False)
    gg.join()
    assert np.array_equal(F.tonumpy(gA)[inner], A[inner])
    # the caller's own update between two chains: new time level, field scaled (same device arrays)
    Ht[...] = A
    gHt.copy_(gA)
    A[inner] *= 1.25
    wrap(A, dims=dims)
    gA.copy_(F.asdevice(A))
    gC.copy_(gA)
    pair(False); pair(False)
    gg.join()
    assert np.array_equal(F.tonumpy(gA)[inner], A[inner]) and np.array_equal(F.tonumpy(gR)[inner], R[inner])
    # strips, field form, strips inside one chain
    pair(False); pair(False, strips=0); pair(False); pair(False)
    gg.join()
    assert np.array_equal(F.tonumpy(gA)[inner], A[inner]) and np.array_equal(F.tonumpy(gR)[inner], R[inner])


def test_x_strips_with_one_sided_faces_equal_the_field_form(fpr, periodic_grid):
    """A rank with ONE x-face (low or high) and one y- / z-face, as in a (2,2,2) decomposition: no oracle for a rank that is its
    own neighbour on one side only, but the strips and the field form (diff3_xstrips = 0) must agree bit for bit."""
    F = fpr
    c = F.ctx()
    n = (128, 36, 20)
    dx = 10.0 / 126
    coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
    Ht = rnd(n, 78)
    out = {}
    for drop in (0b010101, 0b101010, 0b111110, 0b111101):
        for strips in (1, 0):
            gg = F.grid.GlobalGrid(*n, dims=(1, 1, 1), periods=(1, 1, 1), transport="rccl", use_dist=False, drop_faces=drop)
            c.set_option("diff3_xstrips", strips)
            try:
                gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(Ht), F.fzeros(*n), F.fzeros(*n)
                gC = gA.clone()
                sq = F.fzeros(8)
                for p in range(4):
                    gg.step2(gHt, gA, gB, gC, gR, *coef, 0.2, sq[2 * p:2 * p + 2], join=False)
                    gA, gC = gC, gA
                gg.join()
                out[strips] = (F.tonumpy(gA), F.tonumpy(gR), sq.cpu().tolist())
            finally:
                c.set_option("diff3_xstrips", 1)
                F.grid.finalize_global_grid()
        inner = (slice(1, -1),) * 3
        assert np.array_equal(out[1][0][inner], out[0][0][inner]) and np.array_equal(out[1][1][inner], out[0][1][inner]), drop
        for d in range(3):
            for side in (0, -1):
                idx = [slice(1, -1)] * 3
                idx[d] = side
                assert np.array_equal(out[1][0][tuple(idx)], out[0][0][tuple(idx)]), (drop, d, side)
        assert all(abs(a - b) <= 1e-13 * abs(b) for a, b in zip(out[1][2], out[0][2])), (drop, out[1][2], out[0][2])


def test_core_launch_serves_every_unit_whatever_the_comm_unit_map_says(fpr, periodic_grid):
    """The core launch of a z-slab rank's pair has one workgroup per compute unit; the ones on a comm unit (a map a probe launch
    filled) leave, the others claim the (tile, chunk) units by tickets.  With a WRONG map -- every unit marked, or every other key --
    the workgroups on 'comm units' take the work that is left once every workgroup has started: same fields, residual and norms
    bit for bit (512 x 512 x 66: 255 (tile, chunk) units for 240 workgroups, the reserved form with left-over slices)."""
    F = fpr
    n, periods = (512, 512, 66), (0, 0, 1)
    gg = periodic_grid(n, periods)
    dx = 10.0 / 510
    coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
    Ht = wrap(rnd(n, 91), dims=(2,))
    c = F.ctx()

    def pairs():
        gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(Ht), F.fzeros(*n), F.fzeros(*n)
        gC = gA.clone()
        sq = F.fzeros(6)
        for p in range(3):
            gg.step2(gHt, gA, gB, gC, gR, *coef, 0.2, sq[2 * p:2 * p + 2], join=False)
            gA, gC = gC, gA
        gg.join()
        return F.tonumpy(gA), F.tonumpy(gR), sq.cpu().tolist(), c.get_option("diff3_last_bal")

    ref = pairs()
    if c.L.fpr_comm_cus(c.h) % 32 == 0:
        pytest.skip("the probe launch did not identify the comm stream's units on this device: masked core stream, no map")
    assert ref[3] > 0          # the reserved form ran
    try:
        for mode in (1, 2):
            c.set_option("diff3_reserved_test", mode)
            got = pairs()
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2] and got[3] == ref[3], mode
    finally:
        c.set_option("diff3_reserved_test", 0)


def test_one_call_pair_without_sums_and_after_an_error(fpr, periodic_grid):
    """fpr_diffusion3d_step2_halo with sumsq2_dev = NULL (no norm is computed anywhere in the choreography) and joined at once gives the
    fields of the call that computes the sums; a call the library refuses (arrays of another size than fpr_grid_init's) raises, leaves
    no pair pending, and the next valid calls give the same result again."""
    F = fpr
    n, periods = (128, 24, 20), (1, 0, 1)
    gg = periodic_grid(n, periods)
    dx = 10.0 / 126
    coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
    Ht = wrap(rnd(n, 77), dims=(0, 2))

    def pairs(with_sums, join):
        gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(Ht), F.fzeros(*n), F.fzeros(*n)
        gC = gA.clone()
        sq = F.fzeros(2) if with_sums else None
        for _ in range(3):
            gg.step2(gHt, gA, gB, gC, gR, *coef, 0.2, sq, join=join)
            gA, gC = gC, gA
        gg.join()
        return F.tonumpy(gA), F.tonumpy(gR), (sq.cpu().tolist() if with_sums else None)

    a_f, a_r, a_s = pairs(True, False)
    b_f, b_r, _ = pairs(False, True)
    assert np.array_equal(a_f, b_f) and np.array_equal(a_r, b_r) and all(v > 0 for v in a_s)
    m = (64, 24, 20)
    small = [F.fzeros(*m) for _ in range(5)]
    gg_pending_before = gg.pending
    with pytest.raises(F.FprError):
        F.ctx().call("fpr_diffusion3d_step2_halo", *[F._lib.fptr(x, 3) for x in small], *m, *coef, 0.2, None, 0)
    assert not gg_pending_before and not gg.pending
    c_f, c_r, c_s = pairs(True, False)
    assert np.array_equal(a_f, c_f) and np.array_equal(a_r, c_r) and a_s == c_s


def test_synchronize_joins_a_pending_pair(fpr, periodic_grid):
    """fpr_synchronize drains all three streams, so it IS a join: a pair left pending (join = 0), then fpr_synchronize, then
    work on the compute stream that rewrites the pair's inputs (`Ht .= Hτ`, part1_kernel_programming.jl:203), then the next
    pair -- which must fork from the compute stream again, behind that work.  (Round 3 left `pair_pending` set: the second
    pair skipped the fork and could read Ht / Hτ while they were still being written.)  Compared with the same sequence
    joined explicitly; the heavy copies are repeated so that an unordered pair would start long before they end."""
    import torch

    F = fpr
    n = (256, 192, 160)
    gg = periodic_grid(n, (0, 0, 1))
    dx = 10.0 / 126
    coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
    H0 = wrap(rnd(n, 91), dims=(2,))
    big = [F.fzeros(*n) for _ in range(2)]

    def run(use_synchronize):
        gHt, gA, gB, gR = F.asdevice(H0), F.asdevice(H0), F.fzeros(*n), F.fzeros(*n)
        gC = gA.clone()
        c = F.ctx()
        c.call("fpr_diffusion3d_step2_halo", *[F._lib.fptr(x, 3) for x in (gHt, gA, gB, gC, gR)], *n, *coef, 0.2, None, 0)
        if use_synchronize:
            c.synchronize()
        else:
            c.call("fpr_diffusion3d_join")
            torch.cuda.synchronize()
        for _ in range(24):                      # compute-stream work ahead of the commit (keeps the stream busy for a while)
            big[0].copy_(big[1])
        gHt.copy_(gC)                            # Ht .= Hτ on the compute stream
        gA.copy_(gC)
        c.call("fpr_diffusion3d_step2_halo", *[F._lib.fptr(x, 3) for x in (gHt, gA, gB, gC, gR)], *n, *coef, 0.2, None, 0)
        c.call("fpr_diffusion3d_join")
        torch.cuda.synchronize()
        return F.tonumpy(gC), F.tonumpy(gR)

    ref_f, ref_r = run(False)
    for _ in range(3):
        f, r = run(True)
        assert np.array_equal(f, ref_f) and np.array_equal(r, ref_r)


@pytest.mark.parametrize("periods", [(0, 0, 1), (1, 1, 1)], ids=["z", "xyz"])
def test_config4_512cubed_over_rccl_self_neighbour(fpr, oracle, periodic_grid, periods):
    """BASELINE config 4's per-GPU workload (512^3 local array with neighbours) through the library's RCCL transport on
    the one card a box has: one rank that is its own periodic neighbour (z = the native slab layout; xyz = every face
    of a (2,2,2)-like rank), 4 single steps (GlobalGrid.step: boundary slabs, exchange beside the interior) and 2 fused
    pairs (GlobalGrid.step2: the core as one balanced launch on the compute stream, shell launches and both exchanges
    beside it on the comm stream) against the OpenMP oracle with wrapped halos: fields and residuals bit for bit, norms
    to 1e-13.  What this cannot cover is the link transport between two devices (N > 1 needs N GPUs)."""
    F = fpr
    n = (512, 512, 512)
    dims = tuple(d for d in range(3) if periods[d])
    gg = periodic_grid(n, periods)
    ext = [m - 2 if p else m for m, p in zip(n, periods)]
    dx, dy, dz = 10.0 / ext[0], 10.0 / ext[1], 10.0 / ext[2]
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    Ht = oracle.init_gaussian(n, dx, dy, dz, (4.0, 5.5, 6.0))
    Ht *= 1.0 + 0.25 * rnd(n, 77)
    wrap(Ht, dims=dims)
    A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
    gHt, gA, gB, gR = F.asdevice(Ht), F.asdevice(A), F.fzeros(*n), F.fzeros(*n)
    inner = (slice(1, -1),) * 3

    def faces_equal(loc, ref):
        for d in dims:   # halo planes (face interiors; edges / corners are not refreshed by one all-faces exchange)
            for side in (0, -1):
                idx = [slice(1, -1)] * 3
                idx[d] = side
                if not np.array_equal(loc[tuple(idx)], ref[tuple(idx)]):
                    return False
        return True

    sq = F.fzeros(1)
    for it in range(4):
        oracle.diffusion3d_step(Ht, A, B, R, *coef)
        wrap(B, dims=dims)
        A, B = B, A
        gg.step(gHt, gA, gB, gR, *coef, dt, sq)
        gA, gB = gB, gA
        ref = oracle.sumsq_scaled(R, dt)
        assert abs(float(sq.item()) - ref) <= 1e-13 * ref
    loc = F.tonumpy(gA)
    assert np.array_equal(loc[inner], A[inner]) and faces_equal(loc, A)
    assert np.array_equal(F.tonumpy(gR)[inner], R[inner])
    del loc
    gC = gA.clone()
    assert gg.can_step2(gHt, gA, gB, gC, gR)
    assert gg.reserve_cus() >= 12
    sq2 = F.fzeros(2)
    for it in range(2):
        refs = []
        for k in range(2):
            oracle.diffusion3d_step(Ht, A, B, R, *coef)
            wrap(B, dims=dims)
            A, B = B, A
            refs.append(oracle.sumsq_scaled(R, dt))
        gg.step2(gHt, gA, gB, gC, gR, *coef, dt, sq2)
        gA, gC = gC, gA
        got = sq2.cpu().tolist()
        assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs)), (got, refs)
        loc = F.tonumpy(gA)
        assert np.array_equal(loc[inner], A[inner]) and faces_equal(loc, A)
        del loc
        assert np.array_equal(F.tonumpy(gR)[inner], R[inner])
    if periods != (0, 0, 1):
        return
    # z-slabs at full size: TWO fused triples between ranks (GlobalGrid.step3: core planes [3, 509) as one launch of the three-step kernel on
    # 224 units -- 260 units of work, 36 of them in thin slices --, the shells in single-step launches on 6-plane slabs with three one-plane
    # exchanges) against six oracle steps with wrapped halos
    assert gg.can_step3(gHt, gA, gB, gR)
    sq3 = F.fzeros(3)
    for it in range(2):
        refs = []
        for k in range(3):
            oracle.diffusion3d_step(Ht, A, B, R, *coef)
            wrap(B, dims=dims)
            A, B = B, A
            refs.append(oracle.sumsq_scaled(R, dt))
        gg.step3(gHt, gA, gB, gR, *coef, dt, sq3)
        gA, gB = gB, gA
        got = sq3.cpu().tolist()
        assert all(abs(g - r) <= 1e-13 * r for g, r in zip(got, refs)), (got, refs)
        loc = F.tonumpy(gA)
        assert np.array_equal(loc[inner], A[inner]) and faces_equal(loc, A)
        del loc
        assert np.array_equal(F.tonumpy(gR)[inner], R[inner])
    assert F.ctx().L.fpr_comm_cus(F.ctx().h) == 32


def _run_bench(argv, timeout=600):
    """bench.py as the driver runs it: returns (process, full record from bench_detail.json, the compact LAST line of stdout parsed).
    The line must be the last line of stdout, strict JSON, below 4 KB (the driver keeps only the tail of stdout)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, FPR_BENCH_DETAIL_DIR=td)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout,
                           cwd=root, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0]
        assert len(lines[0]) < 4096

        def no_constants(x):
            raise ValueError(x)

        line = json.loads(lines[0], parse_constant=no_constants)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline"):
            assert k in line, k
        full = json.load(open(os.path.join(td, "bench_detail.json")))
    assert line["n_gpus"] == full["n_gpus"] and abs(line["value"] - full["value"]) <= 1e-5 * full["value"]
    return r, full, line


def test_bench_two_ranks_rehearsal_on_one_card():
    """bench.py's N>1 control flow between REAL processes on the one card a test box has: the self-launcher, the gloo
    control plane, identical collective counts on every rank through pre-warm / warm-up / timed region, the shell/core
    choreography of the fused pairs, one JSON line from rank 0.  Planes travel through the host (RCCL refuses two ranks
    on one device), so the rate it prints is not a measurement; the norm it prints is checked against a 1-rank run of the
    same global problem, `bench.py --as-one-rank-of 1,1,2` (run_all_benchmarks.sh:21-28 is the reference's multi-rank protocol)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-secondary", "--no-single-leg", "--prewarm-ms", "0"]
    r, out, line = _run_bench(["--gpus", "2", "--rehearse-shared-gpu", "--n", "128"] + common)
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["rehearsal"] is True and line["norm_check"]["ok"] is True
    assert out["n_gpus"] == 2 and out["config"]["process_grid"] == [1, 1, 2] and out["config"]["global_grid"] == [128, 128, 254]
    # z-slabs run THREE iterations per launch between ranks (fpr_diffusion3d_step3_halo): 12 steps = 4 core launches of the three-step
    # kernel, each with its chain of single-step launches on the two-plane shells
    assert "rehearsal" in out and out["legs"]["fused_triples"]["launches"] == 4
    assert 0.0 < out["roofline"]["frac"] <= 1.0 and out["roofline"]["iterations_per_launch"] == 3
    assert out["roofline"]["launches_by_kind"]["single_step_boxes"] >= 4 * 5       # per triple and face: 2 + 2 + 1 shell launches
    assert out["config"]["last_err"] is not None and 0.0 < out["config"]["last_err"] < 1.0
    # the same global problem (128 x 128 x 254, lz = 20) on ONE rank, same number of iterations: the sum of squares behind
    # the norm rank 0 printed must be the single-domain one (summation order differs: 1e-12)
    r1, one, _ = _run_bench(["--gpus", "1", "--as-one-rank-of", "1,1,2", "--n", "128"] + common)
    assert one["config"]["local_grid"] == [128, 128, 254] and one["config"]["global_grid"] == [128, 128, 254]
    a, b = out["config"]["last_sumsq"], one["config"]["last_sumsq"]
    assert a > 0 and abs(a - b) <= 1e-12 * b, (a, b)
    # ... and bench.py checked both itself against the committed control runs (tests/golden/scale_norms.json: the check the
    # driver's N > 1 runs carry in their line)
    for o in (out, one):
        nc = o["norm_check"]
        assert nc["ok"] is True and nc["key"] == "n128_dims1,1,2" and nc["iterations"] == o["config"]["iterations_since_start"] == 34
    assert out["config"]["choreography"] == "triples" and out["config"]["attempt"] == 1 and "first_attempt" not in out
    # fused PAIRS between the same two ranks (--no-fuse3: what every other process grid runs): same control
    rq, pairs, _ = _run_bench(["--gpus", "2", "--rehearse-shared-gpu", "--n", "128", "--no-fuse3"] + common)
    assert pairs["config"]["choreography"] == "pairs" and pairs["norm_check"]["ok"] is True and pairs["legs"]["fused_pairs"]["launches"] == 6
    assert pairs["roofline"]["launches_by_kind"]["fused_boxes"] >= 2 * 6       # per pair: the core launch + the shell launch(es)
    # the watchdog's fallback choreography (single steps, no split of the device) on the same problem: same norm
    rp, plain, pline = _run_bench(["--gpus", "2", "--rehearse-shared-gpu", "--n", "128", "--choreography", "plain"] + common)
    assert pline["config"]["choreography"] == "plain" and pline["norm_check"]["ok"] is True
    assert plain["config"]["choreography"] == "plain" and plain["norm_check"]["ok"] is True
    assert plain["roofline"]["launches_by_kind"]["core"] == 0


def test_bench_watchdog_ends_a_stalled_gpu_rank_and_the_plain_fallback_checks_its_norm():
    """The watchdog between REAL GPU processes: rank 1 stops after it has allocated its fields on the card (a deadlocked
    collective looks the same from outside); the supervisors end both workers, fresh ones run the plain choreography, and
    the line they print carries the first attempt's record and a norm that matches the single-rank control."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r, out, line = _run_bench(["--gpus", "2", "--rehearse-shared-gpu", "--n", "128", "--steps", "12", "--warmup", "4", "--prewarm-ms", "0",
                               "--no-cpu-baseline", "--no-secondary", "--dry-run-hang", "1", "--watchdog-s", "12"])
    assert line["config"]["attempt"] == 2 and line["config"]["choreography"] == "plain" and "no progress" in line["config"]["fallback_reason"]
    assert out["config"]["choreography"] == "plain" and out["config"]["attempt"] == 2
    assert out["first_attempt"]["choreography"] == "pairs" and "no progress" in out["first_attempt"]["reason"]
    assert out["norm_check"]["ok"] is True and out["norm_check"]["key"] == "n128_dims1,1,2"


def test_bench_four_ranks_rehearsal_x_y_z_decompositions_agree():
    """Four real processes on the one card (host-staged planes, see the two-rank test above) in the process grids (2,2,1),
    (2,1,2) and (1,2,2): faces in x, y and z through the shell/core choreography of the fused pairs, the narrow-box kernels
    and the pack kernels.  The three global problems are the same problem with its axes permuted (isotropic diffusion of a
    centred Gaussian), so the norm rank 0 prints after 12 iterations must not depend on the decomposition."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "4", "--rehearse-shared-gpu", "--n", "128", "--steps", "12", "--warmup", "4", "--no-cpu-baseline",
              "--no-secondary", "--no-single-leg", "--prewarm-ms", "0"]
    errs = []
    for dims in ("2,2,1", "2,1,2", "1,2,2"):
        r, out, line = _run_bench(["--dims", dims] + common)
        assert line["config"]["process_grid"] == [int(x) for x in dims.split(",")] and line["norm_check"]["ok"] is True
        assert out["n_gpus"] == 4 and out["config"]["process_grid"] == [int(x) for x in dims.split(",")]
        assert out["legs"]["fused_pairs"]["launches"] == 6 and 0.0 < out["roofline"]["frac"] <= 1.0
        assert out["norm_check"]["ok"] is True and out["norm_check"]["key"] == "n128_dims" + dims
        errs.append(out["config"]["last_err"])
    assert errs[0] is not None and 0.0 < errs[0] < 1.0
    assert abs(errs[1] - errs[0]) <= 1e-12 * errs[0] and abs(errs[2] - errs[0]) <= 1e-12 * errs[0]


@pytest.mark.parametrize("world", [2, 4])
def test_library_transport_between_processes_on_one_card(world, tmp_path):
    """csrc/comm.hip's exchange code EXECUTED between real processes: fpr_grid_init's neighbour table,
    post_group's receive-low-first / send-high-first order (both neighbours of a dimension the same rank; a rank that is its own
    neighbour beside real ones), the pack / unpack kernels, the three forms of the exchange, the norm's all-reduce and gather! --
    over the host-staged transport under post_group (fpr_comm_init_hosted; RCCL refuses two ranks on one device), against a numpy
    model of update_halo! (part1_kernel_programming.jl:182,187; ImplicitGlobalGrid, overlap 2) on every process grid of
    part1_scaling_experiments.jl:35-41 that `world` ranks allow.  tests/hosted_exchange_worker.py is one rank."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rdzv = str(tmp_path / "rdzv")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "hosted_exchange_worker.py"), str(r), str(world), rdzv],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=root) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=420)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, "rank %d: %s" % (r, e[-3000:])
        assert "rank %d of %d: %d grids OK" % (r, world, 8 if world == 2 else 4) in o


def test_bench_two_ranks_rehearsal_python_twin_of_the_choreography():
    """The rehearsals above run the LIBRARY's transport code and one-call pairs (--rehearse-transport hosted, the default since
    round 5); the Python twin of the choreography (grid.HaloExchanger over gloo, GlobalGrid.step2_begin / _middle / _end) stays
    covered too: same global problem, same norm."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r, out, _ = _run_bench(["--gpus", "2", "--rehearse-shared-gpu", "--rehearse-transport", "python", "--n", "128", "--steps", "12", "--warmup", "4",
                            "--no-cpu-baseline", "--no-secondary", "--no-single-leg", "--prewarm-ms", "0"])
    assert out["norm_check"]["ok"] is True and out["norm_check"]["key"] == "n128_dims1,1,2"
    assert "Python HaloExchanger" in out["config"]["halo"]


@pytest.mark.parametrize("dims,choreography", [("2,2,2", "pairs"), ("2,2,2", "plain"), ("1,1,8", "pairs"), ("1,1,8", "plain"), ("1,1,8", "triples")],
                         ids=["2x2x2-pairs", "2x2x2-plain", "1x1x8-pairs", "1x1x8-plain", "1x1x8-triples"])
def test_eight_ranks_as_threads_equal_the_single_domain_run(dims, choreography):
    """The two process grids the reference's scaling runs use at eight ranks (part1_scaling_experiments.jl:35-41: (2,2,2); z-slabs
    (1,1,8) is this repository's default), EXECUTED by the library's own exchange code and one-call pair choreography with eight
    ranks -- every rank a thread of ONE process with its own library context (bind_context), planes over fpr_comm_init_hosted and
    in-process queues (grid.ThreadWorld; tests/thread_ranks_worker.py is that process).  A GPU box admits six processes on its card, so
    eight real processes cannot run there.  Local grids 128^3, 12 pseudo-iterations; every rank's field and residual
    equal the single-domain run's slice bit for bit, the all-reduced sums of squares agree to 1e-12 (summation order)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "thread_ranks_worker.py"), dims, choreography], capture_output=True, text=True,
                       timeout=420, cwd=root)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "dims (%s) %s n 128: OK" % (dims.replace(",", ", "), choreography) in r.stdout
    assert r.stdout.count("field True residual True") == 8


def test_decomposed_solver_loop_with_triples_equals_the_plain_loop():
    """part1.diffusion_3D_kernel_programming on four ranks (threads of one process, z-slabs): its loop runs three iterations per launch
    between ranks (GlobalGrid.step3), all-reduces the three norms and, when the reference's loop (:179-192) would have ended inside a
    triple, replays the iterations up to there singly from the triple's intact input.  Against the same loop with one iteration per launch
    (options diff3_fuse3 = diff3_fuse2 = 0): the same iteration counts and errors per physical step, every rank's field bit for bit."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "thread_ranks_worker.py"), "1,1,4", "solver", "128", "0", "8e-3"],
                       capture_output=True, text=True, timeout=420, cwd=root)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "dims (1, 1, 4) solver n 128: OK" in r.stdout and r.stdout.count("field True comm units 32") == 4

