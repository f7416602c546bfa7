"""GPU tests of the multi-rank building blocks on ONE device: HIP halo pack/unpack kernels against the
plane convention used by the CPU (gloo) tests, and the full boundary-slab / exchange / interior
choreography of GlobalGrid.step with an in-process fake of torch.distributed (two or more emulated
ranks on one GPU; the real RCCL transport needs N GPUs and is only run by the driver)."""
import numpy as np
import pytest

from fixtures_io import splitmix64_uniform
from oracle.oracle import asf, farr

pytestmark = pytest.mark.gpu


def rnd(shape, seed):
    return asf(splitmix64_uniform(int(np.prod(shape)), seed).reshape(shape, order="F"))


@pytest.mark.parametrize("shape", [(10, 9, 8), (64, 5, 7), (33, 34, 35)], ids=str)
def test_pack_unpack_planes(fpr, shape):
    F = fpr
    nx, ny, nz = shape
    A = rnd(shape, 1)
    gA = F.asdevice(A)
    c = F.ctx()
    for face in range(6):
        d, side = face >> 1, face & 1
        n = shape[d]
        plane = A.take(n - 2 if side else 1, axis=d)
        buf = F.fzeros(plane.size)
        c.call("fpr_halo_pack3d", F._lib.fptr(gA, 3), nx, ny, nz, face, buf.data_ptr(), 0)
        assert np.array_equal(buf.cpu().numpy(), plane.ravel(order="F"))
        B = F.asdevice(A)
        newp = rnd(plane.shape, 7 + face)
        c.call("fpr_halo_unpack3d", F._lib.fptr(B, 3), nx, ny, nz, face, F.asdevice(newp.ravel(order="F")).data_ptr(), 0)
        ref = A.copy(order="F")
        idx = [slice(None)] * 3
        idx[d] = n - 1 if side else 0
        ref[tuple(idx)] = newp
        assert np.array_equal(F.tonumpy(B), ref)


class FakeDist:
    """In-process stand-in for torch.distributed P2P: sends are device copies into a mailbox, receives
    are completed in Work.wait().  All emulated ranks must post (step_begin) before anyone waits."""

    class P2POp:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    isend, irecv = "isend", "irecv"

    class Work:
        def __init__(self, fn):
            self.fn = fn

        def wait(self):
            self.fn()

    def __init__(self, mail, rank):
        self.mail, self.rank = mail, rank

    def batch_isend_irecv(self, ops):
        works = []
        for o in ops:
            if o.op == "isend":
                self.mail.setdefault((self.rank, o.peer), []).append(o.tensor.clone())
                works.append(FakeDist.Work(lambda: None))
            else:
                def fin(o=o):
                    o.tensor.copy_(self.mail[(o.peer, self.rank)].pop(0))
                works.append(FakeDist.Work(fin))
        return works


@pytest.mark.parametrize("dims,n", [((1, 1, 2), (20, 12, 10)), ((2, 2, 1), (12, 10, 9)), ((2, 2, 2), (10, 10, 10)),
                                     ((1, 1, 3), (64, 20, 9))], ids=str)
def test_step_choreography_emulated_ranks(fpr, oracle, dims, n):
    F = fpr
    world = int(np.prod(dims))
    nx, ny, nz = n
    ng = tuple(d * (m - 2) + 2 for d, m in zip(dims, n))
    lx, ly, lz = (d * 10.0 for d in dims)
    dx, dy, dz = lx / ng[0], ly / ng[1], lz / ng[2]
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    mail = {}
    ranks = []
    for r in range(world):
        gg = F.grid.GlobalGrid(nx, ny, nz, dims=(1, 1, 1), use_dist=False)
        gg.dims, gg.nprocs, gg.me = dims, world, r
        gg.coords = gg.coords_of(r)
        gg.neighbors = {}
        for d in range(3):
            for side in (0, 1):
                cc = list(gg.coords)
                cc[d] += 1 if side else -1
                if 0 <= cc[d] < dims[d]:
                    gg.neighbors[2 * d + side] = tuple(cc)
        gg.dist = FakeDist(mail, r)
        Ht = F.asdevice(oracle.init_gaussian(n, dx, dy, dz, (lx / 2, ly / 2, lz / 2), gg.coords))
        ranks.append(dict(gg=gg, Ht=Ht, A=Ht.clone(), B=F.fzeros(*n), R=F.fzeros(*n), sq=F.fzeros(1)))
    # single-domain oracle
    Hg = oracle.init_gaussian(ng, dx, dy, dz, (lx / 2, ly / 2, lz / 2))
    Ag, Bg, Rg = Hg.copy(order="F"), farr(*ng), farr(*ng)
    for it in range(5):
        states = [s["gg"].step_begin(s["Ht"], s["A"], s["B"], s["R"], *coef, dt, s["sq"]) for s in ranks]
        for s, st in zip(ranks, states):
            s["gg"].step_end(st)
            s["A"], s["B"] = s["B"], s["A"]
        oracle.diffusion3d_step(Hg, Ag, Bg, Rg, *coef)
        Ag, Bg = Bg, Ag
        tot = sum(float(s["sq"].item()) for s in ranks)
        ref = oracle.sumsq_scaled(Rg, dt)
        assert abs(tot - ref) <= 1e-13 * ref
    assert all(len(v) == 0 for v in mail.values())
    for r, s in enumerate(ranks):
        c = s["gg"].coords
        off = tuple(ci * (m - 2) for ci, m in zip(c, n))
        loc = F.tonumpy(s["A"])
        glob = Ag[off[0]:off[0] + nx, off[1]:off[1] + ny, off[2]:off[2] + nz]
        assert np.array_equal(loc[1:-1, 1:-1, 1:-1], glob[1:-1, 1:-1, 1:-1])
        for face in s["gg"].neighbors:
            d, side = face >> 1, face & 1
            idx = [slice(1, -1)] * 3
            idx[d] = -1 if side else 0
            assert np.array_equal(loc[tuple(idx)], glob[tuple(idx)])


def _emulated_ranks(F, oracle, dims, n, mail):
    world = int(np.prod(dims))
    ng = tuple(d * (m - 2) + 2 for d, m in zip(dims, n))
    lx, ly, lz = (d * 10.0 for d in dims)
    dx, dy, dz = lx / ng[0], ly / ng[1], lz / ng[2]
    ranks = []
    for r in range(world):
        gg = F.grid.GlobalGrid(*n, dims=(1, 1, 1), use_dist=False)
        gg.dims, gg.nprocs, gg.me = dims, world, r
        gg.coords = gg.coords_of(r)
        gg.neighbors = {}
        for d in range(3):
            for side in (0, 1):
                cc = list(gg.coords)
                cc[d] += 1 if side else -1
                if 0 <= cc[d] < dims[d]:
                    gg.neighbors[2 * d + side] = tuple(cc)
        gg.dist = FakeDist(mail, r)
        Ht = F.asdevice(oracle.init_gaussian(n, dx, dy, dz, (lx / 2, ly / 2, lz / 2), gg.coords))
        ranks.append(dict(gg=gg, Ht=Ht))
    return ranks, ng, (dx, dy, dz), (lx, ly, lz)


@pytest.mark.parametrize("dims,n", [((1, 1, 2), (128, 18, 10)), ((1, 1, 3), (130, 33, 9)), ((1, 1, 4), (128, 16, 8)),
                                     ((2, 1, 1), (128, 18, 10)), ((1, 2, 1), (128, 20, 9)), ((2, 2, 1), (130, 18, 9)),
                                     ((2, 2, 2), (128, 18, 10)), ((3, 1, 2), (128, 16, 9)), ((1, 3, 1), (256, 34, 8))], ids=str)
def test_fused_pair_choreography_emulated_ranks(fpr, oracle, dims, n):
    """GlobalGrid.step2 (two iterations per fused launch, level-1 halo cells exchanged in between) on emulated ranks --
    z-slabs and the reference's x / y / xyz decompositions (part1_scaling_experiments.jl:35-41) -- equals the
    single-domain oracle bit for bit, including both per-iteration norms."""
    F = fpr
    mail = {}
    ranks, ng, (dx, dy, dz), (lx, ly, lz) = _emulated_ranks(F, oracle, dims, n, mail)
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    for s in ranks:
        s.update(A=s["Ht"].clone(), O=F.fzeros(*n), C=s["Ht"].clone(), R=F.fzeros(*n), sq=F.fzeros(2))
        assert s["gg"].can_step2(s["Ht"], s["A"], s["O"], s["C"], s["R"])
    Hg = oracle.init_gaussian(ng, dx, dy, dz, (lx / 2, ly / 2, lz / 2))
    Ag, Bg, Rg = Hg.copy(order="F"), farr(*ng), farr(*ng)
    for pair in range(3):
        sts = [s["gg"].step2_begin(s["Ht"], s["A"], s["O"], s["C"], s["R"], *coef, dt, s["sq"]) for s in ranks]
        for s, st in zip(ranks, sts):
            s["gg"].step2_middle(st)
        for s, st in zip(ranks, sts):
            s["gg"].step2_end(st)
            s["A"], s["C"] = s["C"], s["A"]
        refs = []
        for _ in range(2):
            oracle.diffusion3d_step(Hg, Ag, Bg, Rg, *coef)
            Ag, Bg = Bg, Ag
            refs.append(oracle.sumsq_scaled(Rg, dt))
        for j in range(2):
            tot = sum(float(s["sq"][j].item()) for s in ranks)
            assert abs(tot - refs[j]) <= 1e-13 * refs[j]
    assert all(len(v) == 0 for v in mail.values())
    nx, ny, nz = n
    for s in ranks:
        c = s["gg"].coords
        off = tuple(ci * (m - 2) for ci, m in zip(c, n))
        loc = F.tonumpy(s["A"])
        glob = Ag[off[0]:off[0] + nx, off[1]:off[1] + ny, off[2]:off[2] + nz]
        assert np.array_equal(loc[1:-1, 1:-1, 1:-1], glob[1:-1, 1:-1, 1:-1])
        assert np.array_equal(F.tonumpy(s["R"])[1:-1, 1:-1, 1:-1], Rg[off[0]:off[0] + nx, off[1]:off[1] + ny, off[2]:off[2] + nz][1:-1, 1:-1, 1:-1])
        for face in s["gg"].neighbors:
            d, side = face >> 1, face & 1
            idx = [slice(1, -1)] * 3
            idx[d] = -1 if side else 0
            assert np.array_equal(loc[tuple(idx)], glob[tuple(idx)])


def test_device_split_streams_and_small_ops(fpr):
    """fpr_reserve_comm_cus / fpr_comm_cus / fpr_stream_handle / fpr_fill_on / fpr_add_on: the comm and core streams of a split
    device are library-owned, ordered against the compute stream by fpr_stream_wait, and a split that is not the same share of every XCD is
    refused (k must be a multiple of 8; within an XCD the core launch of a pair copes with any split: it takes tickets)."""
    import ctypes as C
    import torch

    F = fpr
    c = F.ctx()
    c.reserve_comm_cus(0)                    # (earlier tests with fused pairs between ranks leave the device split)
    try:
        with pytest.raises(F.FprError):
            c.reserve_comm_cus(12)           # not the same number of units out of every XCD
        assert c.L.fpr_comm_cus(c.h) == 0
        h0 = [C.c_void_p() for _ in range(3)]
        for s in range(3):
            c.call("fpr_stream_handle", s, C.byref(h0[s]))
        assert h0[2].value == h0[0].value and h0[1].value != h0[0].value      # unsplit: the core stream IS the compute stream
        c.reserve_comm_cus(16)                   # core stream on every unit; the comm stream's 16 units found by a probe launch
        assert c.L.fpr_comm_cus(c.h) in (16, 32)  # (32: the probe did not find them, masked core stream instead)
        if c.L.fpr_comm_cus(c.h) == 16:
            assert c.get_option("comm_units_found") == 16
        c.reserve_comm_cus(32)
        assert c.L.fpr_comm_cus(c.h) == 32
        h = [C.c_void_p() for _ in range(3)]
        for s in range(3):
            c.call("fpr_stream_handle", s, C.byref(h[s]))
        assert h[0].value == h0[0].value and len({h[0].value, h[1].value, h[2].value}) == 3
        a, b = F.fzeros(6), F.fzeros(6)
        a.fill_(2.0)                                     # compute stream
        c.call("fpr_stream_wait", 1, 0)
        c.call("fpr_fill_on", b.data_ptr(), 1.5, 6, 1)   # comm stream
        c.call("fpr_add_on", b.data_ptr(), a.data_ptr(), 6, 1)
        c.call("fpr_stream_wait", 2, 1)
        c.call("fpr_add_on", a.data_ptr(), b.data_ptr(), 6, 2)   # core stream: 2 + 3.5
        c.call("fpr_stream_wait", 0, 2)
        assert torch.equal(a.cpu(), torch.full((6,), 5.5, dtype=torch.float64)) and torch.equal(b.cpu(), torch.full((6,), 3.5, dtype=torch.float64))
    finally:
        c.reserve_comm_cus(0)
    assert c.L.fpr_comm_cus(c.h) == 0
