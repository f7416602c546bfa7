"""One rank of tests/test_gpu_rccl.py::test_library_transport_between_two_processes_on_one_card (started as a child process).

Every rank uses cuda:0.  The library's exchange code (csrc/comm.hip: fpr_grid_init's neighbour table, post_group's
receive-low-first / send-high-first order, the pack / unpack kernels, fpr_halo_exchange3d[_begin/_end/_comm], fpr_allreduce_sum_dev,
fpr_gather3d) runs over the host-staged transport (fpr_comm_init_hosted, bytes through torch.distributed / gloo) and is checked
against what update_halo! must produce: every halo cell that has a neighbour holds the value of the global cell it stands for
(ImplicitGlobalGrid, overlap 2), edges and corners included; halo cells at a physical boundary stay untouched.

usage: hosted_exchange_worker.py RANK WORLD RDZV_FILE"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def start_array(gg, coords, n, periods, G):
    """Rank `coords`' array before the exchange: cells that stand for a global cell hold its value; every plane that is the halo
    of a face with a neighbour holds -7 throughout."""
    idx = []
    for d in range(3):
        ng = G.shape[d]
        i = coords[d] * (n[d] - 2) + np.arange(n[d]) - (1 if periods[d] else 0)
        idx.append(i % ng if periods[d] else np.clip(i, 0, ng - 1))
    A = np.asfortranarray(G[np.ix_(*idx)].copy())
    for d in range(3):
        for side in (0, 1):
            if neighbour(gg, coords, d, side, periods) is not None:
                sl = [slice(None)] * 3
                sl[d] = 0 if side == 0 else n[d] - 1
                A[tuple(sl)] = -7.0
    return A


def neighbour(gg, coords, d, side, periods):
    c = list(coords)
    c[d] += 1 if side else -1
    if 0 <= c[d] < gg.dims[d]:
        return tuple(c)
    if periods[d]:
        c[d] %= gg.dims[d]
        return tuple(c)
    return None


def simulate_update_halo(gg, n, periods, G, sequential=True):
    """update_halo! on every rank in numpy (ImplicitGlobalGrid, overlap 2): per dimension the low halo plane receives the low
    neighbour's last interior plane and the high halo plane the high neighbour's first interior plane, WHOLE planes; dimension by
    dimension (x, y, z), each on the result of the one before (sequential), or all faces at once from the start arrays."""
    ranks = [gg.coords_of(r) for r in range(gg.nprocs)]
    A = {c: start_array(gg, c, n, periods, G) for c in ranks}
    for d in range(3):
        src = {c: a.copy(order="F") for c, a in A.items()} if sequential else None
        if not sequential and d == 0:
            src0 = {c: a.copy(order="F") for c, a in A.items()}
        S = src if sequential else src0
        for c in ranks:
            for side in (0, 1):
                nb = neighbour(gg, c, d, side, periods)
                if nb is None:
                    continue
                dst, frm = [slice(None)] * 3, [slice(None)] * 3
                dst[d] = 0 if side == 0 else n[d] - 1
                frm[d] = n[d] - 2 if side == 0 else 1
                A[c][tuple(dst)] = S[nb][tuple(frm)]
    return A


def main():
    rank, world, rdzv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import fpr_amd

    F = fpr_amd.load(0)
    n = (20, 18, 14)
    checked = 0
    cases = [((2, 1, 1), (1, 0, 0)), ((2, 1, 1), (0, 0, 0)), ((1, 2, 1), (0, 1, 0)), ((1, 2, 1), (0, 0, 0)), ((1, 1, 2), (0, 0, 1)),
             ((1, 1, 2), (0, 0, 0)), ((2, 1, 1), (1, 1, 1)), ((1, 1, 2), (1, 1, 1))] if world == 2 else \
            [((2, 2, 1), (1, 1, 0)), ((2, 1, 2), (0, 0, 0)), ((1, 2, 2), (1, 1, 1)), ((2, 2, 1), (0, 1, 1))]
    for dims, periods in cases:
        gg = F.grid.GlobalGrid(*n, dims=dims, periods=periods, transport="hosted")
        assert gg.transport_kind == "rccl" and gg.hosted and F.ctx().L.fpr_comm_size(F.ctx().h) == world
        ng = (gg.nx_g(), gg.ny_g(), gg.nz_g())
        # with non-periodic dims the global array has the two physical boundary cells: shift by the halo
        Gn = tuple(d * (m - 2) + (0 if p else 2) for d, m, p in zip(dims, n, periods))
        assert Gn == ng
        rng = np.random.default_rng(1234)
        G = rng.random(Gn)
        A0 = start_array(gg, gg.coords, n, periods, G)
        E = simulate_update_halo(gg, n, periods, G)[gg.coords]
        A = F.asdevice(A0)
        gg.update_halo_(A)                      # fpr_halo_exchange3d: x, then y, then z (edges and corners consistent)
        F.synchronize()
        got = F.tonumpy(A)
        assert np.array_equal(got, E), (dims, periods, rank, np.argwhere(got != E)[:5])
        assert (E == -7.0).sum() == 0 or not all(periods)      # (fully periodic: every halo cell, edges and corners included, was refreshed)
        # the split forms: begin / end (all faces at once: edges and corners NOT refreshed) and the comm-stream form
        E1 = simulate_update_halo(gg, n, periods, G, sequential=False)[gg.coords]
        for form in ("begin_end", "comm"):
            B = F.asdevice(A0)
            c = F.ctx()
            if form == "begin_end":
                c.call("fpr_halo_exchange3d_begin", F._lib.fptr(B, 3), *n, 63)
                c.call("fpr_halo_exchange3d_end", F._lib.fptr(B, 3), *n, 63)
            else:
                c.call("fpr_stream_wait", 1, 0)
                c.call("fpr_halo_exchange3d_comm", F._lib.fptr(B, 3), *n, 63)
                c.call("fpr_stream_wait", 0, 1)
            F.synchronize()
            gb = F.tonumpy(B)
            # face cells proper (not on an edge of the local box) of every face with a neighbour; what the edges hold after an
            # all-at-once exchange depends on the order the planes are unpacked in and is read by no 7-point stencil
            for d in range(3):
                for side in (0, 1):
                    if 2 * d + side not in gg.neighbors:
                        continue
                    sl = [slice(1, n[0] - 1), slice(1, n[1] - 1), slice(1, n[2] - 1)]
                    sl[d] = 0 if side == 0 else n[d] - 1
                    assert np.array_equal(gb[tuple(sl)], E1[tuple(sl)]), (form, dims, periods, rank, d, side)
                    assert not (gb[tuple(sl)] == -7.0).any()
        # the norm's all-reduce and gather!
        t = torch.full((3,), float(rank + 1), dtype=torch.float64, device="cuda")
        gg.allreduce_(t)
        F.synchronize()
        assert t.tolist() == [world * (world + 1) / 2.0] * 3
        Ag = np.zeros((n[0] * dims[0], n[1] * dims[1], n[2] * dims[2]), order="F") if rank == 0 else None
        F.ctx().call("fpr_gather3d", F._lib.fptr(A, 3), *n, Ag.ctypes.data if rank == 0 else None)
        if rank == 0:
            blk = Ag[:n[0], :n[1], :n[2]]
            assert np.array_equal(blk, got)
        dist.barrier()
        F.grid.finalize_global_grid()
        checked += 1
    print("hosted_exchange_worker rank %d of %d: %d grids OK" % (rank, world, checked), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
