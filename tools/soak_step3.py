"""Soak of the fused-triple choreography between ranks of z-slabs (one rank, periodic in z = its own neighbour over the library's RCCL
transport): T triples chained on the core / comm streams of the split device (GlobalGrid.step3(join=False), what bench.py runs at N > 1)
against 3T single steps with a device-wide synchronisation after each (nothing overlaps anything).  Field (halo planes included),
residual bit for bit, all 3T norms to 1e-13; repeated R times.  usage: soak_step3.py <n> <triples> <repeats>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]); T = int(sys.argv[2]); R = int(sys.argv[3])
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=(0, 0, 1), transport="rccl", use_dist=False)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((4., 5.5, 6.), dx, dx, dx, Ht)
Ht.mul_(1.0 + 0.001 * torch.arange(n, device=Ht.device, dtype=torch.float64).reshape(n, 1, 1))
gg.update_halo_(Ht)
A0 = Ht.clone()


def run(serial):
    A, B, Rr = A0.clone(), F.fzeros(n, n, n), F.fzeros(n, n, n)
    sq = F.fzeros(3 * T)
    assert gg.can_step3(Ht, A, B, Rr)
    for t in range(T):
        if serial:
            for k in range(3):
                gg.step(Ht, A, B, Rr, *coef, 0.2, sq[3 * t + k:3 * t + k + 1]); torch.cuda.synchronize()
                A, B = B, A
        else:
            gg.step3(Ht, A, B, Rr, *coef, 0.2, sq[3 * t:3 * t + 3], join=False)
            A, B = B, A
    gg.join(); torch.cuda.synchronize()
    return A, Rr, sq


ref = run(True)
bad = 0
t0 = time.time()
inner = (slice(1, -1),) * 3
for r in range(R):
    got = run(False)
    ok = torch.equal(got[0][1:-1, 1:-1, :], ref[0][1:-1, 1:-1, :]) and torch.equal(got[1][inner], ref[1][inner]) and \
        bool(((got[2] - ref[2]).abs() <= 1e-13 * ref[2].abs()).all())
    bad += not ok
print("n=%d periodic z: %d triples chained x %d repeats against %d synchronised single steps: %d mismatching repeats (%.1f s); comm units %d; last norm %.6e"
      % (n, T, R, 3 * T, bad, time.time() - t0, F.ctx().L.fpr_comm_cus(F.ctx().h), float(ref[2][-1])))
F.grid.finalize_global_grid()
sys.exit(1 if bad else 0)
