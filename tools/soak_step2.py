"""Soak of the fused-pair choreography between ranks (one rank, periodic = its own neighbour over the library's RCCL transport):
P pairs chained on the core / comm streams of the split device (step2(join=False), what bench.py runs) against the same P pairs
with a device-wide synchronisation after every phase of every pair (nothing overlaps anything; the phased form keeps the x-shell
in the field, the chained one-call form in compact strips).  Interior cells, the interiors of the halo planes and the residuals must
agree bit for bit (edge and corner halo cells are refreshed by neither form's contract), all 2P norms to 1e-13 (the strips sum
their cells in another order); repeated R times.  usage: soak_step2.py <n> <periods e.g. 001> <pairs> <repeats> [drop_faces]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]); periods = tuple(int(c) for c in sys.argv[2]); P = int(sys.argv[3]); R = int(sys.argv[4])
drop = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=periods, transport="rccl", use_dist=False, drop_faces=drop)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((4., 5.5, 6.), dx, dx, dx, Ht)
Ht.mul_(1.0 + 0.001 * torch.arange(n, device=Ht.device, dtype=torch.float64).reshape(n, 1, 1))
gg.update_halo_(Ht)
A0 = Ht.clone()

def run(serial):
    A, O, C, Rr = A0.clone(), F.fzeros(n, n, n), A0.clone(), F.fzeros(n, n, n)
    sq = F.fzeros(2 * P)
    for p in range(P):
        if serial:
            st = gg.step2_begin(Ht, A, O, C, Rr, *coef, 0.2, sq[2 * p:2 * p + 2]); torch.cuda.synchronize()
            gg.step2_middle(st); torch.cuda.synchronize()
            gg.step2_end(st, True); torch.cuda.synchronize()
        else:
            gg.step2(Ht, A, O, C, Rr, *coef, 0.2, sq[2 * p:2 * p + 2], join=False)
        A, C = C, A
    gg.join(); torch.cuda.synchronize()
    return A, Rr, sq

def same(a, b):
    inner = (slice(1, -1),) * 3
    ok = torch.equal(a[inner], b[inner])
    for d in range(3):
        for side in (0, -1):
            idx = [slice(1, -1)] * 3
            idx[d] = side
            ok = ok and torch.equal(a[tuple(idx)], b[tuple(idx)])
    return ok

ref = run(True)
bad = 0
t0 = time.time()
for r in range(R):
    got = run(False)
    ok = same(got[0], ref[0]) and same(got[1], ref[1]) and bool(((got[2] - ref[2]).abs() <= 1e-13 * ref[2].abs()).all())
    bad += not ok
print("n=%d periods=%s drop=%d: %d pairs chained x %d repeats against the fully serialised run: %d mismatching repeats (%.1f s); last norm %.6e"
      % (n, sys.argv[2], drop, P, R, bad, time.time() - t0, float(ref[2][-1])))
F.grid.finalize_global_grid()
sys.exit(1 if bad else 0)
