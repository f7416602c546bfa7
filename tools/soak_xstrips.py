"""Random shapes / face sets: the x-shell in compact strips (default) against the field form (diff3_xstrips = 0), chained pairs and a
joined pair, interior + halo-plane interiors + residual bit for bit, norms to 1e-13.  Row-tile and plane-chunk edges on purpose
(ny - 2 around multiples of 62, nz - 2 around multiples of 8).  usage: soak_xstrips.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fpr_amd
F = fpr_amd.load(0)
c = F.ctx()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
NY = [16, 17, 24, 63, 64, 65, 66, 125, 126, 127, 128, 190]
NZ = [8, 9, 10, 11, 16, 17, 18, 19, 26, 33, 40]
bad = 0
for case in range(ncases):
    nx = rng.choice([128, 130, 132, 192, 256, 258])
    ny, nz = rng.choice(NY), rng.choice(NZ)
    periods = (1, rng.randint(0, 1), rng.randint(0, 1))
    drop = 0
    if rng.random() < 0.4:
        drop = rng.choice([0b000001, 0b000010])          # one x-face only
        if periods[1] and rng.random() < 0.5: drop |= rng.choice([0b000100, 0b001000])
        if periods[2] and rng.random() < 0.5: drop |= rng.choice([0b010000, 0b100000])
    n = (nx, ny, nz)
    dx = 10.0 / nx
    coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
    g = torch.Generator(device="cpu"); g.manual_seed(1000 + case)
    H0 = torch.rand(nz, ny, nx, generator=g, dtype=torch.float64).permute(2, 1, 0).contiguous().permute(0, 1, 2)
    Ht0 = F.asdevice(np.asfortranarray(H0.numpy()))
    out = {}
    ok = True
    try:
        for strips in (1, 0):
            gg = F.grid.GlobalGrid(*n, dims=(1, 1, 1), periods=periods, transport="rccl", use_dist=False, drop_faces=drop)
            c.set_option("diff3_xstrips", strips)
            try:
                Ht, A, B, R = Ht0.clone(), Ht0.clone(), F.fzeros(*n), F.fzeros(*n)
                Cc = A.clone()
                if not gg.can_step2(Ht, A, B, Cc, R):
                    out = None
                    break
                sq = F.fzeros(10)
                for p in range(4):
                    gg.step2(Ht, A, B, Cc, R, *coef, 0.2, sq[2 * p:2 * p + 2], join=False); A, Cc = Cc, A
                gg.join()
                gg.step2(Ht, A, B, Cc, R, *coef, 0.2, sq[8:10], join=True); A, Cc = Cc, A
                out[strips] = (F.tonumpy(A), F.tonumpy(R), sq.cpu().numpy())
            finally:
                c.set_option("diff3_xstrips", 1)
                F.grid.finalize_global_grid()
        if out is None:
            print("case %d n=%s: no fused pairs at this size" % (case, n)); continue
        inner = (slice(1, -1),) * 3
        ok = np.array_equal(out[1][0][inner], out[0][0][inner]) and np.array_equal(out[1][1][inner], out[0][1][inner])
        for d in range(3):
            for side in (0, -1):
                idx = [slice(1, -1)] * 3; idx[d] = side
                ok = ok and np.array_equal(out[1][0][tuple(idx)], out[0][0][tuple(idx)])
        ok = ok and bool(np.all(np.abs(out[1][2] - out[0][2]) <= 1e-13 * np.abs(out[0][2])))
    except Exception as e:
        ok = False
        print("case %d n=%s periods=%s drop=%d: %r" % (case, n, periods, drop, e))
    bad += not ok
    print("case %2d n=%-15s periods=%s drop=%2d: %s" % (case, n, periods, drop, "equal" if ok else "MISMATCH"))
print("%d cases, %d mismatching" % (ncases, bad))
sys.exit(1 if bad else 0)
