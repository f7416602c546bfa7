"""Residual histories of the three multigrid solves of Navier-Stokes steps at 2049^2 (for the seam pass's prediction of the last cycle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0)
p2 = F.part2
opt = p2.SimIn_t()
opt.nx = opt.ny = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
tr = []
p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=int(sys.argv[2]) if len(sys.argv) > 2 else 8, fused=True, trace=tr)
for i, rec in enumerate(tr):
    for name in ("S", "T", "W"):
        r = rec[name]
        h = [x / r["f_rms"] for x in r["history"]]
        rates = [h[k] / h[k - 1] for k in range(1, len(h))]
        print("step %d %s: tol %.1e cycles %d rel %s rates %s" % (i, name, opt.tol, len(h), " ".join("%.2e" % x for x in h), " ".join("%.3f" % x for x in rates)))
