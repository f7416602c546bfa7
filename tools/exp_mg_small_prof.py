#!/usr/bin/env python3
"""Where k_mg_small (the LDS-resident sub-hierarchy of the V-cycle) spends its time: wall_clock64 stamps of thread 0 at
the section borders of one launch (option mg_small_prof = device address of 32 int64), 100 MHz ticks -> microseconds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65
css = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cc = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # the c of (lap - c) u = f (NS solves: 1 / (beta dt) ~ 5e6)
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
opt = mg.MGOpt()
opt.coarse_solve_size = css
prof = torch.zeros(32, dtype=torch.int64, device=x.device)
F.ctx().set_option("mg_small_prof", prof.data_ptr())
for rep in range(3):
    x.zero_()
    r = mg.Vcycle_2DPoisson_(x, b, 1.0 / (n - 1), cc, 1e-6, css, mg.jacobi, mg.parallel, False)
    F.synchronize()
    t = prof.cpu().numpy()
    k = int((t != 0).sum())
    d = [(t[i + 1] - t[i]) / 100.0 for i in range(k - 1)]
    cit = F.ctx().L.fpr_last_coarse_iters(F.ctx().h)
    print("n=%d css=%d: total %.2f us, %d coarse iterations; sections (us): %s" % (n, css, (t[k - 1] - t[0]) / 100.0, cit, " ".join("%.2f" % v for v in d)))
    prof.zero_()
F.ctx().set_option("mg_small_prof", 0)
