#!/usr/bin/env python3
"""Where k_mid_down / k_mid_up spend their time: wall_clock64 stamps of thread 0 of one interior workgroup at the phase
borders (option mg_mid_prof = device address of 32 int64: down in [0, 16), up in [16, 32)), 100 MHz ticks -> microseconds.
usage: exp_mid_prof.py [levels 2|3] [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4097
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
prof = torch.zeros(32, dtype=torch.int64, device=x.device)
c = F.ctx()
c.set_option("mg_mid2", 1)
c.set_option("mg_mid_prof_levels", nl)
c.set_option("mg_mid_prof", prof.data_ptr())
for rep in range(3):
    x.zero_()
    r = mg.Vcycle_2DPoisson_(x, b, 1.0 / (n - 1), 0.0, 1e-6, 5, mg.jacobi, mg.parallel, False)
    F.synchronize()
    t = prof.cpu().numpy()
    for name, lo in (("down", 0), ("up", 16)):
        tt = t[lo:lo + 16]
        k = int((tt != 0).sum())
        d = [(tt[i + 1] - tt[i]) / 100.0 for i in range(k - 1)]
        print("levels=%d %s: total %.2f us; sections (us): %s" % (nl, name, (tt[k - 1] - tt[0]) / 100.0 if k else 0.0, " ".join("%.2f" % v for v in d)))
    prof.zero_()
c.set_option("mg_mid_prof", 0)
