#!/usr/bin/env python3
"""Where a group of k_jacobi_persist spends its time (option mg_jacp_prof = address of a device buffer of 8 int64 per workgroup):
wall_clock64 ticks (100 MHz) of thread 0 waiting for its neighbours' flags / loading the region + first LDS image / sweeping /
storing the tile, draining, summing / publishing.  Five-level V-cycle 4097^2, coarse 257^2."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
ctx = F.ctx()
n = 4097
h = 1.0 / (n - 1)
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
opt = mg.MGOpt()
opt.coarse_solve_size, opt.coarse_solver = 257, mg.jacobi
variants = [tuple(int(v) for v in q.split(",")) for q in sys.argv[1].split(";")] if len(sys.argv) > 1 else [(8, 2, 0), (8, 2, 1), (7, 2, 1), (7, 1, 1)]
for ps, py, *rest in variants:
    tagged = rest[0] if rest else 1
    ctx.set_option("mg_jacp_tagged", tagged)
    ctx.set_option("mg_patch_sweeps", ps)
    ctx.set_option("mg_jacp_py", py)
    prof = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
    for it in range(2):
        x.zero_()
        prof.zero_()
        F.synchronize()
        ctx.set_option("mg_jacp_prof", prof.data_ptr() if it == 1 else 0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 1, False, opt=opt, return_history=True)
        F.synchronize()
    ctx.set_option("mg_jacp_prof", 0)
    p = prof.cpu().view(512, 8).double()
    p = p[p[:, 5] > 0]
    g = p[:, 5:6]
    per = p[:, :5] / g * 10.0 / 1e3        # us per group
    names = ["wait", "load", "sweeps", "store+drain+sum", "publish"]
    print("tagged %d " % tagged, end="")
    print("sweeps/group %d patch rows %d: %d workgroups, %d groups each; us per group (mean / min / max over workgroups):" % (ps, py, p.shape[0], int(g[0])))
    for k, nm in enumerate(names):
        print("   %-16s %.2f / %.2f / %.2f" % (nm, per[:, k].mean(), per[:, k].min(), per[:, k].max()))
    print("   %-16s %.2f" % ("sum", per.sum(1).mean()))
    for xcc in range(8):
        m = p[:, 6] == xcc
        if m.any():
            print("   XCC %d: %3d workgroups, wait %.2f sweeps %.2f" % (xcc, int(m.sum()), per[m, 0].mean(), per[m, 2].mean()))
ctx.set_option("mg_patch_sweeps", 0)
ctx.set_option("mg_jacp_py", 0)
