#!/usr/bin/env python3
"""Which earlier work in the process changes the timing of the side-by-side T / W solves of the NS step?"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd, bench
F = fpr_amd.load(0)
mg = F.multigrid
what = sys.argv[1] if len(sys.argv) > 1 else "none"
n = 4097
if what in ("l2", "cg", "jac8", "timer"):
    b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    x = F.fzeros(n, n)
    opt = mg.MGOpt()
    if what == "cg":
        opt.coarse_solve_size, opt.coarse_solver = 257, mg.conjugate_gradient
    if what == "jac8":
        opt.coarse_solve_size = 257
    if what == "timer":
        F.ctx().call("fpr_kernel_timer", 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mg.MGsolve_2DPoisson_(x, b, 1.0 / (n - 1), 0.0, 1e-6, 100, False, opt=opt)
    if what == "timer":
        F.ctx().call("fpr_kernel_timer", 0)
    F.synchronize()
    del b, x
r = bench.ns_block(F)
print(what, "side by side %.3f ms, in sequence %.3f ms" % (r["value"] * 1e3, r["solves_one_after_the_other_s_per_step"] * 1e3))
