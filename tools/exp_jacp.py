#!/usr/bin/env python3
"""k_jacobi_persist variants on the five-level V-cycle (4097^2, coarse 257^2, 5140 damped-Jacobi sweeps per cycle): sweeps per group
(option mg_patch_sweeps 8 | 7) x rows of a thread's register patch (mg_jacp_py 2 | 1).  Event time per launch / per sweep, wall time per
V-cycle, and the residual history against the default's.   usage: exp_jacp.py [cycles]"""
import ctypes as C
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variants = [tuple(int(v) for v in q.split(",")) for q in sys.argv[2].split(";")] if len(sys.argv) > 2 else [(8, 2, 0), (8, 2, 1), (7, 2, 1), (7, 1, 1), (8, 1, 1), (8, 2, 0)]
ref = None
for ps, py, *rest in variants:
    tagged = rest[0] if rest else 1
    ctx.set_option("mg_jacp_tagged", tagged)
    ctx.set_option("mg_patch_sweeps", ps)
    ctx.set_option("mg_group_sweeps", 64)
    ctx.set_option("mg_jacp_py", py)
    ctx.set_option("mg_jacobi_persist", 1)
    best = None
    for i in range(3):
        x.zero_()
        F.synchronize()
        ctx.call("fpr_kernel_timer", 1)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, cycles, False, opt=opt, return_history=True)
        F.synchronize()
        dt = (time.perf_counter() - t0) / len(hist)
        tot, cnt = C.c_double(0.0), C.c_long(0)
        ctx.call("fpr_kernel_timer_read", 6, C.byref(tot), C.byref(cnt))
        ctx.call("fpr_kernel_timer", 0)
        if best is None or dt < best[0]:
            best = (dt, tot.value, cnt.value, cit, list(hist))
    if ref is None:
        ref = best[4]
    dev = max(abs(a - c) / abs(c) for a, c in zip(best[4], ref))
    print("tagged %d " % tagged, end="")
    print("sweeps/group %d patch rows %d: %.3f ms per V-cycle, %.2f us per launch (%d launches), %.3f us per sweep, %d sweeps, history vs default %.2e, timeouts %d"
          % (ps, py, best[0] * 1e3, best[1] * 1e3 / max(best[2], 1), best[2], best[1] * 1e3 / max(best[3], 1), best[3], dev,
             ctx.get_option("mg_jacobi_persist_timeouts")), flush=True)
ctx.set_option("mg_patch_sweeps", 0)
ctx.set_option("mg_jacp_py", 0)
