#!/usr/bin/env python3
"""Five-level V-cycle with the Jacobi coarse solver (4097^2, coarse 257^2, 5140 sweeps per cycle): k_jacobi_patch with 8 / 7 / 6 sweeps per
launch (own tiles 16 / 18 / 20 on 32 x 32 regions: 289 / 225 / 169 workgroups) -- event time per launch and wall time per V-cycle."""
import ctypes as C
import os
import sys
import time
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
GS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
PERSIST = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for ps, gsw, pers in [(p_, g_, q_) for q_ in PERSIST for g_ in GS for p_ in ((8, 7, 6, 0) if q_ == 0 else (8,))]:
    ctx.set_option("mg_jacobi_persist", pers)
    ctx.set_option("mg_patch_sweeps", ps)
    ctx.set_option("mg_group_sweeps", gsw if gsw > 0 else 64)
    best = None
    for i in range(2):
        x.zero_()
        F.synchronize()
        ctx.call("fpr_kernel_timer", 1)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 4, False, opt=opt, return_history=True)
        F.synchronize()
        dt = (time.perf_counter() - t0) / len(hist)
        tot, cnt = C.c_double(0.0), C.c_long(0)
        ctx.call("fpr_kernel_timer_read", 6, C.byref(tot), C.byref(cnt))
        ctx.call("fpr_kernel_timer", 0)
        if best is None or dt < best[0]:
            best = (dt, tot.value / max(cnt.value, 1), cnt.value, cit, hist[-1])
    print("mg_jacobi_persist %d mg_group_sweeps %d mg_patch_sweeps %d: %.3f ms per V-cycle, %.2f us per launch (%d timed launches), %d coarse sweeps, last rms %.17g"
          % (pers, gsw, ps, best[0] * 1e3, best[1] * 1e3, best[2], best[3], best[4]), flush=True)
ctx.set_option("mg_patch_sweeps", 0)
ctx.set_option("mg_group_sweeps", 64)
ctx.set_option("mg_jacobi_persist", 1)
