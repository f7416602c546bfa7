#!/usr/bin/env python3
"""Seam pass with one column per lane (k_seam_march_v2) against two (k_seam_march_v3, option mg_seam_cols = 2): 4097^2 (and 2049^2) solves --
fields bit for bit, histories, event time of the seam pass, wall time per V-cycle; apply_BCs on and off."""
import ctypes as C
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
ctx = F.ctx()
for n in (4097,):
    h = 1.0 / (n - 1)
    b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    for bcs in (False,):
        ref = None
        for cols, pf, wpc in ((1, 4, 0), (2, 6, 0), (2, 12, 0), (2, 12, 2), (2, 6, 1), (1, 6, 0), (1, 4, 0), (2, 12, 0)):
            ctx.set_option("mg_seam_cols", cols)
            ctx.set_option("mg_seam_pf", pf)
            ctx.set_option("mg_seam_wg_per_cu", wpc)
            best = None
            for i in range(4):
                x = F.fzeros(n, n)
                F.synchronize()
                ctx.call("fpr_kernel_timer", 1)
                t0 = time.perf_counter()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 30, bcs, opt=mg.MGOpt(), return_history=True)
                F.synchronize()
                dt = time.perf_counter() - t0
                tot, cnt = C.c_double(0.0), C.c_long(0)
                ctx.call("fpr_kernel_timer_read", 4, C.byref(tot), C.byref(cnt))
                ctx.call("fpr_kernel_timer", 0)
                if best is None or dt < best[0]:
                    best = (dt, tot.value / max(cnt.value, 1), cnt.value, list(hist), F.tonumpy(x))
            if ref is None:
                ref = best
            same = np.array_equal(best[4], ref[4])
            dev = max(abs(a - c) / abs(c) for a, c in zip(best[3], ref[3])) if len(best[3]) == len(ref[3]) else -1.0
            print("n %d BCs %d cols %d pf %d wg/cu %d: %.3f ms per solve, %d cycles, %.4f ms per cycle, seam %.2f us x %d, field equal %s, history dev %.1e"
                  % (n, bcs, cols, pf, wpc, best[0] * 1e3, len(best[3]), best[0] * 1e3 / len(best[3]), best[1] * 1e3, best[2], same, dev), flush=True)
ctx.set_option("mg_seam_cols", 0)
ctx.set_option("mg_seam_pf", 4)
ctx.set_option("mg_seam_wg_per_cu", 0)
