#!/usr/bin/env python3
"""Option mg_wave_target (waves a two-sweep pass should have at least; decides the rows per chunk of every marching level): 4097^2 and 2049^2
solves per value, best of 5, microseconds per V-cycle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
ctx = F.ctx()
for n in (4097, 2049):
    h = 1.0 / (n - 1)
    b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    x = F.fzeros(n, n)
    for tgt in (4096, 1024, 2048, 3072, 6144, 8192, 4096):
        ctx.set_option("mg_wave_target", tgt)
        best = 1e9
        for i in range(6):
            x.zero_()
            F.synchronize()
            t0 = time.perf_counter()
            r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, return_history=True)
            F.synchronize()
            if i:
                best = min(best, (time.perf_counter() - t0) / len(hist))
        print("n=%d mg_wave_target %5d: %.1f us per V-cycle (%d cycles)" % (n, tgt, best * 1e6, len(hist)), flush=True)
ctx.set_option("mg_wave_target", 4096)
