#!/usr/bin/env python3
"""k_seam_march_v2 at 4097^2 against its chunk height (option mg_seam_rows_per_chunk; 0 = the host's choice): event time per launch."""
import ctypes as C
import os
import sys
import time

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


def timer(kind):
    tot, cnt = C.c_double(0.0), C.c_long(0)
    ctx.call("fpr_kernel_timer_read", kind, C.byref(tot), C.byref(cnt))
    return tot.value / max(cnt.value, 1)


def run(label):
    opt = mg.MGOpt()
    opt.coarse_solve_size, opt.coarse_solver = 5, mg.jacobi
    best = [1e9, 1e9, 1e9, 1e9]
    for i in range(4):
        x.zero_()
        F.synchronize()
        ctx.call("fpr_kernel_timer", 1)
        t0 = time.perf_counter()
        r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
        F.synchronize()
        wall = (time.perf_counter() - t0) / len(hist)
        v = [timer(4), timer(2), timer(3)]
        ctx.call("fpr_kernel_timer", 0)
        if i:
            best = [min(a, c) for a, c in zip(best, v + [1e9])]
        x.zero_()
        F.synchronize()
        t0 = time.perf_counter()
        r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
        F.synchronize()
        best[3] = min(best[3], (time.perf_counter() - t0) / len(hist))
    print("%-34s seam %.1f us  pre %.1f us  post %.1f us  V-cycle %.1f us" % (label, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e6), flush=True)


for rpc in [0, 96, 108, 116, 124, 150, 164, 172, 180, 196, 0]:
    ctx.set_option("mg_seam_rows_per_chunk", rpc)
    run("seam rows per chunk %3d" % rpc)
ctx.set_option("mg_seam_rows_per_chunk", 0)
for rpc in [0, 32, 48, 64, 96, 128, 164, 0]:
    ctx.set_option("mg_rows_per_chunk", rpc)
    run("two-sweep rows per chunk %3d" % rpc)
ctx.set_option("mg_rows_per_chunk", 0)
