#!/usr/bin/env python3
"""Does the finest-level seam pass of the V-cycle depend on where its arrays lie (as k_diff3_march2 does, placement.py)?
Fresh x / b allocations per round (earlier ones kept alive so that new physical pages are used); per round the event time of
k_seam_march_v2 and the wall time per V-cycle of a 4097^2 solve."""
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
b_host = F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")
keep = []
KT_SEAM, KT_PRE, KT_POST = 4, 2, 3


def timer(kind):
    tot, cnt = C.c_double(0.0), C.c_long(0)
    ctx.call("fpr_kernel_timer_read", kind, C.byref(tot), C.byref(cnt))
    return tot.value / max(cnt.value, 1)


NEW_CTX = len(sys.argv) > 2 and sys.argv[2] == "newctx"    # a fresh context (= a fresh level arena inside the library) every round
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    if NEW_CTX and rnd:
        keep.append(torch.empty((37 + 61 * rnd) << 20, dtype=torch.uint8, device="cuda"))     # a spacer of another size
        F.reset()
        ctx = F.ctx()
    b = F.asdevice(b_host)
    x = F.fzeros(n, n)
    keep += [b, x]
    opt = mg.MGOpt()
    opt.coarse_solve_size, opt.coarse_solver = 5, mg.jacobi
    ts = []
    for i in range(4):
        x.zero_()
        F.synchronize()
        if i == 3:
            ctx.call("fpr_kernel_timer", 1)
        t0 = time.perf_counter()
        r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
        F.synchronize()
        ts.append((time.perf_counter() - t0) / len(hist))
    seam, pre, post = timer(KT_SEAM), timer(KT_PRE), timer(KT_POST)
    ctx.call("fpr_kernel_timer", 0)
    print("round %2d: x at %#x b at %#x: seam %.1f us  pre %.1f us  post %.1f us  V-cycle %.1f us" % (rnd, x.data_ptr(), b.data_ptr(), seam * 1e3, pre * 1e3, post * 1e3, min(ts[1:]) * 1e6), flush=True)
