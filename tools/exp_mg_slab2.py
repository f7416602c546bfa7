#!/usr/bin/env python3
"""Which of the seven arrays the finest passes of a 4097^2 V-cycle stream (x, b, the two ping-pong partners; the first coarse level's
injected residual and two correction buffers) have to differ in placement class for the seam pass to run in its fast mode (103 us against
113; tools/exp_mg_slab.py: four windows of ONE 1 GiB block can be fast or slow).  Twelve 1 GiB blocks, the pair-copy rate of every pair
(fpr_placement_rank on two candidates), then solves with the arrays as windows of chosen blocks."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
nc = 1 + (n - 1) // 2
h = 1.0 / (n - 1)
GiB = 1 << 30
b_host = F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")
b0 = F.asdevice(b_host)
KT_SEAM = 4
NB = 12


def window(block, off_bytes, m=n):
    w = block[off_bytes:off_bytes + 8 * m * m].view(torch.float64).view(m, m)
    return w.permute(1, 0)


def solve(big, coarse, label):
    tx, tb, t1, t2 = big
    mg.provide_arena_(n, n, t1, t2)
    if coarse is None:
        mg.provide_arena_coarse_(n, n, None, None, None)
    else:
        for a in coarse:
            a.zero_()
        mg.provide_arena_coarse_(n, n, *coarse)
    tb.copy_(b0)
    best = None
    for i in range(4):
        tx.zero_()
        F.synchronize()
        if i == 3:
            ctx.call("fpr_kernel_timer", 1)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mg.MGsolve_2DPoisson_(tx, tb, h, 0.0, 1e-6, 100, False, opt=mg.MGOpt(), return_history=False)
        F.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    tot, cnt = C.c_double(0.0), C.c_long(0)
    ctx.call("fpr_kernel_timer_read", KT_SEAM, C.byref(tot), C.byref(cnt))
    ctx.call("fpr_kernel_timer", 0)
    print("%-58s solve %.3f ms, seam pass %.1f us" % (label, best * 1e3, 1e3 * tot.value / max(cnt.value, 1)), flush=True)


blocks, spacers = [], []
for i in range(NB):
    if i:
        spacers.append(torch.empty(4 * GiB, dtype=torch.uint8, device="cuda"))
    blocks.append(torch.zeros(GiB, dtype=torch.uint8, device="cuda"))
rate = [[0.0] * NB for _ in range(NB)]
rep = (C.c_double * 16)()
chosen = (C.c_int * 2)()
pr = (C.c_int * 2)(0, 1)
torch.cuda.synchronize()
for i in range(NB):
    for j in range(i + 1, NB):
        ptrs = (C.c_void_p * 2)(blocks[i].data_ptr(), blocks[j].data_ptr())
        ctx.call("fpr_placement_rank", ptrs, 2, GiB // 8, 2, pr, 1, None, None, chosen, rep)
        rate[i][j] = rate[j][i] = rep[0]
print("pair copy GB/s between the twelve blocks:")
for i in range(NB):
    print("  " + " ".join("%5.0f" % rate[i][j] if i != j else "    -" for j in range(NB)), flush=True)
# block P = 0; R = the block copying slowest with P (same class), Q = fastest with P (another class), S = fast with both P and Q
P = 0
R = min((j for j in range(NB) if j != P), key=lambda j: rate[P][j])
Q = max((j for j in range(NB) if j != P), key=lambda j: rate[P][j])
S = max((j for j in range(NB) if j not in (P, Q)), key=lambda j: min(rate[P][j], rate[Q][j]))
print("P = %d, R = %d (%.0f with P), Q = %d (%.0f with P), S = %d (%.0f with P, %.0f with Q)" % (P, R, rate[P][R], Q, rate[P][Q], S, rate[P][S], rate[Q][S]))
Q4 = GiB // 4
CO = 3 * Q4 + (160 << 20)      # the coarse windows lie behind the big ones of a block: 3 x 33.6 MB from 928 MiB on -- no: keep them inside
CO = 0


def big_in(bl):        # four windows of one block (134 MB each at 0, 160, 320, 480 MiB)
    return [window(blocks[bl], k * (160 << 20)) for k in range(4)]


def coarse_in(bl):     # three coarse windows of one block (33.6 MB each from 700 MiB on)
    return [window(blocks[bl], (700 << 20) + k * (40 << 20), nc) for k in range(3)]


for rnd in range(2):
    solve(big_in(P), None, "big in P, coarse the library's own")
    solve(big_in(P), coarse_in(P), "big in P, coarse in P")
    solve(big_in(P), coarse_in(R), "big in P, coarse in R (P's class)")
    solve(big_in(P), coarse_in(Q), "big in P, coarse in Q (another class)")
    solve(big_in(Q), coarse_in(P), "big in Q, coarse in P")
    solve(big_in(Q), coarse_in(Q), "big in Q, coarse in Q")
    xb_P_t_Q = [window(blocks[P], 0), window(blocks[P], 160 << 20), window(blocks[Q], 0), window(blocks[Q], 160 << 20)]
    for cb, name in ((P, "P"), (Q, "Q"), (S, "S")):
        solve(xb_P_t_Q, coarse_in(cb), "x, b in P, partners in Q, coarse in %s" % name)
    x_P_b_Q = [window(blocks[P], 0), window(blocks[Q], 0), window(blocks[P], 160 << 20), window(blocks[Q], 160 << 20)]
    for cb, name in ((P, "P"), (S, "S")):
        solve(x_P_b_Q, coarse_in(cb), "x, partner1 in P, b, partner2 in Q, coarse in %s" % name)
    four = [window(blocks[c], 0) for c in (P, Q, S, R)]
    for cb, name in ((P, "P"), (S, "S"), (R, "R")):
        solve(four, [window(blocks[cb], (700 << 20) + k * (40 << 20), nc) for k in range(3)], "x P, b Q, partners S, R; coarse in %s" % name)
mg.provide_arena_(n, n, None, None)
mg.provide_arena_coarse_(n, n, None, None, None)
