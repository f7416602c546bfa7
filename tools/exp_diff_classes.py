#!/usr/bin/env python3
"""Which of the fused diffusion launch's four streams (Ht read, field read, field written, dHdtau read + written; the fifth array only lends
its boundary cells) have to differ in placement class, and how far the library's search (fpr_placement_rank + trials) is from the best
assignment there is.  NB candidates of 512^3 behind 4 GiB spacers, the pair-copy rate of every pair, then EVERY assignment of four of
them to the four streams timed with the kernel itself (events, 8 launches behind 4), then the search on the same candidates."""
import ctypes as C
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fpr_amd

F = fpr_amd.load(0)
ctx = F.ctx()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 512
GiB = 1 << 30
dx = 10.0 / (n - 1)
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
blocks, spacers = [], []
for i in range(NB + 1):
    if i:
        spacers.append(torch.empty(4 * GiB, dtype=torch.uint8, device="cuda"))
    blocks.append(F.fzeros(n, n, n))
Bfix = blocks.pop()          # Htau2's role: boundary cells only
rate = np.zeros((NB, NB))
rep = (C.c_double * 16)()
chosen = (C.c_int * 2)()
pr = (C.c_int * 2)(0, 1)
torch.cuda.synchronize()
for i in range(NB):
    for j in range(i + 1, NB):
        ptrs = (C.c_void_p * 2)(blocks[i].data_ptr(), blocks[j].data_ptr())
        ctx.call("fpr_placement_rank", ptrs, 2, GiB // 8, 2, pr, 1, None, None, chosen, rep)
        rate[i, j] = rate[j, i] = rep[0]
print("pair copy GB/s:")
for i in range(NB):
    print("  " + " ".join("%5.0f" % rate[i, j] if i != j else "    -" for j in range(NB)), flush=True)


def ms_of(tHt, tA, tC, tR, warm=2, timed=4, tB=None):
    tB = Bfix if tB is None else tB
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(warm + timed):
        if i == warm:
            e0.record()
        F.part1.diffusion_3D_step_τ2(tHt, tA, tB, tC, tR, *coef)
        F.part1.diffusion_3D_step_τ2(tHt, tC, tB, tA, tR, *coef)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (2 * timed)


for _ in range(50):
    ms_of(blocks[0], blocks[1], blocks[2], blocks[3])          # the card at its working clocks
names = ("Ht", "A", "C", "R")
res = []
t0 = time.time()
for perm in itertools.permutations(range(NB), 4):
    # A and C swap roles every launch: (Ht, A, C, R) and (Ht, C, A, R) are the same assignment
    if perm[1] > perm[2]:
        continue
    res.append((ms_of(*[blocks[k] for k in perm]), perm))
    if len(res) % 200 == 0:
        print("  %d assignments, %.0f s" % (len(res), time.time() - t0), flush=True)
res.sort()
os.makedirs("gpurun_out/r5", exist_ok=True)
np.savez("gpurun_out/r5/exp_diff_classes.npz", rate=rate, ms=np.array([t for t, _ in res]), perm=np.array([p for _, p in res]))
print("%d assignments: best %.4f ms %s, median %.4f, worst %.4f %s" % (len(res), res[0][0], res[0][1], res[len(res) // 2][0], res[-1][0], res[-1][1]))
pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
off = rate[~np.eye(NB, dtype=bool)]
THR = 0.5 * (off.min() + off.max())          # between the two modes of this lease's pair rates
print("mean kernel time by the pair-copy rate of each pair of streams (fast pair: > %.0f GB/s, the middle of this pool's range):" % THR)
for a, b in pairs:
    fast = [t for t, p in res if rate[p[a], p[b]] > THR]
    slow = [t for t, p in res if rate[p[a], p[b]] <= THR]
    print("  %-3s-%-3s fast pair: %.4f ms (%4d)   slow pair: %.4f ms (%4d)" % (names[a], names[b], np.mean(fast) if fast else 0, len(fast), np.mean(slow) if slow else 0, len(slow)))
# least squares: time = t0 + sum over pairs of w_pair * [pair slow]
X = np.array([[1.0] + [1.0 if rate[p[a], p[b]] <= THR else 0.0 for a, b in pairs] for _, p in res])
y = np.array([t for t, _ in res])
w, *_ = np.linalg.lstsq(X, y, rcond=None)
print("least squares, ms added by a slow pair: base %.4f; " % w[0] + ", ".join("%s-%s %+.4f" % (names[a], names[b], w[k + 1]) for k, (a, b) in enumerate(pairs)))
print("residual rms %.4f ms" % float(np.sqrt(np.mean((X @ w - y) ** 2))))
for t, p in res[:8]:
    print("  best  %.4f %s  pair rates %s" % (t, p, " ".join("%s-%s %.0f" % (names[a], names[b], rate[p[a], p[b]]) for a, b in pairs)))
for t, p in res[-4:]:
    print("  worst %.4f %s  pair rates %s" % (t, p, " ".join("%s-%s %.0f" % (names[a], names[b], rate[p[a], p[b]]) for a, b in pairs)))
# the search on the same candidates (+ the fifth array as a candidate of its own)
cands = blocks + [Bfix]
report = {}


def trial(arrs):
    tHt, tA, tC, tR, tB = arrs
    return ms_of(tHt, tA, tC, tR, 2, 3, tB)


_TRIAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_int), C.c_int)


@_TRIAL_FN
def cb(_user, idx, k):
    return float(trial([cands[idx[i]] for i in range(k)]))


flat = [i for p in [(0, 1), (0, 2), (2, 3), (1, 3), (1, 2)] for i in p]
ch5 = (C.c_int * 5)()
ptrs = (C.c_void_p * len(cands))(*[a.data_ptr() for a in cands])
ctx.call("fpr_placement_rank", ptrs, len(cands), GiB // 8, 5, (C.c_int * len(flat))(*flat), len(flat) // 2, C.cast(cb, C.c_void_p), None, ch5, rep)
sel = [int(c) for c in ch5]
print("the library's search on the same candidates: %s, trial best %.4f ms (%d trials); re-timed as above: %.4f ms" % (
    sel, rep[6], int(rep[5]), ms_of(cands[sel[0]], cands[sel[1]], cands[sel[2]], cands[sel[3]], tB=cands[sel[4]])))
print("the best assignment re-timed: %.4f ms" % ms_of(*[blocks[k] for k in res[0][1]]))
