#!/usr/bin/env python3
"""One lease in six or seven offers only candidates of ONE placement class (every pair of the pool copies below 4950 GB/s, the fused launch
takes 0.83 ms whatever the assignment -- the driver's boxes of rounds 2-4; the same card is fine on another lease: profiles/r5_driver_repro.txt,
repro5 / repro14).  This probe asks, on whatever lease it gets, how the classes of a pool depend on HOW the candidates were allocated:
  plain      12 x 1 GiB one after the other
  spaced4    4 GiB untouched spacers between them (what placement.alloc_fields does)
  touched4   the same, spacers written once
  spaced16   16 GiB untouched spacers (the pool spans 200 GiB)
  slab       ONE 72 GiB allocation, candidates carved out at a pitch of 6 GiB
  churn      200 GiB allocated, written and freed first; then as spaced4
For every strategy: the pool's pair copy rates (fpr_placement_rank without a trial) and the fused diffusion launch on the best-ranked five.
usage: slow_state_probe.py [strategies,comma,separated]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd

F = fpr_amd.load(0)
ctx = F.ctx()
n = 512
N = n ** 3
GiB = 1 << 30
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
pairs = [(0, 1), (0, 2), (2, 3), (1, 3), (1, 2)]
flat = [i for p in pairs for i in p]


def as_field(t):
    return t.view(torch.float64)[:N].view(n, n, n).permute(2, 1, 0) if False else torch.as_strided(t.view(torch.float64), (n, n, n), (1, n, n * n))


def fused_ms(arrs, reps=40):
    tHt, tA, tC, tR, tB = arrs
    if not F.part1.can_step_τ2(tHt, tA, tB, tC, tR):
        return float("nan")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(reps + 20):
        if i == 20:
            e0.record()
        if i & 1:
            F.part1.diffusion_3D_step_τ2(tHt, tC, tB, tA, tR, *coef)
        else:
            F.part1.diffusion_3D_step_τ2(tHt, tA, tB, tC, tR, *coef)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def evaluate(name, cands, keep):
    ptrs = (C.c_void_p * len(cands))(*[a.data_ptr() for a in cands])
    rep = (C.c_double * 16)()
    chosen = (C.c_int * 5)()
    torch.cuda.synchronize()
    t0 = time.time()
    if SEARCH:     # the library's whole search (pair ranking, then the kernel itself as the judge), as bench.py runs it
        @_TRIAL_FN
        def cb(_user, idx, k):
            return float(fused_ms([cands[idx[i]] for i in range(k)], 6))

        ctx.call("fpr_placement_rank", ptrs, len(cands), N, 5, (C.c_int * len(flat))(*flat), len(pairs), C.cast(cb, C.c_void_p), None, chosen, rep)
    else:
        ctx.call("fpr_placement_rank", ptrs, len(cands), N, 5, (C.c_int * len(flat))(*flat), len(pairs), None, None, chosen, rep)
    best = [cands[chosen[i]] for i in range(5)]
    ms_best = fused_ms(best)
    global LAST_BEST
    LAST_BEST = ms_best
    ms_first = fused_ms(cands[:5])
    va = [a.data_ptr() for a in cands]
    global LAST_FASTEST
    LAST_FASTEST = rep[0]
    print("%-9s %2d candidates spanning %5.1f GiB of addresses: pair copies %.0f / %.0f / %.0f GB/s (slowest / median / fastest); fused launch %.4f ms on "
          "the best-ranked five %s, %.4f on the first five; %.1f s" % (name, len(cands), (max(va) - min(va)) / GiB + 1, rep[2], rep[1], rep[0], ms_best,
                                                                      [int(chosen[i]) for i in range(5)], ms_first, time.time() - t0), flush=True)
    del keep


LAST_FASTEST = 0.0
LAST_BEST = 0.0
_TRIAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_int), C.c_int)
SEARCH = "search" in sys.argv[2:]          # evaluate every strategy with the full search (trials), not the pair ranking alone
SLOW_MS = 0.775                            # only-if-slow with search: go on when the best assignment of the first strategy is slower than this


def alloc(k, spacer_gib=0, touch=False):
    cands, keep = [], []
    for i in range(k):
        cands.append(F.fzeros(n, n, n))
        if spacer_gib and i + 1 < k:
            s = torch.empty(spacer_gib * GiB, dtype=torch.uint8, device="cuda")
            if touch:
                s.zero_()
            keep.append(s)
    return cands, keep


which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["spaced4", "plain", "touched4", "spaced16", "slab", "slab1", "churn", "spaced4"]
ONLY_IF_SLOW = "only-if-slow" in sys.argv[2:]      # leave after the first strategy unless its fastest pair copies below 5000 GB/s
free, total = torch.cuda.mem_get_info()
print("card %s, %.0f of %.0f GiB free" % (F.ctx().get_option("none") if False else "", free / GiB, total / GiB), flush=True)
for w in which:
    torch.cuda.empty_cache()
    if w == "plain":
        c, k = alloc(12)
    elif w == "spaced4":
        c, k = alloc(12, 4)
    elif w == "touched4":
        c, k = alloc(12, 4, True)
    elif w == "spaced16":
        c, k = alloc(12, 16)
    elif w == "slab":
        slab = torch.empty(72 * GiB, dtype=torch.uint8, device="cuda")
        c = []
        for i in range(12):
            t = slab[i * 6 * GiB:i * 6 * GiB + N * 8]
            t.zero_()
            c.append(torch.as_strided(t.view(torch.float64), (n, n, n), (1, n, n * n)))
        k = [slab]
    elif w == "slab1":          # one slab, candidates side by side (pitch 1 GiB)
        slab = torch.empty(13 * GiB, dtype=torch.uint8, device="cuda")
        c = []
        for i in range(12):
            t = slab[i * GiB:i * GiB + N * 8]
            t.zero_()
            c.append(torch.as_strided(t.view(torch.float64), (n, n, n), (1, n, n * n)))
        k = [slab]
    elif w == "churn":
        big = [torch.empty(50 * GiB, dtype=torch.uint8, device="cuda") for _ in range(4)]
        for b_ in big:
            b_.zero_()
        torch.cuda.synchronize()
        del big
        torch.cuda.empty_cache()
        c, k = alloc(12, 4)
    else:
        continue
    evaluate(w, c, k)
    del c, k
    if ONLY_IF_SLOW and w == which[0] and (LAST_BEST < SLOW_MS if SEARCH else LAST_FASTEST >= 5000.0):
        print("not a slow lease (fastest pair %.0f GB/s, best assignment %.4f ms): nothing to look at" % (LAST_FASTEST, LAST_BEST))
        break
