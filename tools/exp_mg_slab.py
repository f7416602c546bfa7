#!/usr/bin/env python3
"""Is the per-lease spread of the 4097^2 V-cycle (seam pass 99-116 us, solve 1.755-1.875 ms by lease, and on the slow leases every
candidate of the pool the same) the placement class of k_diff3_march2's arrays?  The four arrays the finest passes stream
(x, b, two ping-pong partners; 134 MB each) are taken
  separate   as four plain allocations (what the bench's pool starts from),
  oneblock   as four windows of ONE 1 GiB allocation (one class for sure, if the class belongs to the allocation),
  ranked     as the first 134 MB of the four 1 GiB blocks fpr_placement_rank picks out of twelve (four classes, if there are four),
  ranked+off the same four blocks, each window at another offset inside its block,
and one solve is timed on each (best of 3, as the bench's trial), with the event time of the seam pass."""
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
GiB = 1 << 30
b_host = F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")
b0 = F.asdevice(b_host)
KT_SEAM = 4


def window(block, off_bytes):
    """an (n, n) column-major float64 array inside a uint8 block"""
    w = block[off_bytes:off_bytes + 8 * n * n].view(torch.float64).view(n, n)
    return w.permute(1, 0)


def solve(arrs, label):
    tx, tb, t1, t2 = arrs
    mg.provide_arena_(n, n, t1, t2)
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
    print("%-22s solve %.3f ms, seam pass %.1f us" % (label, best * 1e3, 1e3 * tot.value / max(cnt.value, 1)), flush=True)
    return best * 1e3


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rnd in range(rounds):
    sep = [F.fzeros(n, n) for _ in range(4)]
    solve(sep, "separate")
    blocks, spacers = [], []
    for i in range(12):
        if i:
            spacers.append(torch.empty(4 * GiB, dtype=torch.uint8, device="cuda"))
        blocks.append(torch.zeros(GiB, dtype=torch.uint8, device="cuda"))
    for bi in (0, 5, 11):
        solve([window(blocks[bi], k * (GiB // 4)) for k in range(4)], "oneblock[%d]" % bi)
    ptrs = (C.c_void_p * 12)(*[b.data_ptr() for b in blocks])
    pairs = [(0, 1), (2, 1), (3, 1), (2, 3), (0, 2)]
    flat = [i for p in pairs for i in p]
    rep = (C.c_double * 16)()
    chosen = (C.c_int * 4)()
    torch.cuda.synchronize()
    ctx.call("fpr_placement_rank", ptrs, 12, GiB // 8, 4, (C.c_int * len(flat))(*flat), len(pairs), None, None, chosen, rep)
    ch = [int(c) for c in chosen]
    print("twelve 1 GiB blocks: pair copies %.0f / %.0f / %.0f GB/s (slowest / median / fastest); chosen %s with slowest pair %.0f" % (
        rep[2], rep[1], rep[0], ch, rep[3]), flush=True)
    solve([window(blocks[c], 0) for c in ch], "ranked")
    solve([window(blocks[c], k * (GiB // 4)) for k, c in enumerate(ch)], "ranked+off")
    solve([window(blocks[c], 0) for c in (0, 1, 2, 3)], "first four blocks")
    solve(sep, "separate again")
    mg.provide_arena_(n, n, None, None)
    del blocks, spacers, sep
    torch.cuda.empty_cache()
