"""Comm-stream share (diff3_comm_units) per face set, ONE process (every case on the same arrays: the placement draw of a process
moves the plain pair by +-6 %, so only numbers of one process compare).  usage: exp_faces_units.py [n] [cases] [units]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
want = sys.argv[2].split(",") if len(sys.argv) > 2 else None
units = [int(u) for u in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 24, 32]
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
K = 40
def run(gg, pairs):
    global A, C
    for _ in range(pairs):
        gg.step2(Ht, A, O, C, R, *coef, 0.2, sq, join=False); A, C = C, A
def timed(gg):
    run(gg, 24); torch.cuda.synchronize(); t0 = time.perf_counter(); run(gg, K); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
CASES = (("z", (0, 0, 1), 0), ("yz", (0, 1, 1), 0), ("x", (1, 0, 0), 0), ("xy", (1, 1, 0), 0), ("xyz", (1, 1, 1), 0),
         ("corner", (1, 1, 1), 0b010101), ("x1", (1, 0, 0), 0b000001), ("xz1", (1, 0, 1), 0b010001))
gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
base = timed(gg)
print("plain pair %.1f us" % base)
for name, periods, drop in CASES:
    if want and name not in want: continue
    gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=periods, transport="rccl", use_dist=False, drop_faces=drop)
    out = []
    for k in units:
        F.ctx().set_option("diff3_comm_units", k)
        t = timed(gg)
        out.append("k=%d %.1f us (+%.1f %%)" % (k, t, 100 * (t / base - 1)))
    F.ctx().set_option("diff3_comm_units", 0)
    F.ctx().set_option("diff3_xstrips", 0)   # the x-shell in the field (rounds 2-3)
    t = timed(gg)
    out.append("field form %.1f us (+%.1f %%)" % (t, 100 * (t / base - 1)))
    F.ctx().set_option("diff3_xstrips", 1)
    t = timed(gg)
    print("%-7s %s | default %.1f us (+%.1f %%)" % (name, "  ".join(out), t, 100 * (t / base - 1)))
    gg.join(); F.grid.finalize_global_grid()
gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
print("plain pair again %.1f us" % timed(gg))
