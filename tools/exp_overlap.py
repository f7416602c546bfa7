"""Fused-pair choreography with neighbours at n^3 on ONE GPU (one rank, periodic = its own neighbour through the
library's RCCL transport), a few pairs, for a dispatch timeline under rocprofv3 --kernel-trace: does the RCCL kernel
of an exchange finish inside the core launch beside it, or drain behind it?
usage: exp_overlap.py <n> <periods e.g. 001> <pairs> [k=v,k=v library options]     (FPR_DROP_FACES=21: corner rank of (2,2,2))"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]); periods = tuple(int(c) for c in sys.argv[2]); K = int(sys.argv[3])
if len(sys.argv) > 4 and sys.argv[4] not in ("", "none"):
    for kv in sys.argv[4].split(","):
        k, v = kv.split("="); F.ctx().set_option(k, int(v))
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
if any(periods):
    gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=periods, transport="rccl", use_dist=False,
                           drop_faces=int(os.environ.get("FPR_DROP_FACES", "0")))
    if os.environ.get("FPR_RESERVE"): gg._reserve = int(os.environ["FPR_RESERVE"])
else:
    gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
def run(pairs):
    global A, C
    for _ in range(pairs):
        gg.step2(Ht, A, O, C, R, *coef, 0.2, sq, join=False); A, C = C, A
run(6); torch.cuda.synchronize()
t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("n=%d periods=%s opts=%s: fused pair %.1f us" % (n, sys.argv[2], sys.argv[4] if len(sys.argv) > 4 else "", t / K * 1e6))
if any(periods): F.grid.finalize_global_grid()
