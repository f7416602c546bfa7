"""Cost of the fused-pair choreography with neighbours on x / y / z faces, on ONE GPU: a single rank that is its own
periodic neighbour through the library's RCCL transport (the planes really travel through ncclSend / ncclRecv), against
the plain single-rank fused launch at the same size.  Periodic in k dimensions = 2k faces with a neighbour; an interior
rank of a (2,2,2) decomposition has 3.  usage: exp_step2_faces.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
K = 40
def run(gg, pairs):
    global A, C
    for _ in range(pairs):
        gg.step2(Ht, A, O, C, R, *coef, 0.2, sq, join=False); A, C = C, A
def run1(gg, its):
    global A, O
    for _ in range(its):
        gg.step(Ht, A, O, R, *coef, 0.2, sq[:1]); A, O = O, A
res = {}
# name, periodic dimensions, dropped faces (bit 2*dim+side): "corner" = one face per dimension, the face set of a rank of (2,2,2)
CASES = (("none", (0, 0, 0), 0), ("z", (0, 0, 1), 0), ("yz", (0, 1, 1), 0), ("xyz", (1, 1, 1), 0), ("x", (1, 0, 0), 0), ("y", (0, 1, 0), 0),
         ("xy", (1, 1, 0), 0), ("corner", (1, 1, 1), 0b010101), ("z1", (0, 0, 1), 0b010000))
if len(sys.argv) > 2: CASES = tuple(c for c in CASES if c[0] in sys.argv[2].split(",") or c[0] == "none")
for name, periods, drop in CASES:
    if any(periods):
        gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=periods, transport="rccl", use_dist=False, drop_faces=drop)
    else:
        gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
    if os.environ.get("FPR_RESERVE") and any(periods): gg._reserve = int(os.environ["FPR_RESERVE"])
    run(gg, 24); torch.cuda.synchronize(); t0 = time.perf_counter(); run(gg, K); th = time.perf_counter() - t0   # (24 pairs: past the throttled launches behind the idle period of the set-up, tools/exp_ramp.py)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    run1(gg, 10); torch.cuda.synchronize(); t0 = time.perf_counter(); run1(gg, 2 * K); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    res[name] = t
    print("faces with a neighbour: %-4s  fused pair %.1f us (host enqueue %.1f us) = %.1f us per iteration, +%.1f %% over no neighbours;"
          "  single steps %.1f us per iteration" % (name, t / K * 1e6, th / K * 1e6, t / K / 2 * 1e6, 100 * (t / res["none"] - 1), t1 / (2 * K) * 1e6))
    if any(periods): F.grid.finalize_global_grid()
