"""Cost of the multi-rank choreography of fused iteration pairs (GlobalGrid.step2_begin/middle/end), measured with
emulated z-slab ranks on one GPU (in-process fake of the P2P layer, tests/test_gpu_halo.py): time per rank-pair
against the plain single-rank fused launch at the same size.  usage: exp_step2_overhead.py [n] [ranks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fpr_amd
from test_gpu_halo import FakeDist
F = fpr_amd.load(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dims = (1, 1, W)
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
mail = {}
ranks = []
for r in range(W):
    gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
    gg.dims, gg.nprocs, gg.me = dims, W, r
    gg.coords = gg.coords_of(r)
    gg.neighbors = {}
    if r > 0: gg.neighbors[4] = (0, 0, r - 1)
    if r < W - 1: gg.neighbors[5] = (0, 0, r + 1)
    gg.dist = FakeDist(mail, r)
    Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5. * W), dx, dx, dx, Ht, gg.coords)
    ranks.append(dict(gg=gg, Ht=Ht, A=Ht.clone(), O=F.fzeros(n, n, n), C=Ht.clone(), R=F.fzeros(n, n, n), sq=F.fzeros(2)))
def pair_all():
    sts = [s["gg"].step2_begin(s["Ht"], s["A"], s["O"], s["C"], s["R"], *coef, 0.2, s["sq"]) for s in ranks]
    for s, st in zip(ranks, sts): s["gg"].step2_middle(st)
    for s, st in zip(ranks, sts):
        s["gg"].step2_end(st); s["A"], s["C"] = s["C"], s["A"]
for _ in range(10): pair_all()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 50
for _ in range(K): pair_all()
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
g1 = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
s = ranks[0]
for _ in range(10): g1.step2(s["Ht"], s["A"], s["O"], s["C"], s["R"], *coef, 0.2, s["sq"])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(W * K): g1.step2(s["Ht"], s["A"], s["O"], s["C"], s["R"], *coef, 0.2, s["sq"])
torch.cuda.synchronize(); t_single = time.perf_counter() - t0
print("n=%d, %d emulated ranks: fused pair per rank: host enqueue %.1f us, total %.1f us (includes the fake transport's plane copies); "
      "plain single-rank fused pair %.1f us  => choreography overhead %.1f %%"
      % (n, W, t_host / (W * K) * 1e6, t_all / (W * K) * 1e6, t_single / (W * K) * 1e6, 100.0 * (t_all / t_single - 1.0)))
