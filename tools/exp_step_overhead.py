"""Host-side cost of the multi-rank step choreography (boundary slabs, pack, post, interior, join), measured
with two emulated z-slab ranks on one GPU (in-process fake of the P2P layer, tests/test_gpu_halo.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fpr_amd
from test_gpu_halo import FakeDist
F = fpr_amd.load(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dims = (1, 1, 2)
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
mail = {}
ranks = []
for r in range(2):
    gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
    gg.dims, gg.nprocs, gg.me = dims, 2, r
    gg.coords = gg.coords_of(r)
    gg.neighbors = {5: (0, 0, 1)} if r == 0 else {4: (0, 0, 0)}
    gg.dist = FakeDist(mail, r)
    Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 10.), dx, dx, dx, Ht, gg.coords)
    ranks.append(dict(gg=gg, Ht=Ht, A=Ht.clone(), B=F.fzeros(n, n, n), R=F.fzeros(n, n, n), sq=F.fzeros(1)))
def step_all():
    sts = [s["gg"].step_begin(s["Ht"], s["A"], s["B"], s["R"], *coef, 0.2, s["sq"]) for s in ranks]
    for s, st in zip(ranks, sts):
        s["gg"].step_end(st); s["A"], s["B"] = s["B"], s["A"]
for _ in range(20): step_all()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 100
for _ in range(K): step_all()
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
# single-rank reference at the same size
g1 = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
s = ranks[0]
for _ in range(20): g1.step(s["Ht"], s["A"], s["B"], s["R"], *coef, 0.2, s["sq"])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2 * K): g1.step(s["Ht"], s["A"], s["B"], s["R"], *coef, 0.2, s["sq"])
torch.cuda.synchronize(); t_single = time.perf_counter() - t0
print("n=%d: emulated 2-rank step: host enqueue %.1f us per rank-step, total %.1f us per rank-step; plain single-rank step %.1f us"
      % (n, t_host / (2 * K) * 1e6, t_all / (2 * K) * 1e6, t_single / (2 * K) * 1e6))
