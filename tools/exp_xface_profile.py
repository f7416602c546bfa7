import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, fpr_amd
F = fpr_amd.load(0)
n = 512; dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
gg = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=(1, 0, 0), transport="rccl", use_dist=False)
for _ in range(30):
    gg.step2(Ht, A, O, C, R, *coef, 0.2, sq); A, C = C, A
torch.cuda.synchronize()
