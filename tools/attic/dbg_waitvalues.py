"""z pair at 512^3: stream waits as events against stream memory operations (option stream_wait_values)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, fpr_amd
F = fpr_amd.load(0)
n = 512
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
def run(gg, pairs):
    global A, C
    for _ in range(pairs):
        gg.step2(Ht, A, O, C, R, *coef, 0.2, sq, join=False); A, C = C, A
def t(gg, K=40):
    run(gg, 6); gg.join(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); run(gg, K); gg.join(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K)
    return best * 1e6
g0 = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
base = t(g0)
print("plain %.1f us" % base)
gz = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=(0, 0, 1), transport="rccl", use_dist=False)
for v in (0, 1, 0, 1):
    F.ctx().set_option("stream_wait_values", v)
    us = t(gz)
    print("z stream_wait_values=%d: %.1f us (+%.1f %%)  norm %.6e" % (v, us, 100 * (us / base - 1), sq.cpu()[1].item()), flush=True)
