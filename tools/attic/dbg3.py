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
def t(gg, K=40, join=True, join_warm=None, nosync=False):
    run(gg, 6)
    if join if join_warm is None else join_warm: gg.join()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); run(gg, K)
        if join and mode == "h":
            import ctypes as C
            c = F.ctx()
            for sel in (2, 1):
                h = C.c_void_p(); c.call("fpr_stream_handle", sel, C.byref(h))
                torch.cuda.ExternalStream(h.value).synchronize()
            gg.join()
        elif join: gg.join()
        if nosync: F.ctx().synchronize()
        else: torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    return best * 1e6
mode = sys.argv[1]
if mode == "a":
    g0 = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), use_dist=False)
else:
    g0 = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=(0, 0, 0), transport="rccl", use_dist=False)
base = t(g0)
print(mode, "plain %.1f us" % base)
gz = F.grid.GlobalGrid(n, n, n, dims=(1, 1, 1), periods=(1, 1, 1), transport="rccl", use_dist=False, drop_faces=21)
print(mode, "neighbors", sorted(gz.neighbors))
us = t(gz, join=(mode not in "ce"), join_warm=(mode not in "cf"), nosync=(mode == "g")); print(mode, "corner: %.1f us (+%.1f %%)" % (us, 100 * (us / base - 1)), flush=True)
if mode == "d":
    F.ctx().reserve_comm_cus(0)
    us = t(gz); print(mode, "corner after reserve(0): %.1f us (+%.1f %%)" % (us, 100 * (us / base - 1)), flush=True)
