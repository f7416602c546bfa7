"""Does the per-launch event timer (fpr_kernel_timer) slow the launches it brackets?  Wall time per fused pair, timer off / on.
usage: exp_timer_cost.py [n]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fpr_amd
F = fpr_amd.load(0)
ctx = F.ctx()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C_, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
def pairs(k):
    global A, C_
    for _ in range(k):
        F.part1.diffusion_3D_step_τ2(Ht, A, O, C_, R, *coef, 0.2, sq); A, C_ = C_, A
def wall(k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pairs(k); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
pairs(40)
for rep in range(3):
    for k in (10, 100):
        off = wall(k)
        ctx.call("fpr_kernel_timer", 1)
        on = wall(k)
        tot, cnt = C.c_double(0.0), C.c_long(0)
        ctx.call("fpr_kernel_timer_read", 1, C.byref(tot), C.byref(cnt))
        ctx.call("fpr_kernel_timer", 0)
        print("%3d pairs: wall per pair timer off %.1f us, on %.1f us (event time per launch %.1f us over %d launches)" % (k, off, on, tot.value / max(cnt.value, 1) * 1e3, cnt.value))
