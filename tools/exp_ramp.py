"""Per-launch duration of the fused pair right after a synchronisation (what a 20-step timed window sees) against steady state.
usage: exp_ramp.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
def pair():
    global A, C
    F.part1.diffusion_3D_step_τ2(Ht, A, O, C, R, *coef, 0.2, sq); A, C = C, A
def burst(k, idle_s):
    torch.cuda.synchronize(); time.sleep(idle_s)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for a, b in ev:
        a.record(); pair(); b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in ev]
for _ in range(30): pair()
for idle in (0.0, 0.001, 0.01, 0.1, 1.0):
    d = burst(40, idle)
    print("idle %.3f s: first 12 launches %s us | launches 20-40 avg %.1f" % (idle, " ".join("%.0f" % x for x in d[:12]), sum(d[20:]) / 20))
d = burst(400, 0.0)
print("400 launches: avg of first 10 %.1f, 10-50 %.1f, 50-200 %.1f, 200-400 %.1f" % (sum(d[:10]) / 10, sum(d[10:50]) / 40, sum(d[50:200]) / 150, sum(d[200:]) / 200))
