import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0); mg = F.multigrid
n = 4097
b = F.asdevice(F.part2.splitmix64_uniform(n*n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
def run(label):
    ts = []
    for rep in range(6):
        x.zero_(); F.synchronize(); t0 = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = mg.MGsolve_2DPoisson_(x, b, 1.0/(n-1), 0.0, 1e-6, 100, False)
        F.synchronize(); ts.append(time.time() - t0)
    print("%-28s %.3f ms (min %.3f)" % (label, sorted(ts)[len(ts)//2]*1e3, min(ts)*1e3))
F.ctx().set_option("mg_wave_target", 4096)
for vx in (1, 2, 1, 2):
    F.ctx().set_option("mg_vx", vx)
    run("mg_vx=%d" % vx)
F.ctx().set_option("mg_vx", 2)
for tgt in (2048, 8192):
    F.ctx().set_option("mg_wave_target", tgt)
    run("mg_vx=2 target=%d" % tgt)
