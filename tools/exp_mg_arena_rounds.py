"""Is the V-cycle's slow mode a property of the card, or of the level arena the context happened to allocate?  R rounds, each with a fresh
context (= fresh coarse-level buffers inside the library) and bench.py's own search over placed x / b / ping-pong partners.
usage: exp_mg_arena_rounds.py [rounds]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd
F = fpr_amd.load(0)
n = 4097
h = 1.0 / (n - 1)
b_host = F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")
keep = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    if rnd:
        keep.append(torch.empty((53 + 97 * rnd) << 20, dtype=torch.uint8, device="cuda"))     # shifts what the next context receives
        F.reset()
    mg = F.multigrid
    b0 = F.asdevice(b_host)
    def trial(arrs):
        tx, tb, t1, t2 = arrs
        mg.provide_arena_(n, n, t1, t2)
        tb.copy_(b0)
        best = None
        for _ in range(3):
            tx.zero_(); F.synchronize(); t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mg.MGsolve_2DPoisson_(tx, tb, h, 0.0, 1e-6, 100, False, opt=mg.MGOpt(), return_history=False)
            F.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best * 1e3
    # first the library's own buffers (no placement at all)
    x0 = F.fzeros(n, n)
    mg.provide_arena_(n, n, None, None)
    t_own = None
    for _ in range(3):
        x0.zero_(); F.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mg.MGsolve_2DPoisson_(x0, b0, h, 0.0, 1e-6, 100, False, opt=mg.MGOpt(), return_history=False)
        F.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        t_own = dt if t_own is None or dt < t_own else t_own
    rep = {}
    arrs = F.placement.alloc_fields(4, n, n, pool=10, min_bytes=64 << 20, report=rep, pairs=[(0, 1), (2, 1), (3, 1), (2, 3), (0, 2)], trial=trial, trials=3)
    # the big arrays fixed: now the three arrays of the first coarse level
    tx, tb, t1, t2 = arrs
    mg.provide_arena_(n, n, t1, t2)
    tb.copy_(b0)
    nc = 1 + (n - 1) // 2
    def trial_c(cs):
        for a in cs:
            a.zero_()
        mg.provide_arena_coarse_(n, n, *cs)
        best = None
        for _ in range(3):
            tx.zero_(); F.synchronize(); t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mg.MGsolve_2DPoisson_(tx, tb, h, 0.0, 1e-6, 100, False, opt=mg.MGOpt(), return_history=False)
            F.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best * 1e3
    repc = {}
    cs = F.placement.alloc_fields(3, nc, nc, pool=8, min_bytes=16 << 20, report=repc, trial=trial_c, trials=3, spacer_bytes=2 << 30)
    print("round %d: own buffers %.3f ms per solve; placed: best %.3f first %.3f worst %.3f ms (%d trials, pool %d -> %d); coarse triple placed too: best %.3f first %.3f worst %.3f ms (%d trials, pool %d -> %d)" % (
        rnd, t_own, rep["trial_ms_best"], rep["trial_ms_first"], rep["trial_ms_worst"], rep["trials"], rep["pool_first"], rep["pool"],
        repc["trial_ms_best"], repc["trial_ms_first"], repc["trial_ms_worst"], repc["trials"], repc["pool_first"], repc["pool"]), flush=True)
    mg.provide_arena_coarse_(n, n, None, None, None)
    mg.provide_arena_(n, n, None, None)
    del arrs, x0, b0, cs, tx, tb, t1, t2
