"""The secondary legs of bench.py and its diagnostics, as functions: the V-cycle blocks (4097^2: 11 grids and five grids), the
Navier-Stokes step around them, clocks / power of the card (librocm_smi64, in process), the norm of a run against the committed
single-rank control, and the host-staged transport of the shared-GPU rehearsal.  bench.py keeps the CLI, the headline leg and the
CPU baselines (the only code that may load oracle/: this module never does -- a caller hands `cpu_vcycle` in)."""
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
A_EFF_BYTES = 32.0     # read Htau + read Ht + write Htau2 + write dHdtau, per interior cell
KT_STEP, KT_STEP2, KT_MG_PRE, KT_MG_POST, KT_MG_SEAM, KT_MG_CG, KT_MG_PATCH, KT_CORE, KT_STEP3 = 0, 1, 2, 3, 4, 5, 6, 7, 8   # include/fpr.h FPR_KT_*


def _read_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


# ------------------------------------------------------------------------------------------------------------
# second half of the metric: V-cycle wall time at 4097^2 (+ the NS step around it)
# ------------------------------------------------------------------------------------------------------------
def timer_read(ctx, kind):
    import ctypes as C

    tot, cnt = C.c_double(0.0), C.c_long(0)
    ctx.call("fpr_kernel_timer_read", int(kind), C.byref(tot), C.byref(cnt))
    return tot.value, cnt.value


def place_vcycle_fields(F, n, b0, placement):
    """x, b for MGsolve at n^2 and the arena arrays the passes over the finest grid stream beside them, laid out by placement class.
    Where the seven arrays lie decides the seam pass's mode (4097^2: 99.5 against 113 us): the two ping-pong partners have to differ in
    placement class from f and from the first coarse level's three arrays -- the partners may share a class, and so may
    everything else (tools/exp_mg_slab2.py, EXPERIMENTS 13.10).  So: two 1 GiB allocations P, Q that copy fastest among a pool
    (fpr_placement_rank: different classes); x, b and the coarse level are windows of P, the partners windows of Q; one timed solve
    decides the orientation and whether the plain allocation was better after all.  `b0`: the right-hand side (device); `placement`
    receives the report.  FPRHip.alloc_vcycle_fields is the same for a Julia host."""
    import warnings

    mg = F.multigrid
    h = 1.0 / (n - 1)
    GiB = 1 << 30
    nc = 1 + (n - 1) // 2

    def window(block, off_bytes, m):
        w = block[off_bytes // 8:off_bytes // 8 + m * m].view(m, m)
        return w.permute(1, 0)

    def windows(P, Q):
        x_, b_ = window(P, 0, n), window(P, 160 << 20, n)
        cs_ = [window(P, (700 << 20) + k * (40 << 20), nc) for k in range(3)]
        t1_, t2_ = window(Q, 0, n), window(Q, 160 << 20, n)
        return x_, b_, t1_, t2_, cs_

    def solve_ms(tx, tb):
        best = None
        for _ in range(3):
            tx.zero_()
            F.synchronize()
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mg.MGsolve_2DPoisson_(tx, tb, h, 0.0, 1e-6, 100, False, opt=mg.MGOpt(), return_history=False)
            F.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best * 1e3

    def trial(arrs):
        tx, tb, t1, t2, cs = windows(*arrs)
        for a in cs:
            a.zero_()
        mg.provide_arena_(n, n, t1, t2)              # the finest level's ping-pong partners travel with the candidates
        mg.provide_arena_coarse_(n, n, *cs)
        tb.copy_(b0)
        return solve_ms(tx, tb)

    if 8 * n * n > (144 << 20) or 8 * nc * nc > (36 << 20):
        raise ValueError("the windows are laid out for grids up to 4097^2")
    x_plain = F.fzeros(n, n)
    placement["plain_allocation_ms"] = solve_ms(x_plain, b0)      # four plain arrays + the library's own arena: what a host gets unplaced
    del x_plain
    P, Q = F.placement.alloc_fields(2, GiB // 8, pool=12, report=placement, pairs=[(0, 1)], trial=trial)
    x, b, t1, t2, cs = windows(P, Q)
    for a in cs:
        a.zero_()
    mg.provide_arena_(n, n, t1, t2)
    mg.provide_arena_coarse_(n, n, *cs)
    b.copy_(b0)
    placement["layout"] = ("x, b and the first coarse level's three arrays are windows of one 1 GiB allocation, the finest level's two "
                           "ping-pong partners windows of another; the two allocations are the pair of the pool that copies fastest "
                           "(different placement classes), orientation by a timed solve")
    if placement.get("trial_ms_best", 0.0) > placement["plain_allocation_ms"]:     # cannot happen on the cards seen; kept honest
        placement["note_plain_was_faster"] = True
    return x, b


def vcycle_block(F, cpu_vcycle=None, steps=5, place=True):
    """MGsolve / V-cycle wall time at 4097^2 (multigrid_bench.jl protocol) with the roofline of its
    dominant kernels and the CPU baseline beside it (`cpu_vcycle(n, b_host, css=5, solver=0) -> dict`, bench.py's oracle leg)."""
    with_cpu = cpu_vcycle is not None
    import warnings

    mg = F.multigrid
    ctx = F.ctx()
    n = 4097
    h = 1.0 / (n - 1)
    b_host = F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")
    b0 = F.asdevice(b_host)
    placement = {}
    if place:
        try:
            x, b = place_vcycle_fields(F, n, b0, placement)
            del b0
        except Exception as e:       # the search is an optimisation: the library's own buffers give the same results
            mg.provide_arena_(n, n, None, None)
            mg.provide_arena_coarse_(n, n, None, None, None)
            x, b = F.fzeros(n, n), b0
            placement.clear()
            placement.update({"selected": False, "error": repr(e)})
    else:
        x, b = F.fzeros(n, n), b0
        placement["selected"] = False
    out = {}
    kern = {}
    for label, css, solver in (("l2_jacobi", 5, mg.jacobi), ("l8_cg", 257, mg.conjugate_gradient),
                               ("l8_jacobi", 257, mg.jacobi)):
        opt = mg.MGOpt()
        opt.coarse_solve_size, opt.coarse_solver = css, solver
        ts = []
        ncyc = 0
        reps = steps if label == "l2_jacobi" else (3 if label == "l8_cg" else 2)
        for i in range(reps + 1):
            x.zero_()
            F.synchronize()
            timed_kernels = i == reps   # last repetition: events around the finest passes / the coarse-solver launches
            if timed_kernels:
                ctx.call("fpr_kernel_timer", 1)
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
            F.synchronize()
            if (i > 0 and not (timed_kernels and label != "l2_jacobi" and reps > 1)) or reps == 0:
                ts.append(time.perf_counter() - t0)     # (event pairs around 28 000 small launches would show in the wall time)
            ncyc = len(hist)
            if timed_kernels:
                if label == "l2_jacobi":
                    for name, kind in (("pre", KT_MG_PRE), ("post", KT_MG_POST), ("seam", KT_MG_SEAM)):
                        ms, cnt = timer_read(ctx, kind)
                        kern[name] = (ms / max(cnt, 1), cnt)
                else:
                    ms, cnt = timer_read(ctx, KT_MG_CG if label == "l8_cg" else KT_MG_PATCH)
                    kern[label] = (ms, cnt)
                ctx.call("fpr_kernel_timer", 0)
        t = sorted(ts)[len(ts) // 2]
        out[label] = {"mgsolve_s": t, "vcycles": ncyc, "s_per_vcycle": t / max(ncyc, 1), "coarse_iters": int(cit),
                      "rel_residual": r / frms}
    # --- byte accounting of one l = 2 V-cycle (11 grids): SURVEY 8d / DESIGN 4.2 ---
    pts = sum((2 ** k + 1) ** 2 for k in range(3, 13))            # smoothing levels k = 12 .. 3 (l = 2 is solved)
    acct_bytes = 132.0 * pts                                      # one pass per operation: 2.955 GB
    spv = out["l2_jacobi"]["s_per_vcycle"]
    ncyc = max(out["l2_jacobi"]["vcycles"], 1)
    N2 = float(n * n)
    pre_ms, pre_cnt = kern.get("pre", (0.0, 0))
    post_ms, post_cnt = kern.get("post", (0.0, 0))
    seam_ms, seam_cnt = kern.get("seam", (0.0, 0))
    pre_bytes, post_bytes, seam_bytes = 28.0 * N2, 26.0 * N2, 30.0 * N2
    # what this implementation must move per V-cycle: coarser levels two passes (28 + 26 B/pt), the finest level the passes
    # that actually ran in the timed solve (launch counts from the event timer) spread over its cycles
    phys_bytes = 54.0 * (pts - N2) + (pre_cnt * pre_bytes + post_cnt * post_bytes + seam_cnt * seam_bytes) / ncyc
    cands = [("seam", seam_ms, seam_bytes, seam_cnt), ("post", post_ms, post_bytes, post_cnt), ("pre", pre_ms, pre_bytes, pre_cnt)]
    dom = max(cands, key=lambda c: c[1] * c[3])                   # the finest-level kernel with the largest share of the time
    gbs = lambda byts, ms: byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    names = {"seam": "k_seam_march_v3 (finest level, between two cycles: correction + 2 post-smoothing sweeps + norm of cycle k, "
                     "2 pre-smoothing sweeps + residual + injection of cycle k+1; two columns per lane, 118 owned of a 128-column strip)",
             "post": "k_smooth2_march_v2<NORM,PROLONG> (finest level: prolongation + correction + 2 sweeps + norm)",
             "pre": "k_smooth2_march_v2<RESTRICT> (finest level: 2 sweeps + residual + injection)"}
    traffic, traffic_src, traffic_box = None, None, None
    try:   # HBM-side bytes of the dominant pass from the committed rocprofv3 --pmc passes (tools/profile_mg.sh), not measured live
        for tj in json.load(open(os.path.join(ROOT, "profiles", "mg_traffic.json")))["entries"]:
            if tj.get("n") == n and tj.get("pass") == dom[0]:
                traffic, traffic_src, traffic_box = tj["traffic_bytes_per_launch"], tj.get("source"), tj.get("box")
    except Exception:
        pass
    roof = {
        "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "kernel": names[dom[0]],
        "achieved": gbs(dom[2], dom[1]), "frac": gbs(dom[2], dom[1]) / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
        "traffic_box": traffic_box, "traffic_over_algorithmic": (traffic / dom[2]) if traffic else None,
        "kernel_ms": dom[1], "bytes_per_launch": dom[2],
        "kernels": {"finest_pre_pass": {"ms": pre_ms, "launches": pre_cnt, "bytes": pre_bytes, "GBs": gbs(pre_bytes, pre_ms),
                                        "accounting": "read u, f; write the smoothed field + res_c, corr_c (1/4 each): 28 B/pt"},
                    "finest_post_pass": {"ms": post_ms, "launches": post_cnt, "bytes": post_bytes, "GBs": gbs(post_bytes, post_ms),
                                         "accounting": "read u, f, corr_c (1/4); write the smoothed field: 26 B/pt"},
                    "finest_seam_pass": {"ms": seam_ms, "launches": seam_cnt, "bytes": seam_bytes, "GBs": gbs(seam_bytes, seam_ms),
                                         "accounting": "read u, f, corr_c (1/4); write the field after 4 sweeps + res_c, corr_c "
                                                       "(1/4 each): 30 B/pt for what two passes (26 + 28 B/pt) do"}},
        "vcycle_physical_bytes": phys_bytes, "vcycle_physical_GBs": phys_bytes / spv / 1e9,
        "vcycle_physical_frac": phys_bytes / spv / 1e9 / HBM_PEAK_GBS,
        "vcycle_accounting_bytes": acct_bytes, "vcycle_effective_GBs": acct_bytes / spv / 1e9,
        "vcycle_effective_frac": acct_bytes / spv / 1e9 / HBM_PEAK_GBS,
        "note": "frac: the dominant finest-level kernel's compulsory bytes / its hipEvent duration; vcycle_physical_*: the bytes "
                "a V-cycle of this implementation must move (coarser levels 54 B/pt, finest level the passes that ran) / wall "
                "time per V-cycle (includes the launch-latency-bound coarse levels); vcycle_effective_*: SURVEY 8d's "
                "one-pass-per-operation accounting (132 B/pt/level = 2.955 GB) / the same time -- above what moves because "
                "sweeps share passes",
    }
    block = {"metric": "vcycle_wall_time_4097sq", "value": spv, "unit": "s", "higher_is_better": False, "dtype": "f64",
             "config": {"workload": "2D Poisson V-cycle 4097^2, 2+2 Jacobi smooths, 11 grids (l=2), Jacobi coarse solver; "
                                    "multigrid_bench.jl protocol (x=0, b~U[0,1), c=0, tol 1e-6)",
                        "mgsolve_s": out["l2_jacobi"]["mgsolve_s"], "vcycles": out["l2_jacobi"]["vcycles"], "field_placement": placement},
             "roofline": roof,
             # BASELINE config 2 read literally ("4096^2, 5 levels"): coarse_solve_size = 257 (l = 8), the coarse 257^2 problem solved by
             # 20 * 257 damped-Jacobi sweeps per cycle (the reference's default coarse solver) or by cg!
             "five_levels_s_per_vcycle": {"jacobi": out["l8_jacobi"]["s_per_vcycle"], "conjugate_gradient": out["l8_cg"]["s_per_vcycle"]},
             "variants": {"five_levels_l8_cg": out["l8_cg"], "five_levels_l8_jacobi": out["l8_jacobi"]}}
    if with_cpu:
        try:
            block["cpu_baseline"] = cpu_vcycle(n, b_host)
        except Exception as e:
            block["cpu_baseline"] = {"value": None, "unit": "s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    # ---- BASELINE config 3 read literally: "4096^2, 5 levels" = grids 4097^2 ... 257^2 (multigrid_bench.jl:27 sweeps l = 2:8; this is
    # l = 8), typed like the block above.  The cycle is bound by its COARSE solve, and that by latency, not bytes: the 257^2 problem
    # (0.5 MB per array) lives in the L2 / in registers, so the roofline of the dominant kernel is a time floor per iteration.
    cg_ms, cg_solves = kern.get("l8_cg", (0.0, 0))
    cg_its = max(out["l8_cg"]["coarse_iters"], 1)
    us_per_cg_it = cg_ms * 1e3 / cg_its if cg_ms > 0 else None
    # krylov.jl's recurrence needs p.p_hat before alpha and r.r before beta: two all-to-all hand-offs per iteration that nothing can hide.
    # Floor = two device-scope store -> poll hand-offs at the idle price MI355X_MICROARCH.md lists (1.0 us cross-XCD, 8 bytes); the sums
    # in front of and behind each hand-off, the operator and the updates are what the kernel adds (profiles/r4_cg_persistent_sections.txt)
    CG_FLOOR_US = 2.0
    pt_ms, pt_launches = kern.get("l8_jacobi", (0.0, 0))
    us_per_launch = pt_ms * 1e3 / pt_launches if pt_launches else None
    jac_sweeps = max(out["l8_jacobi"]["coarse_iters"], 1)
    us_per_sweep = pt_ms * 1e3 / jac_sweeps if pt_ms > 0 else None
    # Floor of a sweep of the 257^2 grid in the persistent kernel: the sweeps of a group (0.24 us each: tools/exp_jacp_prof.py,
    # profiles/r5_jacp_prof.txt, 1.66 us for 7 sweeps on 225 workgroups of 512 threads) plus ONE store -> load hand-off per group at the
    # idle price MI355X_MICROARCH.md lists for <= 4 KB (0.8 us) -- every tile needs all its neighbours' previous group, so the tiles advance
    # as one front and the hand-off cannot hide behind another tile's sweeps (EXPERIMENTS 13.6): (7 x 0.24 + 0.8) / 7 = 0.354 us
    PATCH_SWEEP_US = (7 * 0.24 + 0.8) / 7.0
    five = {"metric": "vcycle_wall_time_4097sq_5levels", "unit": "s", "higher_is_better": False, "dtype": "f64",
            "config": {"workload": "2D Poisson V-cycle 4097^2, 5 grids (4097^2 ... 257^2, l = 8), 2+2 Jacobi smooths; "
                                   "multigrid_bench.jl protocol (x=0, b~U[0,1), c=0, tol 1e-6); coarse solve = cg! or 20*257 "
                                   "damped-Jacobi sweeps (the reference's default coarse solver)"},
            "value": min(out["l8_cg"]["s_per_vcycle"], out["l8_jacobi"]["s_per_vcycle"]),
            "value_is": ("conjugate_gradient" if out["l8_cg"]["s_per_vcycle"] <= out["l8_jacobi"]["s_per_vcycle"] else
                         "jacobi (the reference's default coarse solver)") + ": the faster of the two coarse solvers on this run; both are reported below",
            "conjugate_gradient": {
                "value": out["l8_cg"]["s_per_vcycle"], "unit": "s", "mgsolve_s": out["l8_cg"]["mgsolve_s"], "vcycles": out["l8_cg"]["vcycles"],
                "coarse_iters": out["l8_cg"]["coarse_iters"],
                "roofline": {"bound": "latency", "kernel": "k_cg_persistent (one launch per coarse solve: 64 workgroups of 256 threads, x / r / p / "
                                                           "p_hat in registers, Dot2 dot products, two grid barriers per CG iteration)",
                             "achieved": us_per_cg_it, "peak": CG_FLOOR_US, "unit": "us per CG iteration",
                             "frac": (CG_FLOOR_US / us_per_cg_it) if us_per_cg_it else None,
                             "launches": cg_solves, "kernel_ms_total": cg_ms,
                             "share_of_solve": cg_ms * 1e-3 / out["l8_cg"]["mgsolve_s"] if out["l8_cg"]["mgsolve_s"] > 0 else None,
                             "traffic": None,
                             "note": "achieved = hipEvent time of all k_cg_persistent launches of one solve / CG iterations; peak = two "
                                     "device-scope store -> poll hand-offs per iteration at 1.0 us each (krylov.jl's recurrence needs p.p_hat "
                                     "before alpha and r.r before beta; rounds 2-3 quoted the measured cost of two whole barriers, 4.2 us, as the "
                                     "floor and ran at 6.75 us); frac = floor / achieved"}},
            "jacobi": {
                "value": out["l8_jacobi"]["s_per_vcycle"], "unit": "s", "mgsolve_s": out["l8_jacobi"]["mgsolve_s"], "vcycles": out["l8_jacobi"]["vcycles"],
                "coarse_iters": out["l8_jacobi"]["coarse_iters"],
                "roofline": {"bound": "latency", "kernel": "k_jacobi_persist_tag (up to 32 groups of 7 sweeps of the 257^2 grid per launch: 225 workgroups, 32x32 regions as 2x1 "
                                                           "register patches, rows through LDS; between groups every cell travels from neighbour to "
                                                           "neighbour as a 16-byte {value, tag} granule: one sc1 store, one sc1 load, no flags; exit test behind the launch)",
                             "achieved": us_per_sweep, "peak": PATCH_SWEEP_US, "unit": "us per sweep",
                             "frac": (PATCH_SWEEP_US / us_per_sweep) if us_per_sweep else None,
                             "us_per_launch": us_per_launch, "launches_timed": pt_launches, "sweeps": jac_sweeps, "traffic": None,
                             "note": "achieved = hipEvent time of the coarse-solver launches of one solve / sweeps; peak = (7 sweeps at 0.24 us inside "
                                     "a group + one store -> load hand-off at 0.8 us) / 7: the tiles advance as one front, so a group cannot hide its "
                                     "hand-off; frac = floor / achieved (round 3: one launch per 8 sweeps, 1.5 us per sweep; round 4: flags between "
                                     "neighbours, 0.75 us; the sweeps alone: 0.24 us)"}}}
    if with_cpu:
        for key, solver in (("conjugate_gradient", 1), ("jacobi", 0)):
            try:
                five[key]["cpu_baseline"] = cpu_vcycle(n, b_host, 257, solver)
            except Exception as e:
                five[key]["cpu_baseline"] = {"value": None, "unit": "s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        five["cpu_baseline"] = five["conjugate_gradient" if five["value_is"].startswith("conjugate") else "jacobi"]["cpu_baseline"]
    five["roofline"] = five["conjugate_gradient" if five["value_is"].startswith("conjugate") else "jacobi"]["roofline"]
    block["vcycle_5levels"] = five
    return block


def ns_block(F):
    """BASELINE config 5: Navier-Stokes step around the V-cycle at 2049^2 (buoyancy-driven convection -- the reference
    has no lid-driven cavity), semi-implicit beta = 0.5, tol 1e-7, 3 MG solves per step."""
    p2 = F.part2

    def run(fused, timing=None, max_steps=23, concurrent=True, native=True):
        opt = p2.SimIn_t()
        opt.nx = opt.ny = 2049
        opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
        return p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=max_steps, fused=fused, timing=timing,
                                   concurrent_solves=concurrent, native_step=native)

    run(True, max_steps=5)           # warm-up: arenas of both contexts, worker thread, LDS attributes
    # 20 timed steps (the reference times from the fourth step on, part2.jl:182-184), twice: the loop's host thread shares its cores with whatever
    # else the box runs (one lease in sixty read 1.32 ms where every other read 1.13-1.17); the faster run is the value, both are reported
    res = run(True)
    res2 = run(True)
    runs = [res.t_elapsed / max(res.timed_iters, 1), res2.t_elapsed / max(res2.timed_iters, 1)]
    per_step = min(runs)
    res_py = run(True, native=False)   # the same step composed from Python (thread pool for the W solve)
    res_seq = run(True, concurrent=False)
    tm = {}
    res_t = run(True, timing=tm, max_steps=9)     # diagnostic run: stream synchronisation around every multigrid solve
    mg_per_step = tm.get("mg_s", 0.0) / max(res_t.timed_iters, 1)
    res_u = run(False, max_steps=9)
    return {"metric": "ns_semi_implicit_step_2049sq", "value": per_step, "unit": "s", "timed_steps": res.timed_iters, "runs_s_per_step": runs,
            "multigrid_s_per_step": mg_per_step, "other_s_per_step": max(res_t.t_elapsed / max(res_t.timed_iters, 1) - mg_per_step, 0.0),
            "composed_from_python_s_per_step": res_py.t_elapsed / max(res_py.timed_iters, 1),
            "solves_one_after_the_other_s_per_step": res_seq.t_elapsed / max(res_seq.timed_iters, 1),
            "kernel_by_kernel_s_per_step": res_u.t_elapsed / max(res_u.timed_iters, 1),
            "note": "beta=0.5, Pr=1, Ra=1e6, tol=1e-7, niters=50; three multigrid solves per step (the first T solve hits niters "
                    "as in the reference), the T and the W solve of a step side by side on two contexts (value) or one after the other; "
                    "value: the time loop inside the library (fpr_ns_run2d), software-pipelined -- the next step's S solve runs behind the W solve on the "
                    "second context beside the T solve, same results bit for bit; composed_from_python: the same launches issued piecewise from "
                    "Python in the reference's order; "
                    "multigrid_s_per_step from a diagnostic run with the solves in sequence and synchronised; the step around them "
                    "runs as two passes (fpr_ns_velocity_max2d, fpr_ns_rhs2d) -- "
                    "kernel_by_kernel = the reference's seven kernels + maxima + broadcasts, same results bit for bit"}


# ------------------------------------------------------------------------------------------------------------
# the line the driver keeps: scalars only, < 4 KB (a 23 KB line left BENCH_r05.parsed null)
# ------------------------------------------------------------------------------------------------------------
COMPACT_MAX_BYTES = 4096
DETAIL_FILE = "bench_detail.json"     # everything else (blocks, legs, notes) goes here, next to bench.py


def _g(d, *ks):
    for k in ks:
        if not isinstance(d, dict):
            return None
        d = d.get(k)
    return d


def _num(v, digits=6):
    """Numbers at the precision a reader uses (the detail file keeps every digit)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        if v == int(v) and abs(v) < 2.0 ** 53:
            return int(v) if abs(v) >= 1e6 else v       # byte counts stay exact
        return float("%.*g" % (digits, v))
    return v


def _short(s, n):
    s = "" if s is None else str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def compact_record(out):
    """The ONE line bench.py prints last on stdout, built from the full record `out` (which goes to DETAIL_FILE): the contract's keys, `config`
    / `roofline` / `cpu_baseline` / `norm_check` as flat objects of scalars, one scalar per secondary quantity.  No prose, no nesting
    below those four objects.  compact_check() holds it to COMPACT_MAX_BYTES."""
    r, c = out.get("roofline") or {}, out.get("config") or {}
    fp = c.get("field_placement") or {}
    cb, nc = out.get("cpu_baseline") or {}, out.get("norm_check") or {}
    ds = _g(out, "device_state", "after_timed_region") or {}
    unplaced_ms = c.get("unplaced_kernel_ms")
    bpl = r.get("bytes_per_launch")
    rec = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    rec["config"] = {
        "workload": _short(c.get("workload"), 120), "local_grid": c.get("local_grid"), "process_grid": c.get("process_grid"),
        "rccl_ranks": c.get("rccl_ranks"), "choreography": _short(c.get("choreography"), 16), "attempt": c.get("attempt"),
        "self_launched": c.get("self_launched"), "control_plane": _short(c.get("control_plane"), 8) if c.get("control_plane") else None, "iterations_per_launch": r.get("iterations_per_launch"),
        "placement_selected": bool(fp.get("selected")), "placement_mode": _short(fp.get("mode"), 24) if fp.get("mode") else None,
        "mem_held_GiB": c.get("mem_held_GiB"), "gpu_unique_id": ds.get("unique_id"),
        "sclk_MHz_after": ds.get("sclk_MHz"), "power_W_after": ds.get("power_W"), "power_cap_W": ds.get("power_cap_W"),
        "rehearsal": bool(out.get("rehearsal")) or None, "fallback_reason": _short(_g(out, "first_attempt", "reason"), 160) if out.get("first_attempt") else None,
    }
    rec["roofline"] = {
        "bound": r.get("bound"), "kernel": _short((r.get("kernel") or "").split(" ")[0], 40), "kernel_ms": r.get("kernel_ms"), "launches": r.get("launches"),
        "bytes_per_launch": bpl, "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"),
        "traffic": r.get("traffic"), "traffic_over_algorithmic": r.get("traffic_over_algorithmic"),
        "effective_achieved": r.get("effective_achieved"), "effective_frac": r.get("effective_frac"),
        "unplaced_kernel_ms": unplaced_ms,
        "unplaced_frac": (bpl / (unplaced_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (unplaced_ms and bpl) else None,
        "single_kernel_ms": _g(out, "roofline_single", "kernel_ms"), "single_frac": _g(out, "roofline_single", "frac"),
        "pair_kernel_ms": _g(out, "roofline_pairs", "kernel_ms"), "pair_frac": _g(out, "roofline_pairs", "frac"),
        "steady_ms_per_iteration": _g(out, "power_probe", "fused_pairs", "ms_per_iteration"),      # (key kept: the main fused kernel, whatever its depth)
    }
    if cb:
        rec["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": _short(cb.get("sample"), 140), "ms_per_step": cb.get("ms_per_step")}
    if nc:
        rec["norm_check"] = {"ok": nc.get("ok"), "rel": nc.get("rel"), "iterations": nc.get("iterations")}
    seam_ms = _g(out, "vcycle", "roofline", "kernels", "finest_seam_pass", "ms")
    rec.update({
        "vcycle_s": _g(out, "vcycle", "value"), "vcycle_seam_us": seam_ms * 1e3 if seam_ms else None,
        "vcycle_seam_frac": _g(out, "vcycle", "roofline", "frac"), "vcycle_cpu_s": _g(out, "vcycle", "cpu_baseline", "value"),
        "vcycle5_jacobi_s": _g(out, "vcycle_5levels", "jacobi", "value"), "vcycle5_cg_s": _g(out, "vcycle_5levels", "conjugate_gradient", "value"),
        "ns_step_s": _g(out, "ns_step", "value"),
        "proj_eff_z_slabs": _g(out, "legs", "fused_pairs_as_interior_rank_of_z_slabs", "projected_weak_scaling_efficiency_z_slabs"),
        "proj_eff_z_slabs_triples": _g(out, "legs", "fused_triples_as_interior_rank_of_z_slabs", "projected_weak_scaling_efficiency_z_slabs_triples"),
        "proj_eff_2x2x2": _g(out, "legs", "fused_pairs_as_rank_of_2x2x2", "projected_weak_scaling_efficiency_2x2x2"),
        "detail": DETAIL_FILE,
    })
    for k in ("dry_run", "ranks_seen", "max_over_ranks", "error"):
        if k in out:
            rec[k] = _short(out[k], 200) if isinstance(out[k], str) else out[k]

    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "traffic")

    def walk(v):
        if isinstance(v, dict):
            return {k: walk(x) for k, x in v.items() if x is not None or k in keep}
        if isinstance(v, (list, tuple)):
            return [walk(x) for x in v]
        return _num(v)

    return walk(rec)


def compact_check(rec):
    """The printed line as text, or an exception: strict JSON, ASCII, below COMPACT_MAX_BYTES, the contract's keys present, no string
    longer than 160 characters, nothing nested deeper than one object."""
    line = json.dumps(rec, allow_nan=False, ensure_ascii=True, separators=(", ", ": "))
    if len(line) >= COMPACT_MAX_BYTES:
        raise ValueError("bench line is %d bytes (limit %d)" % (len(line), COMPACT_MAX_BYTES))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "config"):
        if k not in rec:
            raise ValueError("bench line lacks %r" % k)
    for k, v in rec.items():
        for kk, vv in (v.items() if isinstance(v, dict) else ((k, v),)):
            if isinstance(vv, dict) or (isinstance(vv, str) and len(vv) > 160):
                raise ValueError("bench line: %s.%s is not a short scalar" % (k, kk))
    return line


def emit(out, root=ROOT, stream=None):
    """Write the full record to DETAIL_FILE (and a copy under gpurun_out/ when that directory exists), then print the compact line as the
    LAST line of stdout.  Returns the line."""
    stream = stream or sys.stdout
    for path in (os.path.join(root, DETAIL_FILE), os.path.join(root, "gpurun_out", DETAIL_FILE)):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path + ".tmp", "w") as f:
                    json.dump(out, f, indent=1, default=repr)
                    f.write("\n")
                os.replace(path + ".tmp", path)
        except OSError as e:
            print("bench.py: could not write %s: %r" % (path, e), file=sys.stderr)
    line = compact_check(compact_record(out))
    stream.write(line + "\n")
    stream.flush()
    return line


def host_staged_p2p(torch, dist):
    """--rehearse-shared-gpu: a torch.distributed look-alike for grid.HaloExchanger whose planes travel through host
    memory over gloo.  For rehearsing N ranks on ONE card only -- the product transport is RCCL inside the library."""

    class Work:
        def __init__(self, w, host=None, dev=None):
            self.w, self.host, self.dev = w, host, dev     # `host` also keeps a send's staging buffer alive until wait()

        def wait(self):
            self.w.wait()
            if self.dev is not None:
                self.dev.copy_(self.host)      # on the caller's current stream (the comm stream)

    class P2P:
        ReduceOp = dist.ReduceOp

        class P2POp:
            def __init__(self, op, tensor, peer, group=None):
                self.op, self.tensor, self.peer = op, tensor, peer

        @staticmethod
        def isend(*a, **k):
            raise NotImplementedError

        @staticmethod
        def irecv(*a, **k):
            raise NotImplementedError

        def batch_isend_irecv(self, ops):
            torch.cuda.current_stream().synchronize()     # the packed planes are complete
            works = []
            for o in ops:
                if o.op is P2P.irecv:
                    h = torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                    works.append(Work(dist.irecv(h, o.peer), h, o.tensor))
                else:
                    h = o.tensor.cpu().contiguous()
                    works.append(Work(dist.isend(h, o.peer), h))
            return works

        def all_reduce(self, t, op=None, group=None):
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

        def barrier(self, group=None):
            dist.barrier()

        def gather_object(self, *a, **k):
            return dist.gather_object(*a, **k)

    return P2P()


def device_state(index=0):
    """Clocks, power and the power cap of the card, read IN PROCESS from librocm_smi64 (what `rocm-smi --showclocks --showpower`
    prints) -- one call, outside every timed region -- so that a reader can tell a slow box from a slow kernel (the same binary
    ran the dominant kernel in 0.796-0.856 ms on four boxes).  No child process: a process that has initialised the GPU must not
    fork + exec on this pool, and rocm-smi itself is a python script."""
    import ctypes as C

    try:
        L = C.CDLL("librocm_smi64.so")
    except OSError:
        try:
            L = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
        except OSError as e:
            return {"error": repr(e)}

    class Freqs(C.Structure):
        _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]

    out = {}
    try:
        if L.rsmi_init(C.c_uint64(0)) != 0:
            return {"error": "rsmi_init failed"}
        n = C.c_uint32(0)
        L.rsmi_num_monitor_devices(C.byref(n))
        out["rsmi_devices"] = n.value
        dv = C.c_uint32(index if index < n.value else 0)
        for name, kind in (("sclk_MHz", 0), ("fclk_MHz", 1), ("socclk_MHz", 3), ("mclk_MHz", 4)):
            f = Freqs()
            if L.rsmi_dev_gpu_clk_freq_get(dv, C.c_int(kind), C.byref(f)) == 0 and f.num_supported > 0 and f.current < 33:
                out[name] = f.frequency[f.current] / 1e6
                out[name + "_levels"] = [f.frequency[i] / 1e6 for i in range(min(f.num_supported, 33))]
        v = C.c_uint64(0)
        t = C.c_int(0)
        if L.rsmi_dev_power_get(dv, C.byref(v), C.byref(t)) == 0:
            out["power_W"] = v.value / 1e6
            out["power_kind"] = {0: "average", 1: "current socket"}.get(t.value, str(t.value))
        if L.rsmi_dev_power_cap_get(dv, C.c_uint32(0), C.byref(v)) == 0:
            out["power_cap_W"] = v.value / 1e6
        tv = C.c_int64(0)
        for name, sensor in (("temp_edge_C", 0), ("temp_junction_C", 1), ("temp_memory_C", 2)):
            if L.rsmi_dev_temp_metric_get(dv, C.c_uint32(sensor), C.c_int(0), C.byref(tv)) == 0:
                out[name] = tv.value / 1e3
        uid = C.c_uint64(0)
        if L.rsmi_dev_unique_id_get(dv, C.byref(uid)) == 0:
            out["unique_id"] = "0x%x" % uid.value
        buf = C.create_string_buffer(64)
        for name, fn in (("compute_partition", "rsmi_dev_compute_partition_get"), ("memory_partition", "rsmi_dev_memory_partition_get")):
            try:        # SPX/DPX/CPX and NPS1/NPS2/NPS4: how the card's XCDs and HBM stacks are exposed (a property of the box)
                if getattr(L, fn)(dv, buf, C.c_uint32(64)) == 0:
                    out[name] = buf.value.decode(errors="replace")
            except AttributeError:
                pass
        lvl = C.c_int(0)
        if L.rsmi_dev_perf_level_get(dv, C.byref(lvl)) == 0:
            out["perf_level"] = {0: "auto", 1: "low", 2: "high", 3: "manual"}.get(lvl.value, str(lvl.value))
    except Exception as e:
        out["error"] = repr(e)
    return out


# ------------------------------------------------------------------------------------------------------------
# the norm of a decomposed run against the single-domain control (tests/golden/scale_norms.json)
# ------------------------------------------------------------------------------------------------------------
NORM_RTOL = 1.0e-12      # fields are bit-identical to the single-domain run; the sums differ by their order only
GOLDEN_NORMS = os.path.join(ROOT, "tests", "golden", "scale_norms.json")


def norm_check(n, dims, iters, got, path=None):
    """Compare the sum of squares behind the norm after `iters` pseudo-iterations with the one the same GLOBAL problem gave
    on ONE rank (control runs: bench.py --golden-norms, tools/make_scale_norms.sh).  The reference never asserts a multi-rank
    result (test/part1.jl:22 runs one rank); this does."""
    key = "n%d_dims%d,%d,%d" % ((n,) + tuple(dims))
    g = _read_json(path or GOLDEN_NORMS) or {}
    ent = (g.get("entries") or {}).get(key)
    if ent is None or got is None or not (1 <= iters <= len(ent["sumsq"])):
        return {"ok": None, "key": key, "iterations": iters, "got": got,
                "note": "no control value for this problem / iteration count in tests/golden/scale_norms.json"}
    exp = ent["sumsq"][iters - 1]
    rel = abs(got - exp) / abs(exp)
    return {"ok": bool(rel <= NORM_RTOL), "key": key, "iterations": iters, "expected": exp, "got": got, "rel": rel, "rtol": NORM_RTOL,
            "control": "one rank, global grid %s (%s)" % (ent.get("global_grid"), g.get("source"))}


def golden_norms(F, torch, args):
    """Control runs for norm_check: for every process grid in --golden-dims the global problem that grid solves with
    --n cells per rank, on ONE rank, --golden-iters pseudo-iterations as fused pairs; the local sum of squares behind the
    norm after every iteration goes to --golden-norms (merged into an existing file)."""
    ctx = F.ctx()
    n, T = args.n, args.golden_iters + (args.golden_iters & 1)
    data = _read_json(args.golden_norms) or {}
    data.setdefault("entries", {})
    data["source"] = ("bench.py --golden-norms: single-rank control runs of the global problems (as --as-one-rank-of), fused pairs, "
                      "sum((dHdtau*dt)^2) over the interior after every pseudo-iteration since the Gaussian initial state")
    for spec in filter(None, args.golden_dims.split(";")):
        d = tuple(int(x) for x in spec.split(","))
        nloc = tuple(k * (n - 2) + 2 for k in d)
        gg = F.grid.GlobalGrid(*nloc, dims=(1, 1, 1), transport=None)
        lx, ly, lz = (k * 10.0 for k in d)
        dx, dy, dz = lx / gg.nx_g(), ly / gg.ny_g(), lz / gg.nz_g()
        D, dt = 1.0, 0.2
        coef = (min(dx, dy, dz) ** 2 / D / 8.1, 1.0 / dt, 1.0 / dx, 1.0 / dy, 1.0 / dz, D / dx, D / dy, D / dz)
        Ht = F.fzeros(*nloc)
        F.part1.init_local_gaussian((lx / 2, ly / 2, lz / 2), dx, dy, dz, Ht, gg.coords)
        A = Ht.clone(memory_format=torch.preserve_format)
        B = F.fzeros(*nloc)
        C_ = A.clone(memory_format=torch.preserve_format)
        res = F.fzeros(*nloc)
        assert gg.can_step2(Ht, A, B, C_, res), nloc
        sq = torch.zeros(T, dtype=torch.float64, device=Ht.device)
        for i in range(0, T, 2):
            gg.step2(Ht, A, B, C_, res, *coef, dt, sq[i:i + 2])
            A, C_ = C_, A
        torch.cuda.synchronize()
        data["entries"].setdefault("n%d_dims%d,%d,%d" % ((n,) + d), {}).update(      # (an entry's oracle pins, if any, stay)
            {"n": n, "dims": list(d), "global_grid": list(nloc), "sumsq": [float(v) for v in sq.cpu().tolist()]})
        print("golden norms: n=%d dims=%s global %s, %d iterations, last sumsq %.17g" % (n, d, nloc, T, float(sq[-1].item())),
              file=sys.stderr)
        del Ht, A, B, C_, res, sq, gg
        torch.cuda.empty_cache()
    with open(args.golden_norms, "w") as f:
        json.dump(data, f, indent=0)
        f.write("\n")



def probe_under_load(torch, device_index, fn, seconds=1.5):
    """`fn(k)` (k pseudo-iterations of some kernel) back to back for about `seconds` while a host thread reads librocm_smi64 every
    20 ms: ms per iteration, clocks and power under that load (first third of the samples dropped: the ramp)."""
    import threading

    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            d = device_state(device_index)
            samples.append((d.get("sclk_MHz"), d.get("power_W"), d.get("fclk_MHz"), d.get("mclk_MHz"), d.get("temp_junction_C")))
            stop.wait(0.02)

    fn(16)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    t_mid, n_mid = None, 0
    while time.perf_counter() - t0 < seconds:
        fn(32)
        torch.cuda.synchronize()
        n += 32
        if t_mid is None and time.perf_counter() - t0 >= seconds / 3.0:
            t_mid, n_mid = time.perf_counter(), n
    t1 = time.perf_counter()
    dt = t1 - t0
    stop.set()
    th.join(2.0)
    late = samples[len(samples) // 3:] or samples
    avg = lambda k: (sum(x[k] for x in late if x[k] is not None) / max(sum(1 for x in late if x[k] is not None), 1)) if late else None
    mn = lambda k: min((x[k] for x in late if x[k] is not None), default=None)
    # The first third is the ramp, for the rate as for the samples: 100-200 ms into a step from light load to this kernel the card's power
    # management stalls the launches once for about 100 ms (tools/exp_sustain.py: one chunk of 50 pairs takes 145 ms instead of 38, then
    # never again) -- inside a one-second window that reads as +4 ... +13 %.
    settled = (t1 - t_mid) / (n - n_mid) * 1e3 if t_mid is not None and n > n_mid else dt / n * 1e3
    return {"ms_per_iteration": settled, "ms_per_iteration_incl_ramp": dt / n * 1e3, "iterations": n, "samples": len(samples),
            "sclk_MHz_avg": avg(0), "sclk_MHz_min": mn(0),
            "power_W_avg": avg(1), "fclk_MHz_avg": avg(2), "mclk_MHz_avg": avg(3), "temp_junction_C_avg": avg(4)}


def power_probe(torch, device_index, fused_n, single_n, reset_state, seconds=1.5):
    """Clocks and power UNDER each diffusion kernel (a diagnostic outside every timed region): the same launches for about a second
    each while a host thread reads librocm_smi64 every 20 ms.  The fused kernel does twice the FP64 work per byte of the one-iteration
    kernel; whether the card holds its clocks under that load is what separates a slow box from a slow kernel.  `fused_n(k)` /
    `single_n(k)` run k pseudo-iterations; `reset_state()` puts the field state back to an even buffer between the two."""
    reset_state()
    pp = {"fused_pairs": probe_under_load(torch, device_index, fused_n, seconds)}
    reset_state()
    pp["single_steps"] = probe_under_load(torch, device_index, single_n, seconds)
    pp["idle"] = device_state(device_index)
    pp["note"] = ("about 1.5 s of back-to-back launches per kernel, librocm_smi64 read every 20 ms by a host thread; rate, clocks and power over "
                  "the last second (the first third is the ramp: one ~100 ms stall of the card's power management falls into it; "
                  "ms_per_iteration_incl_ramp is the whole window); not part of any timed region")
    return pp
