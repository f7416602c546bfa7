#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native stencil hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no RANK in the environment the script starts its N ranks ITSELF (one child process per GPU,
before anything touches a GPU); under torch.distributed.run it uses the ranks it is given.

One "step" = one pseudo-transient iteration of the 3D diffusion hot path on a 512^3 local grid per GPU
(BASELINE.json configs[1] / configs[3]): fused 7-point update + fused convergence norm, plus, for N > 1, the RCCL
halo exchange (inside libfpr_hip.so, overlapped with the interior update on a second stream) and the all-reduce of
the norm.  Consecutive iterations run in pairs as ONE launch (temporal blocking, bit-identical to two single
launches); every iteration's norm is still computed.  A second, separately reported leg times the
one-iteration-per-launch kernel.

Metric (`value`): effective memory throughput A_eff = 32 B per interior cell per ITERATION, summed over
all GPUs.  `roofline` is physical: bytes one launch has to move at the very least / its measured duration (<= 1 of
peak by construction); the per-iteration (effective) figure is reported beside it.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_PKG = os.path.join(ROOT, "finalprojectrepo.jl_amd")


def _load(name):
    """A module of the package by file path: the launcher must run before anything imports torch or touches a GPU, and the
    package itself (import fpr_amd) is imported only by the worker."""
    spec = importlib.util.spec_from_file_location("fpr_bench_" + name, os.path.join(_PKG, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


launch = _load("launch")          # self_launch, supervise, hb: one process per GPU + watchdog (no torch, no GPU)
legs_mod = _load("benchlegs")     # V-cycle / Navier-Stokes blocks, device_state, power probe, norm check (never loads oracle/)
hb = launch.hb
HBM_PEAK_GBS, A_EFF_BYTES = legs_mod.HBM_PEAK_GBS, legs_mod.A_EFF_BYTES
KT_STEP, KT_STEP2, KT_CORE, KT_STEP3 = legs_mod.KT_STEP, legs_mod.KT_STEP2, legs_mod.KT_CORE, legs_mod.KT_STEP3
timer_read, device_state, norm_check, ns_block, vcycle_block = (legs_mod.timer_read, legs_mod.device_state, legs_mod.norm_check,
                                                                 legs_mod.ns_block, legs_mod.vcycle_block)      # (tools/ and tests/ use these names)
GOLDEN_NORMS, NORM_RTOL = legs_mod.GOLDEN_NORMS, legs_mod.NORM_RTOL


# ------------------------------------------------------------------------------------------------------------
# CPU baselines (oracle/, kind "port"): rank 0, N = 1 only, bounded samples
# ------------------------------------------------------------------------------------------------------------
def cpu_baseline_diffusion(n, budget_s=10.0):
    """C restatement of the reference CPU path timed on the host cores: pseudo-iterations of the fused update + the
    unfused norm pass (as the reference does, part1_kernel_programming.jl:181-191) on an n^3 grid."""
    threads = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle.oracle import Oracle, farr

    orc = Oracle(openmp=True)
    dx = 10.0 / n
    Ht = orc.init_gaussian((n, n, n), dx, dx, dx, (5.0, 5.0, 5.0))
    A, B, R = Ht.copy(order="F"), farr(n, n, n), farr(n, n, n)
    coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
    orc.diffusion3d_step(Ht, A, B, R, *coef)  # warm-up / page touch
    t0 = time.time()
    its = 0
    while True:
        orc.diffusion3d_step(Ht, A, B, R, *coef)
        A, B = B, A
        orc.sumsq_scaled(R, 0.2)
        its += 1
        if time.time() - t0 > budget_s or its >= 50:
            break
    dt = time.time() - t0
    cells = (n - 2) ** 3
    return {
        "value": A_EFF_BYTES * cells * its / dt / 1e9,
        "unit": "GB/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d pseudo-iterations of the fused 7-pt update + separate norm pass at %d^3 (OpenMP, %d threads, %.1f s)"
                  % (its, n, threads, dt),
        "ms_per_step": dt / its * 1e3,
    }


def cpu_baseline_vcycle(n, b_host, css=5, solver=0):
    """One V-cycle of the oracle's MGsolve (multigrid_bench.jl:27-42 protocol: x = 0, b ~ U[0,1), c = 0; coarse grid css^2,
    coarse solver 0 = Jacobi / 1 = cg!) at n^2 on the host cores."""
    import numpy as np

    threads = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle.oracle import Oracle, farr

    orc = Oracle(openmp=True)
    x = farr(n, n)
    reps = 3 if css == 5 else 1          # a five-level cycle with the Jacobi coarse solver is 5140 sweeps of 257^2: seconds
    if reps > 1:
        orc.mgsolve2d(x, b_host, 1.0 / (n - 1), 0.0, 1e-6, 1, False, css, solver)   # warm-up / page touch (one V-cycle)
    ts = []
    for _ in range(reps):
        x[:] = 0.0
        t0 = time.time()
        orc.mgsolve2d(x, b_host, 1.0 / (n - 1), 0.0, 1e-6, 1, False, css, solver)
        ts.append(time.time() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"value": t, "unit": "s", "cores": threads, "kind": "port",
            "sample": "one V-cycle (MGsolve with niters = 1: rms(f) + V-cycle) of the C port at %d^2, coarse grid %d^2, %s coarse "
                      "solver, OpenMP %d threads, %s" % (n, css, "cg!" if solver else "Jacobi", threads,
                                                         "median of 3" if reps > 1 else "one run")}

# ------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", "--local-n", dest="n", type=int, default=512,
                    help="local grid size per GPU (n^3); under torch.distributed.run write --local-n (its parser takes --n for its own --nnodes)")
    ap.add_argument("--dims", type=str, default="", help="process grid, e.g. 2,2,2 (default: z-slabs 1,1,N)")
    ap.add_argument("--as-one-rank-of", type=str, default="",
                    help="a,b,c: run on ONE rank the global problem that a*b*c ranks with --n would run (local grid = "
                         "dims*(n-2)+2, same physical size, same number of untimed steps) -- the control for the norm "
                         "a decomposed run prints")
    ap.add_argument("--check-every", type=int, default=16, help="host convergence check every n iterations")
    ap.add_argument("--prewarm-ms", type=float, default=600.0,
                    help="untimed pre-warm before the W warm-up steps; a COUNT of iterations derived from it (96 x 8 per 300 ms), so that "
                         "the number of iterations behind the printed norm is known; 0.6 s: 150-200 ms into a load step the card's power management "
                         "stalls the launches once for about 100 ms (tools/exp_sustain.py) -- that has to be over before the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the V-cycle / NS blocks")
    ap.add_argument("--no-single-leg", action="store_true", help="skip the one-iteration-per-launch leg")
    ap.add_argument("--variant", type=str, default="", help="k=v,... diffusion kernel options (diff3_*)")
    ap.add_argument("--no-neighbour-leg", action="store_true", help="skip the leg that runs the pair as an interior z-slab rank (self-neighbour over RCCL)")
    ap.add_argument("--no-fuse2", action="store_true", help="main leg with one iteration per launch (k_diff3_march)")
    ap.add_argument("--no-fuse3", action="store_true", help="single rank: main leg with two iterations per launch (k_diff3_march2) instead of three (k_diff3_march3)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / control-plane rehearsal on CPU (gloo): no GPU, no compute, one JSON line")
    ap.add_argument("--choreography", choices=("pairs", "plain"), default="pairs",
                    help="between ranks: pairs = fused pairs (shell chain + both exchanges on the comm stream of the CU-split device "
                         "beside ONE core launch); plain = single steps (boundary slabs -> exchange || interior), no split -- the "
                         "watchdog's fallback")
    ap.add_argument("--watchdog-s", type=float, default=120.0, help="N > 1: seconds without progress of a rank before the attempt is failed")
    ap.add_argument("--watchdog-import-s", type=float, default=420.0, help="the same before the worker has imported torch (fresh box)")
    ap.add_argument("--no-placement", action="store_true", help="allocate the field arrays plainly (no pool of candidates timed pairwise)")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the clocks / power diagnostic (about 2 s, outside the timed regions)")
    ap.add_argument("--no-norm-check", action="store_true", help="N > 1: do not compare the norm with tests/golden/scale_norms.json")
    ap.add_argument("--golden-norms", type=str, default="",
                    help="FILE: control runs on ONE rank (as --as-one-rank-of) of the global problems the decomposed runs solve; "
                         "writes the sum of squares behind the norm after every iteration (tools/make_scale_norms.sh)")
    ap.add_argument("--golden-dims", type=str, default="1,1,2;1,1,4;1,1,8;2,1,1;2,2,1;2,2,2")
    ap.add_argument("--golden-iters", type=int, default=320)
    ap.add_argument("--dry-run-hang", type=int, default=-1, help="(test) this rank stops making progress in the first attempt (--dry-run: before its first collective; else: after it has "
                         "allocated its fields on the GPU)")
    ap.add_argument("--dry-run-hang-always", action="store_true", help="(test) ... and in the fallback attempt too")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="diagnostic for a 1-GPU box: every rank on cuda:0, halo planes and the norm's all-reduce staged "
                         "through the host over gloo (RCCL refuses two ranks on one device).  Runs the N>1 control flow, the "
                         "shell/core choreography and the HIP kernels between real processes; its rates are NOT measurements")
    ap.add_argument("--rehearse-transport", choices=("hosted", "python"), default="hosted",
                    help="with --rehearse-shared-gpu: hosted = the library's exchange code and one-call pairs over fpr_comm_init_hosted "
                         "(default); python = grid.HaloExchanger's twin of the choreography")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        launch.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])   # never returns
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ and os.environ.get("FPR_BENCH_WORKER") != "1":
        launch.supervise(args, os.path.abspath(__file__), sys.argv[1:])          # never returns: this process only watches its worker (and never touches a GPU)

    hb("start")
    import torch
    import torch.distributed as dist

    hb("imported")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and world > 1
    attempt = int(os.environ.get("FPR_BENCH_ATTEMPT", "1"))
    choreography = os.environ.get("FPR_BENCH_CHOREOGRAPHY", args.choreography)
    first_failure = json.loads(os.environ["FPR_BENCH_FIRST_FAILURE"]) if os.environ.get("FPR_BENCH_FIRST_FAILURE") else None
    if use_dist:
        # control plane only (RCCL unique id, barriers, max over ranks): gloo on the host.  The data path -- halo planes
        # and the norm's all-reduce -- is RCCL inside libfpr_hip.so.
        rdzv = os.environ.get("FPR_BENCH_RDZV_FILE")
        if rdzv:
            dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    hb("process_group")

    if args.dry_run:
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        seen = torch.ones(1, dtype=torch.float64)
        if rank == args.dry_run_hang and (attempt == 1 or args.dry_run_hang_always):
            time.sleep(1.0e6)        # (test) a rank that deadlocked: the others wait for it in the barrier below
        if use_dist:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(seen)
            dist.barrier()
        hb("done")
        if rank == 0:
            legs_mod.emit({"metric": "diffusion3d_effective_memory_throughput", "value": None, "unit": "GB/s",
                           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "dry_run": True,
                           "ranks_seen": int(seen.item()), "max_over_ranks": float(t.item()),
                           "config": {"workload": "dry run: control plane only, no GPU", "process_grid": [1, 1, world],
                                      "self_launched": os.environ.get("FPR_BENCH_SELF_LAUNCHED") == "1",
                                      "control_plane": "gloo" if use_dist else "none",
                                      "choreography": choreography, "attempt": attempt},
                           "first_attempt": first_failure}, root=os.environ.get("FPR_BENCH_DETAIL_DIR", ROOT))
        if use_dist:
            dist.destroy_process_group()
        return

    shared = args.rehearse_shared_gpu
    device_index = 0 if shared else local_rank
    torch.cuda.set_device(device_index)
    import fpr_amd

    F = fpr_amd.load(device_index)
    ctx = F.ctx()
    for kv in filter(None, args.variant.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    n = args.n
    if args.golden_norms:
        assert world == 1, "--golden-norms is a series of single-rank control runs"
        legs_mod.golden_norms(F, torch, args)
        return
    dims = tuple(int(x) for x in args.dims.split(",")) if args.dims else (1, 1, world)
    as_one = tuple(int(x) for x in args.as_one_rank_of.split(",")) if args.as_one_rank_of else None
    if as_one:
        assert world == 1, "--as-one-rank-of is a single-rank control run"
        nloc = tuple(d * (n - 2) + 2 for d in as_one)
        phys = as_one
    else:
        nloc = (n, n, n)
        phys = dims
    # --rehearse-shared-gpu: the LIBRARY's transport code and choreography over a host-staged transport (fpr_comm_init_hosted, bytes
    # through gloo) by default; --rehearse-transport python = the Python twin of the choreography (grid.HaloExchanger over gloo)
    shared_kind = ("hosted" if args.rehearse_transport == "hosted" else "dist") if shared else "rccl"
    gg = F.grid.GlobalGrid(*nloc, dims=dims, transport=shared_kind if world > 1 else None)
    if shared and world > 1 and shared_kind == "dist":
        gg.dist = legs_mod.host_staged_p2p(torch, dist)
    rccl_ranks = ctx.L.fpr_comm_size(ctx.h)
    if world > 1 and shared_kind != "dist" and rccl_ranks != world:
        raise RuntimeError("the library's RCCL communicator has %d ranks, the job %d" % (rccl_ranks, world))
    hb("rccl_init")
    # physics as diffusion_3D_kernel_programming with scale_physical_size=true (weak scaling keeps dx fixed)
    lx, ly, lz = (d * 10.0 for d in phys)
    dx, dy, dz = lx / gg.nx_g(), ly / gg.ny_g(), lz / gg.nz_g()
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1.0 / dt, 1.0 / dx, 1.0 / dy, 1.0 / dz, D / dx, D / dy, D / dz)
    # The field arrays.  A single rank runs three iterations per launch (k_diff3_march3), which does not care where its arrays lie
    # (profiles/r6_placement_on_off.txt: 0.916 ms placed, 0.913-0.916 plain on one lease): plain allocations, as the reference's `@zeros`.
    # Ranks with neighbours run fused pairs, whose time depends on the physical pages the arrays received (0.76 against 0.85-0.91 ms): there
    # a pool of candidates is timed once, outside every timed region, and the best-matched five stay (placement.py; --no-placement: plain).
    # (z-slab decompositions run triples between ranks too: fpr_diffusion3d_step3_halo)
    _d3 = as_one or dims     # (a control run --as-one-rank-of dx,dy,dz takes the launches the decomposed run takes: same iteration counts)
    want_fuse3 = _d3[0] == _d3[1] == 1 and not args.no_fuse3 and not args.no_fuse2 and not (world > 1 and choreography == "plain")
    placement = {"selected": False}
    unplaced = {}
    try:
        _free0, _total0 = torch.cuda.mem_get_info()
        unplaced["mem_free_GiB_at_start"], unplaced["mem_total_GiB"] = _free0 / 2.0 ** 30, _total0 / 2.0 ** 30
    except Exception:
        pass
    if args.no_placement or as_one or want_fuse3:
        Ht, Hτ, Hτ3, res, Hτ2 = (F.fzeros(*nloc) for _ in range(5))
    else:
        def trial(arrs):
            """A few fused pairs on the candidate arrays (zeros: the arithmetic does not depend on the values), timed by events."""
            tHt, tA, tC, tR, tB = arrs
            if not F.part1.can_step_τ2(tHt, tA, tB, tC, tR):
                raise RuntimeError("the fused kernel does not serve these arrays")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(3):
                if i == 1:
                    e0.record()
                for _ in range(3 if i else 2):
                    F.part1.diffusion_3D_step_τ2(tHt, tA, tB, tC, tR, *coef)
                    F.part1.diffusion_3D_step_τ2(tHt, tC, tB, tA, tR, *coef)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / 12.0

        try:
            # streamed together at equal offsets: (Ht, field read), (field written, residual) -- the field alternates between Hτ and Hτ3
            Ht, Hτ, Hτ3, res, Hτ2 = F.placement.alloc_fields(5, *nloc, report=placement, pairs=[(0, 1), (0, 2), (2, 3), (1, 3), (1, 2)], trial=trial)
        except Exception as e:       # the search is an optimisation: plainly allocated arrays give the same results
            torch.cuda.empty_cache()
            Ht, Hτ, Hτ3, res, Hτ2 = (F.fzeros(*nloc) for _ in range(5))
            placement.clear()
            placement.update({"selected": False, "error": repr(e)})
    F.part1.init_local_gaussian((lx / 2, ly / 2, lz / 2), dx, dy, dz, Ht, gg.coords)
    Hτ.copy_(Ht)
    # Hτ3: third work buffer for the fused pairs: carries Hτ's boundary; the field alternates between Hτ and Hτ3 while
    # Hτ2 keeps playing the reference's second buffer (its boundary cells / halo planes are all that is read)
    Hτ3.copy_(Ht)
    can_fuse2 = gg.can_step2(Ht, Hτ, Hτ2, Hτ3, res)
    hb("fields")
    if rank == args.dry_run_hang and (attempt == 1 or args.dry_run_hang_always):
        time.sleep(1.0e6)            # (test) this rank stops here with its GPU context alive: the others wait in the first collective
    iters_done = [0]          # pseudo-iterations since the initial state (what tests/golden/scale_norms.json is indexed by)
    K, W, ce = args.steps, args.warmup, max(1, args.check_every)
    sq = torch.zeros(2 * (K + W + ce) + 64, dtype=torch.float64, device=Ht.device)
    sqrtN = math.sqrt(world * nloc[0] * nloc[1] * nloc[2])
    sums = []

    # field state: `cur` is the buffer holding the current field; parity 0 = an "even" buffer (Hτ or Hτ3, the
    # reference's first work buffer and its stand-in), parity 1 = Hτ2.  Fused pairs run even -> even.
    state = {"cur": Hτ, "parity": 0}
    errs = []

    def one_step(sq1):
        if state["parity"] == 0:
            gg.step(Ht, state["cur"], Hτ2, res, *coef, dt, sq1)
            state["cur"], state["parity"] = Hτ2, 1
        else:
            gg.step(Ht, Hτ2, Hτ, res, *coef, dt, sq1)
            state["cur"], state["parity"] = Hτ, 0

    can_fuse3 = want_fuse3 and gg.can_step3(Ht, Hτ, Hτ2, res)

    def run(nsteps, base, fuse2):
        """fuse2: False / 0 = one iteration per launch, True / 2 = fused pairs, 3 = fused triples (remainders as a pair or single steps)"""
        i = 0
        while i < nsteps:
            prev = i
            if fuse2 == 3 and i + 2 < nsteps:
                X = state["cur"]
                Y = Hτ if X is Hτ2 else Hτ2
                gg.step3(Ht, X, Y, res, *coef, dt, sq[base + i:base + i + 3], join=False)
                state["cur"], state["parity"] = Y, state["parity"] ^ 1
                i += 3
            elif fuse2 and state["parity"] == 0 and i + 1 < nsteps:   # (between ranks a pair behind triples runs on the triples' split of the device)
                out = Hτ3 if state["cur"] is Hτ else Hτ
                gg.step2(Ht, state["cur"], Hτ2, out, res, *coef, dt, sq[base + i:base + i + 2], join=False)
                state["cur"] = out
                i += 2
            else:
                one_step(sq[base + i:base + i + 1])
                i += 1
            iters_done[0] += i - prev
            if i // ce > prev // ce or i == nsteps:  # convergence check: all-reduce the chunk (RCCL), host reads it
                chunk = sq[base + (prev // ce) * ce:base + i]
                gg.allreduce_(chunk)
                # ... one launch LATER, as the product's solver does (option diff3_ahead: launches are enqueued ahead of the host's view of
                # the norm): the value travels to pinned host memory behind the all-reduce, the host picks it up after it has enqueued the
                # next launch (or at the end of the run) -- no idle card between the launches on either side of a check
                slot = norm_host[norm_seq[0] & 1]          # (at most two checks are pending: the one just enqueued and the one before it)
                norm_seq[0] += 1
                slot.copy_(chunk[-1:], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                norm_pending.append((slot, ev))
                if len(norm_pending) > 1 or i == nsteps:
                    read_norms(all_of_them=(i == nsteps))

    norm_host = [torch.zeros(1, dtype=torch.float64).pin_memory() for _ in range(2)]
    norm_pending, norm_seq = [], [0]

    def read_norms(all_of_them):
        while norm_pending and (all_of_them or len(norm_pending) > 1):
            slot, ev = norm_pending.pop(0)
            ev.synchronize()
            sums.append(float(slot[0]))
            errs.append(math.sqrt(sums[-1]) / sqrtN)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(fuse2, prewarm_ms):
        """W (+1 if needed to start from an even state) untimed warm-up steps, then EXACTLY K timed steps between
        barriers; returns max-over-ranks wall time and the per-kernel event times of the timed region."""
        # (the event timer creates its 16384 events at its first use -- tens of milliseconds during which the card idles, and the
        # launches behind an idle period of >= 10 ms run up to 20 % slower for ~10 launches, tools/exp_ramp.py: not between the
        # warm-up and the timed region)
        ctx.call("fpr_kernel_timer", 1)
        ctx.call("fpr_kernel_timer", 0)
        # clock ramp, RCCL channel set-up.  Between ranks the number of pre-warm steps must be the SAME everywhere (every
        # step is a collective pattern): a fixed count there, a time budget on a single rank
        # ... and a fixed count on a single rank too (since round 5): the line can then compare its own norm after a KNOWN number
        # of iterations with the committed control (norm_check below); 192 x 8 iterations = 0.6 s of fused pairs at 512^3
        if use_dist or as_one:
            for _ in range(8 if prewarm_ms > 100 else 2):
                run(8, 0, fuse2)
                if use_dist:
                    torch.cuda.synchronize()
                    hb("prewarm")
            torch.cuda.synchronize()
        else:
            for _ in range(max(1, int(round(prewarm_ms / 300.0 * 96)))):
                run(8, 0, fuse2)
                torch.cuda.synchronize()
        # every timed launch of the fused leg must be the fused kernel: an odd W is rounded up to whole pairs (reported as
        # warmup_extra_steps).  The warm-up of the fused leg is all pairs, too: two one-iteration launches right before the timed
        # region (how an odd W used to be evened out) draw 120 W less than the fused kernel, and the ten launches behind such an
        # interlude ran 3-4 % slower than the steady state before and after (profiles/r4_bench_phases.txt, EXPERIMENTS 12.4)
        extra = (W & 1) if (fuse2 and state["parity"] == 0) else 0
        if fuse2 == 3:           # whole triples, and an even number of them: the timed region starts from an even buffer
            extra = (-W) % 6
        run(W + extra, 0, fuse2)
        if use_dist:
            torch.cuda.synchronize()
            hb("warmup")
        if fuse2 and state["parity"] == 1:     # (not reached from an even state; kept for a caller that starts odd)
            run(1, W + extra, False)
            extra += 1
        errs.clear()
        barrier()
        ctx.call("fpr_kernel_timer", 1)
        t0 = time.perf_counter()
        run(K, W + extra, fuse2)
        barrier()
        elapsed = time.perf_counter() - t0
        kt = {kind: timer_read(ctx, kind) for kind in (KT_STEP, KT_STEP2, KT_CORE, KT_STEP3)}
        ctx.call("fpr_kernel_timer", 0)
        hb("timed")
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, kt, extra

    cells = (nloc[0] - 2) * (nloc[1] - 2) * (nloc[2] - 2)
    _cb = gg.boundary_boxes()[1] if world > 1 else ((1, 1, 1), tuple(m - 1 for m in nloc))
    core_cells = (_cb[1][0] - _cb[0][0]) * (_cb[1][1] - _cb[0][1]) * (_cb[1][2] - _cb[0][2])
    min_bytes = A_EFF_BYTES * cells     # what ONE launch must move at the very least, however many iterations it fuses

    def kernel_roofline(kind, kt, traffic_entry):
        ms_tot, cnt = kt[kind]
        ipl = 3 if kind == KT_STEP3 else (2 if kind == KT_STEP2 else 1)
        nbytes = min_bytes
        if world > 1 and kind == KT_STEP3:
            # between ranks a fused triple is ONE core launch (planes [3, nz-3) next to z-neighbours) and the thin shell chain beside it
            nbytes = A_EFF_BYTES * (nloc[0] - 2) * (nloc[1] - 2) * (nloc[2] - 2 - 2 * sum(1 for f in gg.neighbors if f >= 4))
        elif world > 1 and kind == KT_STEP2 and kt[KT_CORE][1]:
            # between ranks a fused pair is ONE core launch on the core stream (the dominant kernel, priced here: 32 B per
            # core cell) and thin shell launches BESIDE it on the comm stream (reported below; they overlap it in time)
            ms_tot, cnt = kt[KT_CORE]
            nbytes = A_EFF_BYTES * core_cells
        elif world > 1:
            # single steps between ranks: boundary slabs, then the interior beside the exchange -- price the pass
            ms_tot = kt[KT_STEP][0] + kt[KT_STEP2][0]
            cnt = max(K // ipl, 1)
        kms = ms_tot / cnt if cnt else 0.0
        ach = nbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "kernel": {KT_STEP3: "k_diff3_march3 (three iterations per launch)", KT_STEP2: "k_diff3_march2 (two iterations per launch)"}.get(
                 kind, "k_diff3_march (one iteration per launch)"),
             "achieved": ach, "frac": ach / HBM_PEAK_GBS, "kernel_ms": kms, "launches": cnt,
             "bytes_per_launch": nbytes, "iterations_per_launch": ipl,
             "accounting": "32 B per interior cell per LAUNCH: read Htau, read Ht, write the new field, write dHdtau (the "
                           "field between two fused iterations never leaves the chip)",
             "effective_achieved": ach * ipl, "effective_frac": ach * ipl / HBM_PEAK_GBS,
             "effective_accounting": "SURVEY 8d: 32 B per interior cell per ITERATION x iterations per launch",
             "traffic": None, "traffic_source": None}
        if world > 1:
            r["kernel"] += ("; between ranks: the CORE launch of the local grid (one per pair / triple, on the core stream of the split device); the "
                            "shell launches run beside it on the comm stream") if (kt[KT_CORE][1] or kind == KT_STEP3) else "; between ranks: all launches of one pass"
            r["launches_by_kind"] = {"single_step_boxes": kt[KT_STEP][1], "fused_boxes": kt[KT_STEP2][1] + kt[KT_CORE][1], "core": kt[KT_CORE][1]}
            r["shell_launches_ms_total"] = kt[KT_STEP][0] + kt[KT_STEP2][0]
        if traffic_entry:
            # counters come from a committed rocprofv3 --pmc run on ANOTHER box (its id in traffic_source): a property of the
            # kernel (bytes per launch), not divided by this box's kernel time
            r["traffic"] = traffic_entry["traffic_bytes_per_launch"]
            r["traffic_source"] = traffic_entry.get("source")
            r["traffic_box"] = traffic_entry.get("box")
            r["traffic_over_algorithmic"] = r["traffic"] / nbytes if nbytes else None
        return r

    def traffic_for(fused):
        """HBM-side bytes per launch from rocprofv3 --pmc passes of this very command (FETCH_SIZE x2 gfx950 correction +
        WRITE_SIZE, collected in separate passes -- tools/pmc_fused2.sh); recorded under profiles/, not measured live."""
        try:
            for tj in json.load(open(os.path.join(ROOT, "profiles", "diffusion_traffic.json")))["entries"]:
                if tj.get("n") == n and world == 1 and not as_one and tj.get("depth", 2 if tj.get("fuse2") else 1) == (int(fused) if fused else 1):
                    return tj
        except Exception:
            pass
        return None

    main_fused = can_fuse2 and not args.no_fuse2 and not (world > 1 and choreography == "plain")
    main_mode = 3 if (main_fused and can_fuse3) else (2 if main_fused else 0)
    dev_before = device_state(device_index) if rank == 0 else None
    elapsed, kt, extra = timed_leg(main_mode, args.prewarm_ms)
    last_err = errs[-1] if errs else None
    last_sumsq = sums[-1] if sums else None
    iters_main = iters_done[0]      # pseudo-iterations behind last_sumsq (pre-warm + warm-up + timed)
    value = A_EFF_BYTES * cells * world * K / elapsed / 1e9
    main_kind = {3: KT_STEP3, 2: KT_STEP2}.get(main_mode, KT_STEP)
    roofline = kernel_roofline(main_kind, kt, traffic_for(main_mode))
    for other_kind in (KT_STEP, KT_STEP2, KT_STEP3):
        if other_kind != main_kind and kt[other_kind][1]:
            roofline.setdefault("other_launches_in_timed_region", []).append({"kind": other_kind, "launches": kt[other_kind][1], "ms_total": kt[other_kind][0]})
    legs = {{3: "fused_triples", 2: "fused_pairs"}.get(main_mode, "single_steps"):
            {"ms_per_step": elapsed / K * 1e3, "value_GBs": value, "kernel_ms": roofline["kernel_ms"],
             "launches": roofline["launches"], "warmup_extra_steps": extra}}
    out = {
        "metric": "diffusion3d_effective_memory_throughput",
        "value": value,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "3D diffusion %d^3 per GPU, fused 7-pt update + norm, %d iteration%s per launch" % (n, max(main_mode, 1), "s" if main_fused else ""),
                   "local_grid": list(nloc), "process_grid": list(dims), "global_grid": [gg.nx_g(), gg.ny_g(), gg.nz_g()],
                   "bytes_per_cell_per_iteration": A_EFF_BYTES,
                   "norm": "fused every iteration; all-reduce + host check every %d" % ce,
                   "halo": ("RCCL ncclSend/ncclRecv groups inside libfpr_hip.so on the comm stream, overlapped with the "
                            "interior update") if world > 1 else "none (1 rank)",
                   "rccl_ranks": rccl_ranks, "control_plane": "gloo" if use_dist else "none",
                   "self_launched": os.environ.get("FPR_BENCH_SELF_LAUNCHED") == "1",
                   "pct_of_hbm_peak_effective_per_gpu": 100.0 * value / world / HBM_PEAK_GBS,
                   "pct_of_hbm_peak_physical_dominant_kernel": 100.0 * roofline["frac"],
                   "last_err": last_err, "last_sumsq": last_sumsq, "iterations_since_start": iters_main,
                   "choreography": (("triples" if main_mode == 3 else "pairs") if main_fused else "plain") if world > 1 else "none (1 rank)", "attempt": attempt,
                   "field_placement": placement},
        "roofline": roofline,
        "legs": legs,
    }
    if rank == 0:
        out["device_state"] = {"before_timed_region": dev_before, "after_timed_region": device_state(device_index),
                               "host": socket.gethostname(), "note": "rocm-smi, one call each, outside the timed region"}
    if first_failure is not None:
        out["first_attempt"] = first_failure      # the watchdog failed the first attempt; this line comes from the fallback
    norm_failed = False
    if not args.no_norm_check:
        # N = 1 (since round 5): the same check against the single-rank control n<N>_dims1,1,1, whose entries at three iteration
        # counts are themselves pinned on the CPU oracle (tests/test_oracle_pins.py::test_scale_norms_...)
        out["norm_check"] = norm_check(n, as_one or dims, iters_main, last_sumsq)
        norm_failed = out["norm_check"].get("ok") is False
    if shared:
        out["rehearsal"] = ("%d ranks sharing one GPU, planes staged through the host over gloo: a control-flow and "
                            "choreography rehearsal, not a measurement" % world)
        out["config"]["halo"] = "REHEARSAL: host-staged gloo (%s)" % ("library transport code over fpr_comm_init_hosted" if shared_kind == "hosted"
                                                                     else "Python HaloExchanger")
    # second leg: the one-iteration-per-launch kernel (north_star's ">= 60 % of HBM peak on the inner update" in
    # the one-pass accounting), with its own event timer; not part of `value`
    if world == 1 and main_fused and not args.no_single_leg:
        state["cur"], state["parity"] = Hτ, 0
        Hτ.copy_(Ht)
        e2, kt2, _ = timed_leg(False, 50.0)
        r2 = kernel_roofline(KT_STEP, kt2, traffic_for(False))
        out["roofline_single"] = r2
        legs["single_steps"] = {"ms_per_step": e2 / K * 1e3, "value_GBs": A_EFF_BYTES * cells * K / e2 / 1e9,
                                "kernel_ms": r2["kernel_ms"], "launches": r2["launches"]}
        if main_mode == 3:
            # the pair kernel on the same arrays (what a rank WITH neighbours runs per launch; the projections below are relative to it)
            state["cur"], state["parity"] = Hτ, 0
            Hτ.copy_(Ht); Hτ3.copy_(Ht)
            e2p, kt2p, _ = timed_leg(2, 50.0)
            out["roofline_pairs"] = kernel_roofline(KT_STEP2, kt2p, traffic_for(2))
            legs["fused_pairs"] = {"ms_per_step": e2p / K * 1e3, "value_GBs": A_EFF_BYTES * cells * K / e2p / 1e9,
                                   "kernel_ms": out["roofline_pairs"]["kernel_ms"], "launches": out["roofline_pairs"]["launches"]}
        import ctypes as C
        # fourth / fifth leg: what a rank WITH neighbours does per pair, on this one card -- a rank that is its own periodic
        # neighbour over the library's RCCL transport (the planes really travel through ncclSend / ncclRecv on the comm stream
        # of the split device).  Links excluded; they are hidden by construction (the chain with both exchanges ends inside
        # the core launch).  Not part of `value`.
        def neighbour_leg(key, periods, drop, eff_key, note, depth=2):
            gp = None
            try:
                if args.no_neighbour_leg:
                    raise RuntimeError("skipped (--no-neighbour-leg)")
                gp = F.grid.GlobalGrid(*nloc, dims=(1, 1, 1), periods=periods, transport="rccl", use_dist=False, drop_faces=drop)
                state["cur"] = Hτ
                if depth == 3 and not gp.can_step3(Ht, Hτ, Hτ2, res):
                    raise RuntimeError("the three-step choreography does not serve this grid")
                def pair_n(nsteps):
                    for i in range(nsteps // depth):
                        if depth == 3:
                            outb = Hτ2 if state["cur"] is Hτ else Hτ
                            gp.step3(Ht, state["cur"], outb, res, *coef, dt, sq[3 * i:3 * i + 3], join=False)
                        else:
                            outb = Hτ3 if state["cur"] is Hτ else Hτ
                            gp.step2(Ht, state["cur"], Hτ2, outb, res, *coef, dt, sq[2 * i:2 * i + 2], join=False)
                        state["cur"] = outb
                # (the communicator has just been set up: the card idled for more than 10 ms, and launches 4-12 behind such a
                # pause are throttled, tools/exp_ramp.py -- 24 pairs of warm-up carry the leg past that)
                pair_n(max(W + (W & 1) + 4, 48))
                gp.join()
                barrier()
                K4 = max(K, 80) if depth == 2 else max(K, 120) // 6 * 6   # steady state: the fork from / join into the compute stream weigh 1/40 each
                t0 = time.perf_counter()
                pair_n(K4)                         # wall time without the event timer (its records stand between the launches)
                gp.join()
                barrier()
                e4 = time.perf_counter() - t0
                ctx.call("fpr_kernel_timer", 1)
                pair_n(K)                          # once more for the kernels' own durations
                gp.join()
                barrier()
                kt4 = {kind: timer_read(ctx, kind) for kind in (KT_STEP, KT_STEP2, KT_CORE, KT_STEP3)}
                ctx.call("fpr_kernel_timer", 0)
                npair, npair_t = max(K4 // depth, 1), max(K // depth, 1)
                plain_pair_ms = depth * legs["fused_triples" if depth == 3 else "fused_pairs"]["ms_per_step"]
                kcore = KT_STEP3 if depth == 3 else KT_CORE
                legs[key] = {
                    "ms_per_step": e4 / (depth * npair) * 1e3, "pair_ms": e4 / npair * 1e3, "plain_pair_ms": plain_pair_ms,
                    "iterations_per_launch": depth,
                    "over_plain_pair": e4 / npair * 1e3 / plain_pair_ms,
                    eff_key: plain_pair_ms / (e4 / npair * 1e3),
                    "core_kernel_ms": kt4[kcore][0] / max(kt4[kcore][1], 1), "core_launches": kt4[kcore][1],
                    "shell_launches_ms_per_pair": (kt4[KT_STEP][0] + kt4[KT_STEP2][0]) / npair_t, "pairs_timed": npair, "comm_units": ctx.L.fpr_comm_cus(ctx.h),
                    "note": note}
            except Exception as e:   # a projection, never required for the GPU number
                legs[key] = {"error": repr(e)}
            finally:
                # whatever happened in the leg: no pair left pending, the single-rank RCCL grid gone, the device unsplit -- the
                # blocks that follow (and the second context) run in the state they expect; a failure here must not mask the leg's own
                for undo in ((lambda: gp.join()) if gp is not None else (lambda: None), F.grid.finalize_global_grid,
                             lambda: ctx.reserve_comm_cus(0), ctx.synchronize):
                    try:
                        undo()
                    except Exception:
                        pass

        neighbour_leg("fused_pairs_as_interior_rank_of_z_slabs", (0, 0, 1), 0, "projected_weak_scaling_efficiency_z_slabs",
                      "one rank, periodic in z = its own neighbour over ncclSend / ncclRecv (RCCL, comm stream of the CU-split "
                      "device); two faces with a neighbour like an interior rank of (1,1,N); a projection from one card, link "
                      "time not included (hidden behind the core launch by construction)")
        if main_mode == 3:
            neighbour_leg("fused_triples_as_interior_rank_of_z_slabs", (0, 0, 1), 0, "projected_weak_scaling_efficiency_z_slabs_triples",
                          "as above with THREE iterations per launch (fpr_diffusion3d_step3_halo: core planes [3, nz-3) in one launch, the two "
                          "planes next to each z-face in three rounds of single-step launches + one-plane exchanges on the comm stream's 32 "
                          "units); relative to this card's plain triples, i.e. to the N = 1 line", depth=3)
        neighbour_leg("fused_pairs_as_rank_of_2x2x2", (1, 1, 1), 0b010101, "projected_weak_scaling_efficiency_2x2x2",
                      "one rank, periodic in x, y and z with the three low faces dropped = one face with a neighbour per dimension, "
                      "the face set of every rank of the reference's (2,2,2) layout (part1_scaling_experiments.jl:40); the x-shell "
                      "in compact strips (csrc/diffusion3d_xstrip.hpp); a projection from one card, link time not included")
    # Clocks and power UNDER each kernel (benchlegs.power_probe): about a second of back-to-back launches per kernel, outside
    # every timed region; its fused-pair rate is the steady-state figure beside the 20-step window's
    if rank == 0 and world == 1 and main_fused and not args.no_power_probe and not as_one:
        try:
            def reset_state():
                state["cur"], state["parity"] = Hτ, 0

            out["power_probe"] = legs_mod.power_probe(torch, device_index, lambda k: run(k, 0, main_mode), lambda k: run(k, 0, False), reset_state)
        except Exception as e:
            out["power_probe"] = {"error": repr(e)}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline_diffusion(n)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        if not args.no_secondary and world == 1:
            state.clear()
            del Ht, Hτ, Hτ2, res, Hτ3
            torch.cuda.empty_cache()
            try:
                out["vcycle"] = vcycle_block(F, cpu_vcycle=None if args.no_cpu_baseline else cpu_baseline_vcycle, place=not args.no_placement)
                out["vcycle_5levels"] = out["vcycle"].pop("vcycle_5levels")
            except Exception as e:
                out["vcycle"] = {"error": repr(e)}
            try:
                out["ns_step"] = ns_block(F)
            except Exception as e:
                out["ns_step"] = {"error": repr(e)}
    if use_dist:
        dist.barrier()                  # the last collective: every rank got through the whole run
        F.grid.finalize_global_grid()
    if norm_failed and os.environ.get("FPR_BENCH_WORKER") == "1" and attempt == 1:
        # every rank holds the same all-reduced sum and takes this exit: the supervisors fail the attempt and start the
        # fallback, whose line carries this record as `first_attempt`
        hb("norm_failed", norm_check=out["norm_check"])
        dist.destroy_process_group()
        sys.exit(3)
    hb("done" if not norm_failed else "norm_failed", **({"norm_check": out["norm_check"]} if norm_failed else {}))
    if rank == 0:
        # (plain allocations: the timed arrays ARE what a host that simply allocates gets)
        out["config"]["unplaced_kernel_ms"] = out["roofline"]["kernel_ms"] if not placement.get("selected") else None
        out["config"]["mem_free_GiB_at_start"], out["config"]["mem_total_GiB"] = unplaced.get("mem_free_GiB_at_start"), unplaced.get("mem_total_GiB")
        try:        # memory this process still holds beyond what it uses (placement leaves nothing behind; the caching allocator's reserve counts)
            out["config"]["mem_held_GiB"] = max(0.0, (torch.cuda.memory_reserved() - torch.cuda.memory_allocated()) / 2.0 ** 30)
        except Exception:
            pass
        legs_mod.emit(out, root=os.environ.get("FPR_BENCH_DETAIL_DIR", ROOT))     # full record -> bench_detail.json; the compact line is the last line of stdout
    if use_dist:
        dist.destroy_process_group()
    if norm_failed:
        print("bench.py: norm check FAILED: %r" % (out["norm_check"],), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
