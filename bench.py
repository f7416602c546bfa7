#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native stencil hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pseudo-transient iteration of the 3D diffusion hot path on a 512^3 local grid per
GPU (BASELINE.json configs[1] / configs[3]): fused 7-point update + fused convergence norm, plus, for
N > 1, the RCCL halo exchange overlapped with the interior update and the all-reduce of the norm.
Consecutive iterations run in pairs as ONE fused launch (temporal blocking, bit-identical to two single
launches; --no-fuse2 times the one-iteration-per-launch path); every iteration's norm is still computed.
Metric: effective memory throughput A_eff = 32 B per interior cell per iteration, summed
over all GPUs.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
A_EFF_BYTES = 32.0     # read Htau + read Ht + write Htau2 + write dHdtau, per interior cell


def cpu_baseline(n, budget_s=12.0):
    """C restatement of the reference CPU path (oracle/, kind 'port') timed on the host cores:
    a bounded sample of the same workload -- pseudo-iterations of the fused update + the unfused norm
    pass (as the reference does, part1_kernel_programming.jl:181-191) on an n^3 grid."""
    import numpy as np

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


def vcycle_secondary(F, steps=3):
    """Secondary metric: MGsolve / V-cycle wall time at 4097^2 (multigrid_bench.jl protocol)."""
    import numpy as np

    mg = F.multigrid
    n = 4097
    h = 1.0 / (n - 1)
    b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    x = F.fzeros(n, n)
    out = {}
    for label, css, solver in (("l2_jacobi", 5, mg.jacobi), ("l8_cg", 257, mg.conjugate_gradient),
                               ("l8_jacobi", 257, mg.jacobi)):
        opt = mg.MGOpt()
        opt.coarse_solve_size, opt.coarse_solver = css, solver
        ts = []
        ncyc = 0
        for i in range(steps if label != "l8_jacobi" else 1):
            x.zero_()
            F.synchronize()
            t0 = time.time()
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=opt, return_history=True)
            F.synchronize()
            ts.append(time.time() - t0)
            ncyc = len(hist)
        t = sorted(ts)[len(ts) // 2]
        out[label] = {"mgsolve_s": t, "vcycles": ncyc, "s_per_vcycle": t / max(ncyc, 1), "coarse_iters": int(cit),
                      "rel_residual": r / frms}
    # BASELINE config 5: Navier-Stokes step around the V-cycle at 2049^2 (buoyancy-driven convection --
    # the reference has no lid-driven cavity), semi-implicit beta=0.5, tol 1e-7, 3 MG solves per step
    try:
        p2 = F.part2
        opt = p2.SimIn_t()
        opt.nx = opt.ny = 2049
        opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
        res = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=7)
        out["ns_semi_implicit_2049sq"] = {"s_per_step": res.t_elapsed / max(res.timed_iters, 1), "timed_steps": res.timed_iters,
                                          "note": "beta=0.5, Pr=1, Ra=1e6, tol=1e-7, niters=50; T solve hits niters as in the reference"}
    except Exception as e:
        out["ns_semi_implicit_2049sq"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=512, help="local grid size per GPU (n^3)")
    ap.add_argument("--dims", type=str, default="", help="process grid, e.g. 2,2,2 (default: z-slabs 1,1,N)")
    ap.add_argument("--check-every", type=int, default=16, help="host convergence check every n iterations")
    ap.add_argument("--prewarm-ms", type=float, default=300.0, help="untimed pre-warm before the W warm-up steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--variant", type=str, default="", help="k=v,... diffusion kernel options (diff3_*)")
    ap.add_argument("--no-fuse2", action="store_true", help="one iteration per launch (k_diff3_march) instead of fused pairs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # launched by torch.distributed.run
    if use_dist:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world), file=sys.stderr)

    import fpr_amd

    F = fpr_amd.load(local_rank)
    ctx = F.ctx()
    for kv in filter(None, args.variant.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    n = args.n
    dims = tuple(int(x) for x in args.dims.split(",")) if args.dims else (1, 1, world)
    gg = F.grid.GlobalGrid(n, n, n, dims=dims)
    # physics as diffusion_3D_kernel_programming with scale_physical_size=true (weak scaling keeps dx fixed)
    lx, ly, lz = (d * 10.0 for d in dims)
    dx, dy, dz = lx / gg.nx_g(), ly / gg.ny_g(), lz / gg.nz_g()
    D, dt = 1.0, 0.2
    dτ = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dτ, 1.0 / dt, 1.0 / dx, 1.0 / dy, 1.0 / dz, D / dx, D / dy, D / dz)
    Ht = F.fzeros(n, n, n)
    F.part1.init_local_gaussian((lx / 2, ly / 2, lz / 2), dx, dy, dz, Ht, gg.coords)
    Hτ = Ht.clone(memory_format=torch.preserve_format)
    Hτ2 = F.fzeros(n, n, n)
    res = F.fzeros(n, n, n)
    # third work buffer for the fused pairs: carries Hτ's boundary; the field alternates between Hτ and Hτ3 while
    # Hτ2 keeps playing the reference's second buffer (its boundary cells / halo planes are all that is read)
    Hτ3 = Hτ.clone(memory_format=torch.preserve_format)
    fuse2 = (not args.no_fuse2) and gg.can_step2(Ht, Hτ, Hτ2, Hτ3, res)
    if not fuse2:
        del Hτ3
    K, W, ce = args.steps, args.warmup, max(1, args.check_every)
    sq = torch.zeros(K + W + ce, dtype=torch.float64, device=Ht.device)
    errs = []
    sqrtN = math.sqrt(world * n ** 3)

    # field state: `cur` is the buffer holding the current field; parity 0 = an "even" buffer (Hτ or Hτ3, the
    # reference's first work buffer and its stand-in), parity 1 = Hτ2.  Fused pairs run even -> even.
    state = {"cur": Hτ, "parity": 0}

    def one_step(sq1):
        if state["parity"] == 0:
            gg.step(Ht, state["cur"], Hτ2, res, *coef, dt, sq1)
            state["cur"], state["parity"] = Hτ2, 1
        else:
            gg.step(Ht, Hτ2, Hτ, res, *coef, dt, sq1)
            state["cur"], state["parity"] = Hτ, 0

    def run(nsteps, base):
        i = 0
        while i < nsteps:
            prev = i
            if fuse2 and state["parity"] == 0 and i + 1 < nsteps:
                out = Hτ3 if state["cur"] is Hτ else Hτ
                gg.step2(Ht, state["cur"], Hτ2, out, res, *coef, dt, sq[base + i:base + i + 2])
                state["cur"] = out
                i += 2
            else:
                one_step(sq[base + i:base + i + 1])
                i += 1
            if i // ce > prev // ce or i == nsteps:  # convergence check: all-reduce the chunk, host reads it
                chunk = sq[base + (prev // ce) * ce:base + i]
                if use_dist:
                    dist.all_reduce(chunk)
                errs.append(math.sqrt(float(chunk[-1].item())) / sqrtN)

    # untimed pre-warm (clock ramp, RCCL channel set-up), then the W warm-up steps of the contract
    tpre = time.perf_counter()
    while time.perf_counter() - tpre < args.prewarm_ms * 1e-3:
        run(8, 0)
        torch.cuda.synchronize()
    errs.clear()
    run(W, 0)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.call("fpr_kernel_timer", 1)
    t0 = time.perf_counter()
    run(K, W)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    import ctypes as C

    ktot, kcnt = C.c_double(0.0), C.c_long(0)
    ctx.call("fpr_kernel_timer_read", C.byref(ktot), C.byref(kcnt))
    ctx.call("fpr_kernel_timer", 0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=Ht.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cells = (n - 2) ** 3
    value = A_EFF_BYTES * cells * world * K / elapsed / 1e9
    # dominant kernel: HIP events around every diffusion-kernel launch of the timed region (fpr_kernel_timer), on the
    # library's compute stream.  At N = 1 each launch is one k_diff3_march2 (two iterations) / k_diff3_march (one).
    launches = max(1, kcnt.value)
    its_per_launch = K / launches
    kernel_ms = ktot.value / launches                     # average launch duration
    alg_bytes_launch = A_EFF_BYTES * cells * its_per_launch   # SURVEY 8d per-unit figure x units one launch processes
    achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic = None
    try:  # HBM-side bytes per launch from the rocprofv3 PMC passes committed under profiles/ (same command)
        for tj in json.load(open(os.path.join(ROOT, "profiles", "diffusion_traffic.json")))["entries"]:
            if tj.get("n") == n and world == 1 and tj.get("fuse2") == bool(fuse2):
                traffic = tj["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
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
        "config": {"workload": "3D pseudo-transient diffusion, %d^3 cells per GPU, fused 7-pt update + fused norm%s"
                               % (n, ", two iterations per launch (temporal blocking)" if fuse2 else ""),
                   "local_grid": [n, n, n], "process_grid": list(dims), "global_grid": [gg.nx_g(), gg.ny_g(), gg.nz_g()],
                   "bytes_per_cell": A_EFF_BYTES, "norm": "fused every iteration; all-reduce + host check every %d" % ce,
                   "halo": "RCCL isend/irecv on comm stream overlapped with interior update" if world > 1 else "none (1 rank)",
                   "pct_of_hbm_peak_per_gpu": 100.0 * value / world / HBM_PEAK_GBS,
                   "last_err": errs[-1] if errs else None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "k_diff3_march2" if fuse2 else "k_diff3_march", "kernel_ms": kernel_ms,
                     "launches": launches, "iterations_per_launch": its_per_launch,
                     "algorithmic_bytes_per_launch": alg_bytes_launch,
                     # what one launch has to move at the very least (read Htau, Ht; write the new field, dHdtau
                     # ONCE, however many iterations it fuses) and the rate against that figure: with two iterations
                     # per launch `frac` can exceed 1 -- the intermediate field never travels to HBM
                     "min_bytes_per_launch": A_EFF_BYTES * cells,
                     "achieved_min_bytes": A_EFF_BYTES * cells / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                     "achieved_traffic": traffic / (kernel_ms * 1e-3) / 1e9 if (traffic and kernel_ms > 0) else None},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(n)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        if not args.no_secondary and world == 1:
            state.clear()
            del Ht, Hτ, Hτ2, res
            if fuse2:
                del Hτ3
            torch.cuda.empty_cache()
            try:
                out["secondary"] = {"metric": "mgsolve_wall_time_4097sq", "unit": "s", "results": vcycle_secondary(F)}
            except Exception as e:
                out["secondary"] = {"error": repr(e)}
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
