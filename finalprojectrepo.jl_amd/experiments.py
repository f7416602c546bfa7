"""NEXT row 8f-2 -- benchmark harnesses with the reference's result schemas, so that numbers measured
with this library drop into the reference's CSV / plotting pipeline unchanged.

    part1_scaling_experiments   scripts-part1/part1_scaling_experiments.jl:27-77 (columns :50-60)
    multigrid_bench             scripts-part2/multigrid_bench.jl:27-63           (columns :43-54)
    semi_implicit_vs_explicit   scripts-part2/part2_semi_implicit_vs_explicit_experiments.jl:35-60

Each function appends rows to a CSV (created with the reference's header when missing) and returns the rows.
"""
import csv
import os
import statistics
import time
import warnings

DIFFUSION_COLUMNS = ["delta_t", "Work", "Performance", "Memory", "Intensity", "Throughput", "use_shared_memory",
                     "use_gpu", "strong_scaling", "n_threads", "n_mpi_ranks"]
MULTIGRID_COLUMNS = ["execution_policy", "coarse_solver", "k", "l", "median_time", "mean_time", "std_time", "seed",
                     "use_gpu", "nthreads"]
NS_COLUMNS = ["nx", "ny", "Pr", "beta", "t_elapsed", "timed_iters"]


def _jl(v):
    """Julia-style CSV cell (true/false for Bool)."""
    if isinstance(v, bool):
        return "true" if v else "false"
    return v


def _append(path, columns, rows):
    if not path:
        return
    new = not os.path.exists(path)
    with open(path, "a", newline="") as fh:
        w = csv.writer(fh)
        if new:
            w.writerow(columns)
        for r in rows:
            w.writerow([_jl(r[c]) for c in columns])


def part1_scaling_experiments(filename=None, n=128, ttot=2.0, tol=1e-6, strong_scaling_modes=(True, False),
                              shared_memory_modes=(True, False), global_grid_factory=None, **solver_kwargs):
    """part1_scaling_experiments.jl:27-77.  strong scaling divides a 128^3 GLOBAL grid by dims
    (:35-41); weak scaling solves n^3 per rank with scale_physical_size=true (:47)."""
    from . import grid, part1

    rows = []
    for strong in strong_scaling_modes:
        probe = grid.GlobalGrid(3, 3, 3)
        dims = probe.dims
        local = tuple(n // d for d in dims) if strong else (n, n, n)
        for use_shmem in shared_memory_modes:
            gg = (global_grid_factory or grid.GlobalGrid)(*local)
            _, _, b, info = part1.diffusion_3D_kernel_programming(
                nx=local[0], ny=local[1], nz=local[2], ttot=ttot, tol=tol, use_shared_memory=use_shmem,
                scale_physical_size=not strong, global_grid=gg, return_device=True, **solver_kwargs)
            row = dict(delta_t=b.Δt, Work=b.Work, Performance=b.Performance, Memory=b.Memory, Intensity=b.Intensity,
                       Throughput=b.Throughput, use_shared_memory=use_shmem, use_gpu=True, strong_scaling=strong,
                       n_threads=1, n_mpi_ranks=gg.nprocs)
            row["_iters"] = sum(info["iters"])
            rows.append(row)
            if gg.me == 0:
                _append(filename, DIFFUSION_COLUMNS, [row])
    return rows


def multigrid_bench(filename=None, ks=range(4, 14), max_l=8, solvers=None, policies=None, samples=5, seed=1):
    """multigrid_bench.jl:27-63: MGsolve_2DPoisson!(x=0, b=rand, h, 0, 1e-6, 100, false) for k in 4:13,
    l in 2:min(k-4, 8), both coarse solvers and both execution policies; BenchmarkTools median/mean/std."""
    from . import asdevice, fzeros, synchronize
    from . import multigrid as mg
    from .part2 import splitmix64_uniform

    solvers = solvers if solvers is not None else (mg.jacobi, mg.conjugate_gradient)
    policies = policies if policies is not None else (mg.parallel, mg.parallel_shmem)
    rows = []
    for k in ks:
        n = 2 ** k + 1
        h = 1.0 / (n - 1)
        b = asdevice(splitmix64_uniform(n * n, seed).reshape((n, n), order="F"))  # `@rand` replaced by a counter RNG
        x = fzeros(n, n)
        for l in range(2, min(k - 4, max_l) + 1):
            for solver in solvers:
                for policy in policies:
                    opt = mg.MGOpt()
                    opt.execution_policy, opt.coarse_solve_size, opt.coarse_solver = policy, 2 ** l + 1, solver
                    ts = []
                    for _ in range(samples + 1):  # first sample = warm-up (arena allocation)
                        x.zero_()
                        synchronize()
                        t0 = time.perf_counter()
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=opt)
                        synchronize()
                        ts.append(time.perf_counter() - t0)
                    ts = ts[1:]
                    row = dict(execution_policy=policy.name, coarse_solver=solver.name, k=k, l=l,
                               median_time=statistics.median(ts), mean_time=statistics.mean(ts),
                               std_time=statistics.pstdev(ts) if len(ts) > 1 else 0.0, seed=seed, use_gpu=True, nthreads=1)
                    rows.append(row)
                    _append(filename, MULTIGRID_COLUMNS, [row])
    return rows


def semi_implicit_vs_explicit(filename=None, nx=2049, ny=513, Prs=(1e-3, 1e-2, 1e-1, 1.0, 10.0), betas=(0.0, 0.5, 1.0),
                              ttot=0.005, max_steps=None):
    """part2_semi_implicit_vs_explicit_experiments.jl:35-60."""
    from . import part2

    rows = []
    for Pr in Prs:
        for beta in betas:
            opt = part2.SimIn_t()
            opt.nx, opt.ny, opt.Pr, opt.tol, opt.beta, opt.ttot = nx, ny, Pr, 1.0e-7, beta, ttot
            res = part2.navier_stokes_2D(opt=opt, verbose=False, max_steps=max_steps)
            row = dict(nx=nx, ny=ny, Pr=Pr, beta=beta, t_elapsed=res.t_elapsed, timed_iters=float(res.timed_iters))
            row["_steps"] = res.steps
            rows.append(row)
            _append(filename, NS_COLUMNS, [row])
    return rows
