"""NEXT row 8f-2: the benchmark harnesses write the reference's CSV schemas."""
import csv

import pytest

pytestmark = pytest.mark.gpu

# first lines of the reference's published result files (benchmark-results/*.csv)
REF_DIFFUSION_HEADER = "delta_t,Work,Performance,Memory,Intensity,Throughput,use_shared_memory,use_gpu,strong_scaling,n_threads,n_mpi_ranks"
REF_MULTIGRID_HEADER = "execution_policy,coarse_solver,k,l,median_time,mean_time,std_time,seed,use_gpu,nthreads"
REF_NS_HEADER = "nx,ny,Pr,beta,t_elapsed,timed_iters"


def test_diffusion_scaling_csv(fpr, tmp_path):
    ex = fpr.experiments
    f = tmp_path / "bench_diffusion_scaling_gpu.csv"
    rows = ex.part1_scaling_experiments(str(f), n=24, ttot=0.8, tol=1e-6)
    lines = f.read_text().splitlines()
    assert lines[0] == REF_DIFFUSION_HEADER and len(lines) == 5
    for r in rows:
        cells = 22 ** 3
        assert r["Work"] >= 0 and r["n_mpi_ranks"] == 1
        # part1_kernel_programming.jl:209-217 accounting: Work/(27*cells) = timed iterations (after 3 warm-up steps)
        assert abs(r["Work"] / (27 * cells) - round(r["Work"] / (27 * cells))) < 1e-9
        if r["Work"] > 0:
            assert abs(r["Intensity"] - 27 / (8 * (7 if r["use_shared_memory"] else 15))) < 1e-12
    rec = list(csv.DictReader(open(f)))
    assert rec[0]["use_shared_memory"] == "true" and rec[1]["use_shared_memory"] == "false" and rec[0]["use_gpu"] == "true"


def test_multigrid_bench_csv(fpr, tmp_path):
    ex, mg = fpr.experiments, fpr.multigrid
    f = tmp_path / "bench_multigrid_gpu.csv"
    rows = ex.multigrid_bench(str(f), ks=(6, 7), samples=2)
    lines = f.read_text().splitlines()
    assert lines[0] == REF_MULTIGRID_HEADER
    # k=6 -> l in 2:2, k=7 -> l in 2:3 ; 2 solvers x 2 policies each
    assert len(rows) == (1 + 2) * 4 == len(lines) - 1
    assert {r["execution_policy"] for r in rows} == {"parallel", "parallel_shmem"}
    assert {r["coarse_solver"] for r in rows} == {"jacobi", "conjugate_gradient"}
    assert all(r["median_time"] > 0 for r in rows)


def test_ns_experiment_csv(fpr, tmp_path):
    ex = fpr.experiments
    f = tmp_path / "part2_semi_implicit_vs_explicit_experiment_results.csv"
    rows = ex.semi_implicit_vs_explicit(str(f), nx=257, ny=65, Prs=(1e-1,), betas=(0.0, 0.5), ttot=1e9, max_steps=5)
    lines = f.read_text().splitlines()
    assert lines[0] == REF_NS_HEADER and len(lines) == 3
    assert all(r["_steps"] == 5 and r["timed_iters"] == 2.0 for r in rows)
