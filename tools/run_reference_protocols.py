"""Runs the reference's own benchmark protocols (part1_scaling_experiments.jl, multigrid_bench.jl) on this
library and writes CSVs with the reference's schemas (finalprojectrepo.jl_amd/experiments.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fpr_amd
F = fpr_amd.load(0)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
for f in ("bench_diffusion_scaling_gpu.csv", "bench_multigrid_gpu.csv"):
    if os.path.exists(os.path.join(out, f)):
        os.remove(os.path.join(out, f))
rows = F.experiments.part1_scaling_experiments(os.path.join(out, "bench_diffusion_scaling_gpu.csv"), n=128, ttot=2.0, tol=1e-6)
for r in rows:
    print("diffusion 128^3 strong=%s shmem=%s: delta_t %.3f s, %d iterations in total, Throughput(ref accounting) %.1f GB/s"
          % (r["strong_scaling"], r["use_shared_memory"], r["delta_t"], r["_iters"], r["Throughput"] / 1e9))
rows = F.experiments.multigrid_bench(os.path.join(out, "bench_multigrid_gpu.csv"), ks=range(4, 14), samples=5)
for r in rows:
    if r["execution_policy"] == "parallel":
        print("multigrid k=%d l=%d %s: median %.6f s" % (r["k"], r["l"], r["coarse_solver"], r["median_time"]))
