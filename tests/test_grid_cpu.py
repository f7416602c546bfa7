"""CPU tests of the multi-rank path: Cartesian topology, halo-plane conventions and the
torch.distributed exchange, exercised with the `gloo` backend on world sizes 2, 4 and 8.

Each rank advances its shard with the ORACLE's fused step (numpy) and refreshes halos through the
product's HaloExchanger (with CPU pack/unpack supplied by this test: the HIP pack kernels need a GPU
and are checked against the same plane convention in tests/test_gpu_halo.py).  Acceptance: the N-shard
run equals the single-domain oracle run on the equivalent global grid bit for bit; the all-reduced
norm agrees to 1e-13."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fpr_amd

grid = fpr_amd.pkg.grid


def test_dims_create_matches_reference_table():
    # part1_scaling_experiments.jl:35-41
    assert grid.dims_create(1) == (1, 1, 1)
    assert grid.dims_create(2) == (2, 1, 1)
    assert grid.dims_create(4) == (2, 2, 1)
    assert grid.dims_create(8) == (2, 2, 2)
    assert grid.dims_create(6) == (3, 2, 1)
    assert np.prod(grid.dims_create(12)) == 12


def test_topology_and_global_sizes():
    g = grid.GlobalGrid(10, 12, 14, dims=(1, 1, 1), use_dist=False)
    assert (g.nx_g(), g.ny_g(), g.nz_g()) == (10, 12, 14) and g.neighbors == {} and g.coords == (0, 0, 0)
    assert g.x_g(1, 0.5) == 0.0 and g.x_g(3, 0.5) == 1.0
    boxes, inner = g.boundary_boxes()
    assert boxes == [] and inner == ((1, 1, 1), (9, 11, 13))

    class Fake(grid.GlobalGrid):
        def __init__(self, rank, dims, n):
            self.nx, self.ny, self.nz = n
            self.dist, self.group, self.nprocs, self.me, self.dims = None, None, int(np.prod(dims)), rank, dims
            self.coords = self.coords_of(rank)
            self.neighbors = {}
            for d in range(3):
                for side in (0, 1):
                    c = list(self.coords)
                    c[d] += 1 if side else -1
                    if 0 <= c[d] < dims[d]:
                        self.neighbors[2 * d + side] = tuple(c)

    dims = (2, 2, 2)
    seen = set()
    for r in range(8):
        f = Fake(r, dims, (10, 10, 10))
        assert f.rank_of(f.coords) == r
        seen.add(f.coords)
        assert len(f.neighbors) == 3
        for face, nb in f.neighbors.items():
            # neighbour relation is symmetric through the opposite face
            o = Fake(f.rank_of(nb), dims, (10, 10, 10))
            assert o.neighbors[face ^ 1] == f.coords
        boxes, inner = f.boundary_boxes()
        cells = sum((h[0] - l[0]) * (h[1] - l[1]) * (h[2] - l[2]) for l, h in boxes + [inner])
        assert cells == 8 ** 3 and len(boxes) == 3
        assert f.nx_g() == 18
    assert len(seen) == 8
    f = Fake(5, (1, 1, 8), (6, 6, 6))
    assert f.coords == (0, 0, 5) and sorted(f.neighbors) == [4, 5] and f.nz_g() == 34 and f.global_offset() == (0, 0, 20)


# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_pack(A, face, buf):
    d, side = face >> 1, face & 1
    n = A.shape[d]
    sl = A.select(d, n - 2 if side else 1)
    buf.copy_(sl.permute(1, 0).reshape(-1))  # column-major flatten of the plane


def _cpu_unpack(A, face, buf):
    d, side = face >> 1, face & 1
    n = A.shape[d]
    sl = A.select(d, n - 1 if side else 0)
    sl.copy_(buf.reshape(sl.shape[1], sl.shape[0]).permute(1, 0))


def _worker(rank, world, port, dims, n, iters, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle, farr

        orc = Oracle()
        nx, ny, nz = n
        gg = grid.GlobalGrid(nx, ny, nz, dims=dims)
        assert gg.nprocs == world and gg.me == rank
        lx, ly, lz = (d * 10.0 for d in dims)  # scale_physical_size=true
        dx, dy, dz = lx / gg.nx_g(), ly / gg.ny_g(), lz / gg.nz_g()
        D, dt = 1.0, 0.2
        dtau = min(dx, dy, dz) ** 2 / D / 8.1
        coef = (dtau, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
        Ht = orc.init_gaussian(n, dx, dy, dz, (lx / 2, ly / 2, lz / 2), gg.coords)
        A, B, R = Ht.copy(order="F"), farr(*n), farr(*n)
        ex = grid.HaloExchanger(n, gg.neighbors, gg.rank_of, _cpu_pack, _cpu_unpack,
                                lambda m: torch.zeros(m, dtype=torch.float64), dist=dist, group=None)
        norms = []
        for it in range(iters):
            orc.diffusion3d_step(Ht, A, B, R, *coef)
            tB = torch.from_numpy(B)
            assert tB.stride() == (1, nx, nx * ny)
            ex.update_halo_(tB)  # exchange the NEW buffer (deviation from the reference)
            A, B = B, A
            t = torch.tensor([orc.sumsq_scaled(R, dt)], dtype=torch.float64)
            norms.append(gg.allreduce_sum(t))
        np.save(os.path.join(outdir, "A_%d.npy" % rank), A)
        np.save(os.path.join(outdir, "R_%d.npy" % rank), R)
        if rank == 0:
            np.save(os.path.join(outdir, "norms.npy"), np.array(norms))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,n", [((2, 1, 1), (10, 9, 8)), ((1, 1, 2), (12, 7, 9)), ((1, 2, 1), (6, 11, 7)),
                                     ((2, 2, 1), (9, 8, 7)), ((2, 2, 2), (8, 8, 8)), ((1, 1, 4), (7, 6, 6))],
                         ids=lambda v: "x".join(map(str, v)))
def test_nshard_equals_single_domain(tmp_path, oracle, dims, n):
    world = int(np.prod(dims))
    iters = 7
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dims, n, iters, str(tmp_path)), nprocs=world, join=True)
    # single-domain oracle on the equivalent global grid
    from oracle.oracle import farr

    ng = tuple(d * (m - 2) + 2 for d, m in zip(dims, n))
    lx, ly, lz = (d * 10.0 for d in dims)
    dx, dy, dz = lx / ng[0], ly / ng[1], lz / ng[2]
    D, dt = 1.0, 0.2
    dtau = min(dx, dy, dz) ** 2 / D / 8.1
    coef = (dtau, 1 / dt, 1 / dx, 1 / dy, 1 / dz, D / dx, D / dy, D / dz)
    Ht = oracle.init_gaussian(ng, dx, dy, dz, (lx / 2, ly / 2, lz / 2))
    A, B, R = Ht.copy(order="F"), farr(*ng), farr(*ng)
    norms = []
    for it in range(iters):
        oracle.diffusion3d_step(Ht, A, B, R, *coef)
        A, B = B, A
        norms.append(oracle.sumsq_scaled(R, dt))
    got_norms = np.load(tmp_path / "norms.npy")
    assert np.allclose(got_norms, norms, rtol=1e-13, atol=0)
    for r in range(world):
        c = (r // (dims[1] * dims[2]), (r // dims[2]) % dims[1], r % dims[2])
        off = tuple(ci * (m - 2) for ci, m in zip(c, n))
        loc = np.load(tmp_path / ("A_%d.npy" % r))
        glob = A[off[0]:off[0] + n[0], off[1]:off[1] + n[1], off[2]:off[2] + n[2]]
        # interior cells and every exchanged halo plane (face interiors) agree bit for bit
        assert np.array_equal(loc[1:-1, 1:-1, 1:-1], glob[1:-1, 1:-1, 1:-1]), "rank %d interior" % r
        for d in range(3):
            for side in (0, 1):
                cc = list(c)
                cc[d] += 1 if side else -1
                if 0 <= cc[d] < dims[d]:
                    idx = [slice(1, -1)] * 3
                    idx[d] = -1 if side else 0
                    assert np.array_equal(loc[tuple(idx)], glob[tuple(idx)]), "rank %d halo face %d" % (r, 2 * d + side)
        locR = np.load(tmp_path / ("R_%d.npy" % r))
        globR = R[off[0]:off[0] + n[0], off[1]:off[1] + n[1], off[2]:off[2] + n[2]]
        assert np.array_equal(locR[1:-1, 1:-1, 1:-1], globR[1:-1, 1:-1, 1:-1])


REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "config")


def _last_line(stdout):
    """The driver keeps the tail of stdout and reads the LAST line: strict JSON (no NaN / Infinity), ASCII, below 4 KB."""
    import json

    last = stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < 4096, len(last)
    assert last.isascii()

    def no_constants(x):
        raise ValueError("non-finite constant %r in the bench line" % x)

    d = json.loads(last, parse_constant=no_constants)
    for k in REQUIRED_LINE_KEYS:
        assert k in d, k
    for k, v in d.items():             # flat: objects of scalars only, no prose
        for kk, vv in (v.items() if isinstance(v, dict) else ((k, v),)):
            assert not isinstance(vv, dict), (k, kk)
            assert not (isinstance(vv, str) and len(vv) > 160), (k, kk)
    return d


def test_bench_line_is_compact_for_a_full_record():
    """a 23 KB line left the driver's record empty.  The compact line built from a full record of the N = 1 run (the round-5
    record, committed as a fixture, and a two-rank one derived from it) stays below 4 KB and carries what the contract names: metric,
    value, roofline (physical frac, kernel time, bytes per launch, traffic), cpu_baseline, norm_check, one scalar per secondary leg."""
    import importlib.util
    import io
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("benchlegs_mod", os.path.join(root, "finalprojectrepo.jl_amd", "benchlegs.py"))
    bl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bl)
    full = json.load(open(os.path.join(root, "tests", "golden", "bench_full_record_r5.json")))
    assert len(json.dumps(full)) > 15000
    two = json.loads(json.dumps(full))
    two["n_gpus"] = 2
    two["config"].update({"process_grid": [1, 1, 2], "rccl_ranks": 2, "choreography": "pairs", "control_plane": "gloo"})
    two["first_attempt"] = {"attempt": 1, "choreography": "pairs", "reason": "no progress of rank 1 for 120 s in phase 'warmup' " + "x" * 500,
                            "rccl_rank0_log_tail": "y" * 2000}
    for k in ("cpu_baseline", "vcycle", "vcycle_5levels", "ns_step", "roofline_single", "power_probe"):
        two.pop(k)
    for rec_in in (full, two):
        buf = io.StringIO()
        line = bl.emit(rec_in, root=os.path.join(root, "no-such-dir"), stream=buf)
        d = _last_line(buf.getvalue())
        assert line == buf.getvalue().rstrip("\n")
        r = d["roofline"]
        for k in ("bound", "kernel", "kernel_ms", "bytes_per_launch", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, k
        assert r["bound"] == "hbm" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
        assert abs(r["achieved"] - r["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
        assert isinstance(r["bytes_per_launch"], int) and r["bytes_per_launch"] == 32 * 510 ** 3
        assert d["norm_check"]["ok"] is True
        assert d["config"]["local_grid"] == [512, 512, 512]
    d1 = json.loads(bl.compact_check(bl.compact_record(full)))
    assert set(d1["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d1["cpu_baseline"]["kind"] == "port"
    for k in ("vcycle_s", "vcycle_seam_us", "vcycle5_jacobi_s", "vcycle5_cg_s", "ns_step_s"):
        assert d1[k] > 0, k
    assert d1["roofline"]["unplaced_kernel_ms"] > 0
    d2 = json.loads(bl.compact_check(bl.compact_record(two)))
    assert d2["n_gpus"] == 2 and d2["config"]["rccl_ranks"] == 2 and len(d2["config"]["fallback_reason"]) <= 160
    # a record that cannot be made compact is an error, not a long line
    import pytest
    with pytest.raises(ValueError):
        bl.compact_check(dict(d1, junk="z" * 5000))


def test_bench_single_rank_dry_run_line(tmp_path):
    import sys
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FPR_BENCH_DETAIL_DIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_line(p.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dry_run"] is True
    assert (tmp_path / "bench_detail.json").exists()


def test_bench_self_launcher_dry_run_world2(tmp_path):
    """`python bench.py --gpus 2` without torchrun must start its two ranks itself (run_all_benchmarks.sh:21-28 uses
    mpiexecjl -np N): CPU rehearsal of the launcher and the gloo control plane, no GPU, no compute."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FPR_BENCH_DETAIL_DIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE JSON line, printed by rank 0 only
    d = _last_line(p.stdout)                  # ... the last line of stdout, compact
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["max_over_ranks"] == 2.0
    assert d["config"]["self_launched"] is True and d["config"]["control_plane"] == "gloo" and d["dry_run"] is True
    # a failing rank must fail the launch (exit code propagated, the other rank is not left hanging)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-such-flag"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def _bench(argv, timeout=300, detail_dir=None):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if detail_dir is not None:
        env["FPR_BENCH_DETAIL_DIR"] = str(detail_dir)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_watchdog_fails_a_stalled_attempt_and_falls_back_to_the_plain_choreography(tmp_path):
    """A rank that stops making progress (a deadlocked collective) must not hang the N > 1 bench: every rank process
    supervises its worker through a heartbeat file; the stalled attempt is failed for ALL ranks, fresh workers run once with
    --choreography plain, and the one JSON line says which choreography produced it and why the first attempt failed."""
    import json
    import time

    t0 = time.time()
    p = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--dry-run-hang", "1", "--watchdog-s", "6"], detail_dir=tmp_path)
    assert p.returncode == 0, p.stderr[-2000:]
    assert time.time() - t0 < 120
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = _last_line(p.stdout)
    assert d["config"]["attempt"] == 2 and d["config"]["choreography"] == "plain" and d["ranks_seen"] == 2
    assert "no progress" in d["config"]["fallback_reason"]
    fa = json.load(open(tmp_path / "bench_detail.json"))["first_attempt"]       # the whole story: the detail file
    assert fa["attempt"] == 1 and fa["choreography"] == "pairs" and "no progress" in fa["reason"]
    assert set(fa["phase_by_rank"]) == {"0", "1"}
    assert "watchdog" in p.stderr


def test_bench_watchdog_second_failure_exits_nonzero_within_bounds():
    """When the fallback stalls too the launch fails loudly (non-zero exit, an error line naming both attempts) instead of
    waiting for ever."""
    import json
    import time

    t0 = time.time()
    p = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--dry-run-hang", "0", "--dry-run-hang-always",
                "--watchdog-s", "6"])
    assert p.returncode != 0
    assert time.time() - t0 < 120
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert len(lines[0]) < 4096 and p.stdout.rstrip("\n").splitlines()[-1] == lines[0]
    assert d["value"] is None and d["error"] == "both attempts failed"
    assert "no progress" in d["config"]["first_attempt_reason"] and d["config"]["second_attempt_reason"]


def test_bench_watchdog_under_torch_distributed_run():
    """The driver starts N > 1 as `python -m torch.distributed.run ... bench.py --gpus N`: the ranks it starts supervise their
    workers the same way (the fallback's workers meet through a fresh rendezvous file, not the agent's store)."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run", "--dry-run-hang", "1", "--watchdog-s", "6"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = _last_line(p.stdout)
    assert d["config"]["attempt"] == 2 and d["config"]["choreography"] == "plain" and d["config"]["self_launched"] is False and d["ranks_seen"] == 2


def test_bench_norm_check_against_the_control_runs(tmp_path):
    """norm_check: the sum of squares a decomposed run prints against the single-rank control of the same global problem
    after the same number of iterations (tests/golden/scale_norms.json): 1e-12, and `ok: None` when there is no control."""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = tmp_path / "norms.json"
    g.write_text(json.dumps({"source": "test", "entries": {"n128_dims1,1,2": {"n": 128, "dims": [1, 1, 2], "global_grid": [128, 128, 254],
                                                                               "sumsq": [4.0, 3.0, 2.0]}}}))
    ok = bench.norm_check(128, (1, 1, 2), 2, 3.0 * (1 + 5e-13), str(g))
    assert ok["ok"] is True and ok["expected"] == 3.0 and ok["rel"] < 1e-12
    bad = bench.norm_check(128, (1, 1, 2), 3, 2.0 * (1 + 1e-9), str(g))
    assert bad["ok"] is False and bad["expected"] == 2.0
    assert bench.norm_check(128, (1, 1, 2), 4, 1.0, str(g))["ok"] is None        # beyond the recorded iterations
    assert bench.norm_check(128, (2, 2, 2), 1, 1.0, str(g))["ok"] is None        # no control for this process grid
    # the committed file: the process grids of run_all_benchmarks.sh:21-28 / part1_scaling_experiments.jl:35-41 and the z-slabs
    # bench.py runs by default, at the bench's local size, far enough for the driver's flags and the defaults
    committed = json.load(open(os.path.join(root, "tests", "golden", "scale_norms.json")))
    for dims in ("1,1,2", "1,1,4", "1,1,8", "2,1,1", "2,2,1", "2,2,2"):
        ent = committed["entries"]["n512_dims" + dims]
        assert len(ent["sumsq"]) >= 64 + 20 + 200 and all(v > 0 for v in ent["sumsq"])
        assert ent["global_grid"] == [int(d) * 510 + 2 for d in dims.split(",")]


def test_comm_share_and_box_order_of_fused_pairs():
    """GlobalGrid.reserve_cus: the comm stream of a fused pair gets 16 or 24 compute units without x-faces, else a multiple of 32 (the
    same number out of every shader engine), 64 only above two x-faces' worth of shell work; boundary_boxes peels the x-slabs last (step2 runs
    their first iteration on the core stream ahead of the core launch and takes them off the end of the list)."""
    class Fake(grid.GlobalGrid):
        def __init__(self, n, faces):
            self.nx, self.ny, self.nz = n
            self.neighbors = {f: (0, 0, 0) for f in faces}

    n = (64, 48, 40)
    for faces, want in (((4, 5), 16), ((4,), 16), ((2, 3), 16), ((2, 3, 4, 5), 24), ((1, 3, 5), 32), ((0, 1), 32),
                        ((0, 1, 2, 3), 64), ((0, 1, 2, 3, 4, 5), 64), ((0, 1, 4), 64)):
        g = Fake(n, faces)
        k = g.reserve_cus()
        assert k == want and k % 8 == 0, (faces, k)
        boxes, core = g.boundary_boxes()
        nxf = sum(1 for f in (0, 1) if f in faces)
        assert len(boxes) == len(faces)
        for lo, hi in boxes[len(boxes) - nxf:]:
            assert hi[0] - lo[0] == 1 and (lo[0] == 1 or hi[0] == n[0] - 1)
        for lo, hi in boxes[:len(boxes) - nxf]:
            assert hi[0] - lo[0] > 1
        # boxes and core are disjoint and cover the interior
        vol = sum((hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]) for lo, hi in boxes + [core])
        assert vol == (n[0] - 2) * (n[1] - 2) * (n[2] - 2)


def test_a_worker_does_not_outlive_a_killed_supervisor(tmp_path):
    """ADVICE r5 (launch.py): a supervisor that is SIGKILLed runs no handler, and its worker lives in a session of its own -- with a hung
    collective it would hold the GPU for ever.  launch._die_with_parent (prctl PR_SET_PDEATHSIG) in the worker, between fork and exec:
    a stand-in supervisor starts a sleeping child the way supervise() does, is killed with SIGKILL, and the child is gone."""
    import signal
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pidfile = tmp_path / "child.pid"
    code = ("import subprocess, sys, time\n"
            "sys.path.insert(0, %r)\n"
            "import importlib.util\n"
            "spec = importlib.util.spec_from_file_location('launch', %r)\n"
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
            "p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(120)'], start_new_session=True, preexec_fn=m._die_with_parent)\n"
            "open(%r, 'w').write(str(p.pid))\n"
            "time.sleep(120)\n") % (root, os.path.join(root, "finalprojectrepo.jl_amd", "launch.py"), str(pidfile))
    sup = subprocess.Popen([sys.executable, "-c", code])
    try:
        t0 = time.time()
        while not pidfile.exists() or not pidfile.read_text():
            assert time.time() - t0 < 60 and sup.poll() is None
            time.sleep(0.05)
        child = int(pidfile.read_text())
        os.kill(child, 0)                       # alive
        sup.send_signal(signal.SIGKILL)
        sup.wait(timeout=10)
        t0 = time.time()
        while time.time() - t0 < 10:
            try:
                os.kill(child, 0)
            except ProcessLookupError:
                break
            time.sleep(0.05)
        else:
            os.kill(child, signal.SIGKILL)
            raise AssertionError("the worker outlived its supervisor")
    finally:
        if sup.poll() is None:
            sup.kill()
