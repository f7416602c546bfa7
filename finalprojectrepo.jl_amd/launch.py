"""One process per GPU and a watchdog around every rank: the launcher of `bench.py --gpus N` (role of `mpiexecjl -np N` in
run_all_benchmarks.sh:21-28) and the supervisor every rank process runs around its worker.

Nothing in this module imports torch or touches a GPU: it runs BEFORE anything initialises HIP (bench.py loads it by file path,
without importing the package).  `script` / `argv` are the program the ranks run (bench.py and its own arguments)."""
import json
import os
import re
import shutil
import signal
import socket
import subprocess
import sys
import time

# ------------------------------------------------------------------------------------------------------------
# launcher: one process per GPU (role of `mpiexecjl -np N` in run_all_benchmarks.sh:21-28)
# ------------------------------------------------------------------------------------------------------------
def self_launch(n, script, argv):
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), wait for them and
    exit with the first non-zero exit code.  Runs before torch or HIP is imported: nothing here touches a GPU.  Every rank
    it starts supervises its own worker (supervise() below), exactly as a rank started by torch.distributed.run does."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FPR_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL between processes)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    rc = 0

    def stop_all(grace=8.0):
        """SIGTERM to every rank (a supervisor: it ends its worker's process group and leaves), SIGKILL after a grace period."""
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t0 = time.time()
        while any(q.poll() is None for q in procs) and time.time() - t0 < grace:
            time.sleep(0.05)
        for q in procs:
            if q.poll() is None:
                q.kill()

    def on_signal(signum, frame):
        stop_all()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                r = p.poll()
                if r is None:
                    continue
                alive.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    stop_all()           # a failed rank would leave the others waiting in a collective
            time.sleep(0.05)
    finally:
        stop_all(grace=3.0)
    sys.exit(rc)


# ------------------------------------------------------------------------------------------------------------
# watchdog: between ranks every step is a collective pattern, and a collective that deadlocks waits for ever.  Each rank
# process (started by torch.distributed.run or by self_launch) therefore does NOT touch a GPU itself: it starts its worker
# as a child and watches the worker's heartbeat file.  No progress for --watchdog-s seconds, or a worker that dies,
# fails the ATTEMPT for every rank (a marker file in the directory all ranks of the job share); the supervisors then
# start FRESH workers once with --choreography plain (single steps: boundary slabs -> exchange || interior, no split
# of the device, no chained pairs).  A second failure exits non-zero on every rank.
# ------------------------------------------------------------------------------------------------------------
HB_PHASES_QUIET = ("start",)       # phases that may be silent for --watchdog-import-s (the first `import torch` on a fresh box pages the image in)


def job_dir():
    """Directory shared by the ranks of ONE job on this node: keyed by the parent process (the torchrun agent or
    self_launch, the same for every rank) and the rendezvous port.  A job that spans nodes must name a directory all its nodes
    share in FPR_BENCH_JOB_DIR (the failure marker that starts the fallback on every rank lives there)."""
    if os.environ.get("FPR_BENCH_JOB_DIR"):
        return os.environ["FPR_BENCH_JOB_DIR"]
    key = "%s_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "fpr_bench_job_" + re.sub(r"[^A-Za-z0-9_.-]", "_", key))


def hb(phase, **extra):
    """Worker side: record progress (phase name + time) for the supervisor.  No-op without a supervisor."""
    path = os.environ.get("FPR_BENCH_HB_FILE")
    if not path:
        return
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(dict(phase=phase, t=time.time(), **extra), f)
    os.replace(tmp, path)


def _read_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _kill_tree(p, grace=3.0):
    """End one worker (exact pid; its own process group so that helpers it started go with it)."""
    if p.poll() is not None:
        return
    try:
        os.killpg(p.pid, signal.SIGTERM)
    except OSError:
        pass
    t0 = time.time()
    while p.poll() is None and time.time() - t0 < grace:
        time.sleep(0.05)
    if p.poll() is None:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        p.wait()


def _die_with_parent():
    """In the worker, between fork and exec: SIGKILL when the supervisor dies (a SIGKILLed supervisor runs no handler; its worker sits in a
    session of its own and would keep the GPU).  prctl(PR_SET_PDEATHSIG = 1, SIGKILL); Linux only, best effort."""
    try:
        import ctypes

        ctypes.CDLL(None, use_errno=True).prctl(1, 9, 0, 0, 0)
    except Exception:
        pass


def supervise(args, script, argv):
    """Rank process of an N > 1 run: start the worker, watch it, fall back once.  Never returns."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    jd = job_dir()
    os.makedirs(jd, exist_ok=True)
    first_failure = None
    rc = 1
    current = [None]                     # the worker this supervisor is responsible for

    def on_signal(signum, frame):
        # torchrun / self_launch / Ctrl-C end the rank process: its worker lives in a session of its own and would be orphaned with the
        # GPU in its hands -- end exactly that process group, then leave
        if current[0] is not None:
            _kill_tree(current[0])
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    last_phase = "start"
    for attempt in (1, 2):
        choreo = args.choreography if attempt == 1 else "plain"
        hbf = os.path.join(jd, "hb_%d_%d.json" % (attempt, rank))
        fail_marker = os.path.join(jd, "fail_%d" % attempt)
        env = dict(os.environ, FPR_BENCH_WORKER="1", FPR_BENCH_ATTEMPT=str(attempt), FPR_BENCH_HB_FILE=hbf, FPR_BENCH_CHOREOGRAPHY=choreo)
        # control-plane rendezvous of the workers: a file in the job directory when every rank of the job runs on this node
        # (self-launched, or torchrun with LOCAL_WORLD_SIZE == WORLD_SIZE) or the job directory is declared shared; between nodes
        # env:// as torch.distributed.run set it up (a fresh port for the fallback attempt: rank 0's worker hosts that store)
        one_node = (os.environ.get("FPR_BENCH_SELF_LAUNCHED") == "1" or bool(os.environ.get("FPR_BENCH_JOB_DIR"))
                    or int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) == world)
        if one_node:
            env["FPR_BENCH_RDZV_FILE"] = os.path.join(jd, "rdzv_%d" % attempt)
        elif attempt > 1:
            env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17 * attempt)
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")
        env.setdefault("NCCL_DEBUG_FILE", os.path.join(jd, "rccl_%d_%d.log" % (attempt, rank)))   # read back on failure
        if first_failure is not None:
            env["FPR_BENCH_FIRST_FAILURE"] = json.dumps(first_failure)
        p = subprocess.Popen([sys.executable, script] + list(argv), env=env, start_new_session=True, preexec_fn=_die_with_parent)
        current[0] = p
        t_start = time.time()
        reason, detail = None, None
        last_phase, last_t = "start", t_start
        while True:
            r = p.poll()
            h = _read_json(hbf)
            if h is None:                # (not written yet, or the job directory is already gone: keep what was seen last)
                h = {"phase": last_phase, "t": last_t}
            last_phase, last_t = h.get("phase", last_phase), h.get("t", last_t)
            if r is not None:
                if r == 0 or h.get("phase") == "done":
                    rc = 0           # the job's result is out (rank 0 prints after the last collective); teardown noise is not a failure
                else:
                    reason = "worker of rank %d exited with code %d in phase %r" % (rank, r, h.get("phase"))
                    detail = {k: v for k, v in h.items() if k not in ("phase", "t")} or None
                break
            if os.path.exists(fail_marker) and h.get("phase") != "done":
                reason = (_read_json(fail_marker) or {}).get("reason", "another rank failed the attempt")
                break
            quiet = time.time() - max(h.get("t", t_start), t_start)
            limit = args.watchdog_import_s if h.get("phase") in HB_PHASES_QUIET else args.watchdog_s
            if h.get("phase") == "done":
                if quiet > 30.0:     # result printed, a rank hangs in teardown: end it
                    _kill_tree(p)
                    rc = 0
                    break
            elif quiet > limit:
                reason = "no progress of rank %d for %.0f s in phase %r" % (rank, quiet, h.get("phase"))
                break
            time.sleep(0.1)
        if reason is None:
            break
        # the attempt failed: tell every rank (first writer wins), end the worker, collect what RCCL said
        try:
            fd = os.open(fail_marker, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
            os.write(fd, json.dumps({"reason": reason, "detail": detail, "rank": rank, "t": time.time()}).encode())
            os.close(fd)
        except OSError:
            fm = _read_json(fail_marker) or {}
            reason, detail = fm.get("reason", reason), fm.get("detail", detail)
        _kill_tree(p)
        log = ""
        try:
            with open(os.path.join(jd, "rccl_%d_%d.log" % (attempt, 0))) as f:
                log = f.read()[-2000:]
        except OSError:
            pass
        phases = {}
        for r_ in range(world):
            hh = _read_json(os.path.join(jd, "hb_%d_%d.json" % (attempt, r_)))
            phases[str(r_)] = hh.get("phase") if hh else None
        failure = {"attempt": attempt, "choreography": choreo, "reason": reason, "detail": detail, "phase_by_rank": phases,
                   "rccl_rank0_log_tail": log}
        print("bench.py watchdog (rank %d): attempt %d (%s) failed: %s" % (rank, attempt, choreo, reason), file=sys.stderr)
        if attempt == 1:
            first_failure = failure
            time.sleep(1.0)          # every supervisor has seen the marker and ended its worker before fresh ones meet
            continue
        # rank 0 reports; the others leave only once it has (the launcher ends every rank as soon as one exits non-zero)
        reported = os.path.join(jd, "reported")
        if rank == 0:
            if phases.get("0") != "norm_failed":     # (a fallback whose norm is wrong has printed its own line, norm_check.ok = false)
                # the whole story (phases by rank, RCCL log tails) goes to the detail file; the line stays short (< 4 KB)
                try:
                    with open(os.path.join(os.path.dirname(os.path.abspath(script)), "bench_detail.json"), "w") as f:
                        json.dump({"error": "both attempts failed", "attempts": [first_failure, failure]}, f, indent=1)
                except OSError:
                    pass
                short = lambda x: (str(x)[:157] + "...") if len(str(x)) > 160 else str(x)
                print(json.dumps({"metric": "diffusion3d_effective_memory_throughput", "value": None, "unit": "GB/s", "n_gpus": world,
                                  "steps": args.steps, "warmup": args.warmup, "error": "both attempts failed",
                                  "config": {"workload": "3D diffusion, N ranks", "first_attempt_reason": short(first_failure.get("reason")),
                                             "second_attempt_reason": short(failure.get("reason"))},
                                  "detail": "bench_detail.json"}))
                sys.stdout.flush()
            try:
                open(reported, "w").close()
            except OSError:
                pass
            time.sleep(1.0)          # (the others are on their way out; the job directory goes last)
        else:
            t_wait = time.time()
            while not os.path.exists(reported) and time.time() - t_wait < 20.0:
                time.sleep(0.05)
        rc = 1
    current[0] = None
    if rank == 0:
        # the job directory goes last: the other supervisors still read their heartbeat files from it (bounded wait for every rank's
        # worker to have reached 'done' / the job to have been reported)
        t_wait = time.time()
        while time.time() - t_wait < 15.0:
            seen = [(_read_json(os.path.join(jd, "hb_%d_%d.json" % (attempt, r_))) or {}).get("phase") for r_ in range(world)]
            if all(ph in ("done", "norm_failed") for ph in seen) or os.path.exists(os.path.join(jd, "reported")):
                break
            time.sleep(0.1)
        time.sleep(0.5)
        shutil.rmtree(jd, ignore_errors=True)
    sys.exit(rc)
