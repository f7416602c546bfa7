#!/usr/bin/env python3
"""N ranks as THREADS of one process (grid.ThreadWorld: one library context per thread, planes over fpr_comm_init_hosted and in-process
queues) against the single-domain run of the same global problem: every rank's field and residual bit for bit, the all-reduced sums of
squares to 1e-12.  One process per configuration (tests/test_gpu_rccl.py::test_eight_ranks_as_threads... starts it): a GPU box admits six
processes on its card, so eight REAL processes cannot run there; the rank code that runs here is the library's own exchange code and
one-call pair / triple choreography.   usage: thread_ranks_worker.py dx,dy,dz pairs|triples|plain [n] [iters]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
import torch

F = fpr_amd.load(0)
dims = tuple(int(x) for x in sys.argv[1].split(","))
mode = sys.argv[2] if len(sys.argv) > 2 else "pairs"
fused = {"pairs": 2, "triples": 3}.get(mode, 0)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 128
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
world = dims[0] * dims[1] * dims[2]
nglob = tuple(d * (n - 2) + 2 for d in dims)
lx, ly, lz = (d * 10.0 for d in dims)
dx, dy, dz = lx / nglob[0], ly / nglob[1], lz / nglob[2]
D, dt = 1.0, 0.2
coef = (min(dx, dy, dz) ** 2 / D / 8.1, 1.0 / dt, 1.0 / dx, 1.0 / dy, 1.0 / dz, D / dx, D / dy, D / dz)


def run(gg, nloc, fused, local_only=False):
    Ht = F.fzeros(*nloc)
    F.part1.init_local_gaussian((lx / 2, ly / 2, lz / 2), dx, dy, dz, Ht, gg.coords)
    A, B, C3, R = Ht.clone(memory_format=torch.preserve_format), F.fzeros(*nloc), Ht.clone(memory_format=torch.preserve_format), F.fzeros(*nloc)
    sq = torch.zeros(iters, dtype=torch.float64, device=Ht.device)
    if fused == 3:      # Hτ and Hτ2 alternate (fpr_diffusion3d_step3_halo between ranks)
        assert gg.can_step3(Ht, A, B, R) and iters % 3 == 0
        for i in range(0, iters, 3):
            gg.step3(Ht, A, B, R, *coef, dt, sq[i:i + 3], join=False)
            A, B = B, A
        gg.join()
    elif fused:
        assert gg.can_step2(Ht, A, B, C3, R)
        for i in range(0, iters, 2):
            gg.step2(Ht, A, B, C3, R, *coef, dt, sq[i:i + 2], join=False)
            A, C3 = C3, A
        gg.join()
    else:
        for i in range(iters):
            gg.step(Ht, A, B, R, *coef, dt, sq[i:i + 1])
            A, B = B, A
    F.ctx().synchronize()
    loc = sq.cpu().numpy().copy()
    gg.allreduce_(sq)
    F.ctx().synchronize()
    return F.tonumpy(A), F.tonumpy(R), sq.cpu().numpy(), loc


if mode == "solver":
    # the decomposed LOOP of part1.diffusion_3D_kernel_programming with triples between ranks (norms all-reduced, an iteration that ends the
    # loop inside a triple replayed singly) against the same loop with one iteration per launch (options diff3_fuse3 = diff3_fuse2 = 0):
    # iteration counts and errors per physical step, every rank's field bit for bit.  (The single-domain run is no control for the COUNTS:
    # the reference's total_N counts the halo cells of every rank, :124, so its norm is scaled differently.)
    kw = dict(ttot=0.4, tol=float(sys.argv[5]) if len(sys.argv) > 5 else 2e-4, verbose=False)

    def solve(plain):
        res_s, err_s = [None] * world, []
        tw = F.grid.ThreadWorld(world)

        def solver_rank(r):
            try:
                c = F.Context(0, secondary=True)
                F.bind_context(c)
                if plain:
                    c.set_option("diff3_fuse3", 0)
                    c.set_option("diff3_fuse2", 0)
                gg = F.grid.GlobalGrid(n, n, n, dims=dims, transport="hosted", dist=tw.rank_view(r))
                _, H, _, info = F.part1.diffusion_3D_kernel_programming(nx=n, ny=n, nz=n, global_grid=gg, **kw)
                res_s[r] = (gg.coords, H, list(info["iters"]), list(info["err"]), F.ctx().L.fpr_comm_cus(F.ctx().h))
                gg.barrier()
                F.grid.finalize_global_grid()
            except BaseException:
                import traceback
                err_s.append((r, traceback.format_exc()))
                try:
                    tw.bar.abort()
                except Exception:
                    pass
            finally:
                F.bind_context(None)

        ths = [threading.Thread(target=solver_rank, args=(r,), daemon=True) for r in range(world)]
        [t.start() for t in ths]
        [t.join(timeout=300) for t in ths]
        for r, e in err_s:
            print("rank", r, e)
        return res_s, not err_s and all(x is not None for x in res_s)

    fused_res, ok1 = solve(False)
    plain_res, ok2 = solve(True)
    ok = ok1 and ok2
    for r in range(world):
        if fused_res[r] is None or plain_res[r] is None:
            continue
        coords, H, its, errs, cus = fused_res[r]
        _, Hp, itsp, errsp, _ = plain_res[r]
        same = np.array_equal(H, Hp)
        print("rank %d coords %s: iterations %s (one per launch: %s) field %s comm units %d" % (r, coords, its, itsp, same, cus))
        ok = ok and same and its == itsp and np.allclose(errs, errsp, rtol=1e-10, atol=0)
    print("dims %s solver n %d: %s" % (dims, n, "OK" if ok else "MISMATCH"))
    sys.exit(0 if ok else 1)

g1 = F.grid.GlobalGrid(*nglob, dims=(1, 1, 1), transport=None, use_dist=False)
A1, R1, sq1, _ = run(g1, nglob, 2)
tw = F.grid.ThreadWorld(world)
results, errors = [None] * world, []


def rank_main(r):
    c = None
    try:
        c = F.Context(0, secondary=True)
        F.bind_context(c)
        gg = F.grid.GlobalGrid(n, n, n, dims=dims, transport="hosted", dist=tw.rank_view(r))
        results[r] = (gg.coords, run(gg, (n, n, n), fused))
        gg.barrier()
        F.grid.finalize_global_grid()
    except BaseException as e:
        import traceback
        errors.append((r, traceback.format_exc()))
        try:
            tw.bar.abort()
        except Exception:
            pass
    finally:
        F.bind_context(None)


ths = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
[t.start() for t in ths]
[t.join(timeout=300) for t in ths]
for r, e in errors:
    print("rank", r, e)
ok = not errors
inner = (slice(1, -1),) * 3
locsum = np.zeros(iters)
for r, res in enumerate(results):
    if res is None:
        continue
    coords, (A, R, sq, loc) = res
    off = tuple(ci * (n - 2) for ci in coords)
    sl = tuple(slice(o, o + n) for o in off)
    fa, fr = np.array_equal(A[inner], A1[sl][inner]), np.array_equal(R[inner], R1[sl][inner])
    # what this rank's cells contribute to the single-domain sums (last iteration only: the residual array holds that one)
    exp_last = float(((R1[sl][inner] * dt) ** 2).sum())
    print("rank %d coords %s: field %s residual %s; local sumsq last %.12g (from the single-domain residual %.12g); reduced ok %s" %
          (r, coords, fa, fr, loc[-1], exp_last, np.allclose(sq, sq1, rtol=1e-12, atol=0)))
    locsum += loc
    ok = ok and fa and fr and np.allclose(sq, sq1, rtol=1e-12, atol=0)
print("sum of local sums:", locsum)
print("single domain    :", sq1)
print("reduced (rank 0) :", results[0][1][2] if results[0] else None)
print("dims %s %s n %d: %s" % (dims, mode if fused else "plain", n, "OK" if ok else "MISMATCH"))
sys.exit(0 if ok else 1)
