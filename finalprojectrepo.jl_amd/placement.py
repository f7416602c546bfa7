"""Field arrays placed for the streaming kernels (data layout in HBM, MI355X).

Which physical pages an allocation received decides how fast kernels run that stream several arrays at equal offsets (the fused
diffusion launch: 0.76 ms on four arrays whose placement labels all differ, 0.85-0.91 ms on arrays of one class).  The measurement
and the search live behind the C ABI (fpr_placement_rank, csrc/placement.hip: one implementation for every host language); this
module only does what the host owns -- allocate a pool of candidates (the first `count` plainly, as `@zeros` would; the rest behind
untouched spacers, because consecutive allocations share a label in runs of four or five GiB), call, extend the pool once if the
library says it is of one class, keep the chosen arrays, free the rest."""
import ctypes as C

_CHURNED = [0]          # churn() calls of this process so far (the mix it brings does not always outlast the next allocate / free cycle:
                        # a later pool of the process may churn again, MAX_CHURNS times in all)
MAX_CHURNS = 3


def churn(fraction=0.7, chunk_bytes=48 << 30):
    """Allocate, write and free most of the card's free memory once.  One lease in six or seven hands a fresh process only allocations
    of ONE placement class, whatever their number, spacing or distance (12 or 22 candidates over 56-188 GiB of addresses: every pair
    copies below 4950 GB/s, the fused launch takes 0.83-0.85 ms on any five of them); after this, the same pool recipe yields the usual
    mix (fastest pair 5116-5149 GB/s, 0.751-0.759 ms) -- tools/slow_state_probe.py, profiles/r5_one_class_lease_probe.txt.  About a
    second; nothing stays allocated."""
    import torch

    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    left, held = int(fraction * free), []
    try:
        while left >= (1 << 30):
            nb = min(left, chunk_bytes)
            t = torch.empty(nb, dtype=torch.uint8, device="cuda")
            t.zero_()
            held.append(t)
            left -= nb
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    del held
    torch.cuda.empty_cache()
    _CHURNED[0] = int(_CHURNED[0]) + 1

_TRIAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_int), C.c_int)
_R = dict(fastest=0, median=1, slowest=2, chosen_slowest=3, chosen_mean=4, trials=5, best=6, first=7, worst=8, identity=9, spread=10,
          want_more=11)          # include/fpr.h FPR_PLACE_*


def alloc_fields(count, *shape, pool=None, min_bytes=256 << 20, report=None, pairs=None, trial=None, trials=4, spacer_bytes=None,
                 extend_by=10, first=None, extend_below_GBs=None, accept=None):
    """`count` zeroed column-major float64 arrays of `shape`, the best-matched of `pool` candidates (default count + 7).  `pairs`:
    positions streamed together; `trial(arrays) -> ms`: the caller's own kernel as the judge; `first`: arrays the caller already
    holds, used as the first candidates (the plain allocation a search must beat); `extend_below_GBs`: the library's option
    place_extend_below_GBs for this call.  `accept(arrays) -> bool`: the caller's own check of the chosen arrays (e.g. the fused launch
    against the one-iteration kernel on the same arrays): a pool with one candidate of another class among eleven alike has a fast pair and
    still no good assignment (0.788-0.795 ms where a mixed pool gives 0.745-0.751) -- when it says no, the pool is rebuilt once behind
    churn() with the chosen arrays as its first candidates, so the second search cannot end below the first.  `report` receives what
    was measured."""
    import torch

    from . import ctx as _ctx
    from ._lib import fzeros

    nbytes = 8
    for s in shape:
        nbytes *= int(s)
    report = {} if report is None else report
    report.update({"arrays": count, "bytes_per_array": nbytes, "selected": False})
    cands = list(first or [])
    free, _ = torch.cuda.mem_get_info()
    k = max(count, min(pool if pool is not None else count + 7, len(cands) + int(0.6 * free // nbytes)))
    if nbytes < min_bytes or count < 2 or k <= count:
        return (cands + [fzeros(*shape) for _ in range(count - len(cands))])[:count]
    spacer = int(spacer_bytes if spacer_bytes is not None else max(4 << 30, 3 * nbytes))
    if spacer and k * nbytes + (k - 1) * spacer > 0.7 * free:
        spacer = max(0, int((0.7 * free - k * nbytes) // max(k - 1, 1)))
    spacers = []

    def grow(upto):
        nonlocal spacer
        while len(cands) < upto:
            if len(cands) >= count and spacer >= (64 << 20):      # the first `count` lie as a plain allocation would
                try:
                    spacers.append(torch.empty(spacer, dtype=torch.uint8, device="cuda"))
                except RuntimeError:
                    spacer = 0
            try:
                cands.append(fzeros(*shape))
            except RuntimeError:
                break

    c = _ctx()
    c.set_option("place_trials", int(trials))
    c.set_option("place_extend_below_GBs", 5050 if extend_below_GBs is None else int(min(extend_below_GBs, 1e15)))
    flat = [int(i) for p in (pairs or []) for i in p]
    rep = (C.c_double * 16)()
    chosen = (C.c_int * count)()

    @_TRIAL_FN
    def cb(_user, idx, n):
        try:
            return float(trial([cands[idx[i]] for i in range(n)]))
        except Exception:
            return 0.0           # cannot judge this assignment

    def rank():
        ptrs = (C.c_void_p * len(cands))(*[a.data_ptr() for a in cands])
        torch.cuda.synchronize()
        c.call("fpr_placement_rank", ptrs, len(cands), cands[0].numel(), count, (C.c_int * max(len(flat), 1))(*flat), len(flat) // 2,
               C.cast(cb, C.c_void_p) if trial is not None else None, None, chosen, rep)

    def slab_pool(why, before):
        """Candidates carved out of ONE allocation at a pitch of six arrays (at least 6 GiB): the other way past a card whose piecewise
        allocations are all of one class (three one-class leases: 0.84 -> 0.746-0.752 ms, profiles/r5_one_class_lease_probe.txt) -- taken when
        a churn has not brought the mix back (one lease in six of that kind).  The allocation stays as long as any chosen array lives."""
        nfirst = len(first or [])
        pitch = max(6 << 30, 6 * nbytes)
        pitch += (-pitch) % (2 << 20)
        free3, _ = torch.cuda.mem_get_info()
        ns = min(k, int(0.6 * free3 // pitch))
        if ns < count:
            return False
        del cands[nfirst:]
        spacers.clear()
        torch.cuda.empty_cache()
        try:
            slab = torch.empty(ns * pitch, dtype=torch.uint8, device="cuda")
        except RuntimeError:
            grow(k)
            return False
        numel = nbytes // 8
        strides, st = [], 1
        for d_ in shape:
            strides.append(st)
            st *= int(d_)
        for i in range(ns):
            t = slab[i * pitch:i * pitch + nbytes].view(torch.float64)
            t.zero_()
            cands.append(torch.as_strided(t, tuple(int(d_) for d_ in shape), tuple(strides)))
        del slab
        report["slab_because_" + why] = before
        report["slab_bytes"] = ns * pitch
        return True

    grow(k)
    report["pool_first"] = len(cands)
    rank()
    import os
    if os.environ.get("FPR_PLACE_FORCE_SLAB") and trial is not None:      # (measurement hook: the carved pool on a lease that does not need it)
        report["forced_slab_first_pool_trial_ms_best"] = rep[_R["best"]]
        if slab_pool("forced", rep[_R["fastest"]]):
            rank()
    if extend_by > 0 and rep[_R["want_more"]] == 1 and int(_CHURNED[0]) < MAX_CHURNS:
        # a pool of one class: more candidates of the same process do not help (22 over 106 GiB were tried); churning the card's memory once does
        # (only the case measured: no fast pair at all; a pool whose trials merely agree -- want_more 2 -- is extended as before)
        report["churned_because_fastest_pair_GBs"] = rep[_R["fastest"]]
        nfirst = len(first or [])
        del cands[nfirst:]
        spacers.clear()
        churn()
        grow(k)
        rank()
        if rep[_R["want_more"]] == 1 and slab_pool("fastest_pair_GBs_after_churn", rep[_R["fastest"]]):
            rank()
    if extend_by > 0 and rep[_R["want_more"]] > 0 and "slab_bytes" not in report:
        why, before = ("fastest_pair_GBs", rep[_R["fastest"]]) if rep[_R["want_more"]] == 1 else ("trial_spread", rep[_R["spread"]])
        n0 = len(cands)
        free2, _ = torch.cuda.mem_get_info()
        grow(n0 + min(extend_by, int(0.7 * free2 // (nbytes + max(spacer, 0)))))
        if len(cands) > n0:
            report["pool_extended_because_" + why] = before
            rank()
    out = [cands[chosen[i]] for i in range(count)]
    if accept is not None and trial is not None and int(_CHURNED[0]) < MAX_CHURNS and not any(k.startswith("churned_because") for k in report):
        try:
            ok = bool(accept(out))
        except Exception:
            ok = True
        if not ok:
            report["churned_because_not_accepted_ms"] = rep[_R["best"]]
            report["pool_before_churn"] = {"fastest": rep[_R["fastest"]], "median": rep[_R["median"]], "trial_ms_best": rep[_R["best"]]}
            keep = list(out)
            del cands[:]
            spacers.clear()
            cands.extend(keep)          # the chosen arrays stay: candidates 0 .. count-1 of the second pool (its first trial)
            del keep, out
            churn()
            grow(k)
            rank()
            out = [cands[chosen[i]] for i in range(count)]
            try:
                report["accepted_after_churn"] = bool(accept(out))
            except Exception:
                report["accepted_after_churn"] = True
            if not report["accepted_after_churn"]:
                keep = list(out)
                first = keep                       # (slab_pool keeps the first len(first) candidates)
                del cands[:]
                cands.extend(keep)
                del keep, out
                if slab_pool("not_accepted_after_churn_ms", rep[_R["best"]]):
                    rank()
                else:
                    grow(k)
                    rank()
                out = [cands[chosen[i]] for i in range(count)]
                try:
                    report["accepted_after_slab"] = bool(accept(out))
                except Exception:
                    pass
    report.update({"selected": True, "pool": len(cands), "chosen": [int(chosen[i]) for i in range(count)], "pairs": [list(p) for p in (pairs or [])],
                   "spacer_bytes": spacer if spacers else 0, "trials": int(rep[_R["trials"]]),
                   "pair_copy_GBs_all": {"slowest": rep[_R["slowest"]], "median": rep[_R["median"]], "fastest": rep[_R["fastest"]]},
                   "pair_copy_GBs_chosen": {"slowest": rep[_R["chosen_slowest"]], "mean": rep[_R["chosen_mean"]]},
                   "note": "fpr_placement_rank (csrc/placement.hip): candidates timed pairwise, then the caller's kernel decides; the "
                           "candidates as given (a plain allocation) are always among the trials"})
    if trial is not None:
        report.update({"trial_ms_best": rep[_R["best"]], "trial_ms_first": rep[_R["first"]], "trial_ms_worst": rep[_R["worst"]],
                       "trial_ms_plain_allocation": rep[_R["identity"]]})
    for a in out:
        a.zero_()
    del cands, spacers
    torch.cuda.empty_cache()
    return out
