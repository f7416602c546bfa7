"""Field arrays placed for kernels that stream several arrays at equal offsets (data layout in HBM, MI355X).

Which physical pages an allocation received decides how fast such kernels run: the two-iteration diffusion launch takes 0.76 ms at
512^3 on four arrays whose placement labels all differ and 0.85-0.91 ms on arrays of one class; the seam pass of the V-cycle 99 against
113 us.  The three-iteration launch of a single rank does not care (0.916 ms placed, 0.913-0.916 plain on the same lease:
profiles/r6_placement_on_off.txt), so bench.py's N = 1 headline allocates plainly; ranks with neighbours (fused pairs) and the V-cycle
block use this module.  The measurement and the search live behind the C ABI (fpr_placement_rank, csrc/placement.hip); this module
does what the host owns: allocate a pool of candidates -- the first `count` plainly, as `@zeros` would, the rest behind untouched
spacers (consecutive allocations share a label in runs of four or five GiB) -- call, keep the chosen arrays, free the rest.  Nothing
but the returned arrays stays allocated.  (Round 5's answers to a lease whose piecewise allocations are all of one class -- churning the
card's memory, a pool carved out of one 72 GiB allocation -- are gone with the kernel that needed them: EXPERIMENTS 14.)"""
import ctypes as C

_TRIAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_int), C.c_int)
_R = dict(fastest=0, median=1, slowest=2, chosen_slowest=3, chosen_mean=4, trials=5, best=6, first=7, worst=8, identity=9, spread=10,
          want_more=11)          # include/fpr.h FPR_PLACE_*


def alloc_fields(count, *shape, pool=None, min_bytes=256 << 20, report=None, pairs=None, trial=None, spacer_bytes=None):
    """`count` zeroed column-major float64 arrays of `shape`, the best-matched of `pool` candidates (default count + 7).  `pairs`:
    positions streamed together; `trial(arrays) -> ms`: the caller's own kernel as the judge (the candidates as allocated -- what a host
    that simply allocates gets -- are always among the trials, so the result is never worse than that).  The candidates' contents are
    scratch during the search (the pair copies overwrite them); the returned arrays are zeroed.  `report` receives what was measured."""
    import torch

    from . import ctx as _ctx
    from ._lib import fzeros

    nbytes = 8
    for s in shape:
        nbytes *= int(s)
    report = {} if report is None else report
    report.update({"arrays": count, "bytes_per_array": nbytes, "selected": False})
    free, _ = torch.cuda.mem_get_info()
    k = max(count, min(pool if pool is not None else count + 7, int(0.6 * free // nbytes)))
    if nbytes < min_bytes or count < 2 or k <= count:
        return [fzeros(*shape) for _ in range(count)]
    spacer = int(spacer_bytes if spacer_bytes is not None else max(4 << 30, 3 * nbytes))
    if spacer and k * nbytes + (k - 1) * spacer > 0.7 * free:
        spacer = max(0, int((0.7 * free - k * nbytes) // max(k - 1, 1)))
    cands, spacers = [], []
    while len(cands) < k:
        try:
            if len(cands) >= count and spacer >= (64 << 20):      # the first `count` lie as a plain allocation would
                spacers.append(torch.empty(spacer, dtype=torch.uint8, device="cuda"))
            cands.append(fzeros(*shape))
        except RuntimeError:
            break
    if len(cands) <= count:
        return (cands + [fzeros(*shape) for _ in range(count - len(cands))])[:count]
    c = _ctx()
    flat = [int(i) for p in (pairs or []) for i in p]
    rep = (C.c_double * 16)()
    chosen = (C.c_int * count)()

    @_TRIAL_FN
    def cb(_user, idx, n):
        try:
            return float(trial([cands[idx[i]] for i in range(n)]))
        except Exception:
            return 0.0           # cannot judge this assignment

    ptrs = (C.c_void_p * len(cands))(*[a.data_ptr() for a in cands])
    torch.cuda.synchronize()
    c.call("fpr_placement_rank", ptrs, len(cands), cands[0].numel(), count, (C.c_int * max(len(flat), 1))(*flat), len(flat) // 2,
           C.cast(cb, C.c_void_p) if trial is not None else None, None, chosen, rep)
    out = [cands[chosen[i]] for i in range(count)]
    report.update({"selected": True, "pool": len(cands), "chosen": [int(chosen[i]) for i in range(count)], "pairs": [list(p) for p in (pairs or [])],
                   "spacer_bytes": spacer if spacers else 0, "trials": int(rep[_R["trials"]]),
                   "pair_copy_GBs_all": {"slowest": rep[_R["slowest"]], "median": rep[_R["median"]], "fastest": rep[_R["fastest"]]},
                   "pair_copy_GBs_chosen": {"slowest": rep[_R["chosen_slowest"]], "mean": rep[_R["chosen_mean"]]},
                   "one_class_pool": rep[_R["want_more"]] == 1, "search_truncated": rep[12] < 0})   # (12 = FPR_PLACE_SEARCH_NODES)
    if trial is not None:
        report.update({"trial_ms_best": rep[_R["best"]], "trial_ms_first": rep[_R["first"]], "trial_ms_worst": rep[_R["worst"]],
                       "trial_ms_plain_allocation": rep[_R["identity"]]})
    for a in out:
        a.zero_()
    del cands, spacers
    torch.cuda.empty_cache()
    return out
