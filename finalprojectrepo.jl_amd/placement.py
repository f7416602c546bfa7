"""Field arrays placed for the streaming kernels (data layout in HBM, MI355X).

Measured on MI355X (tools/place_probe.hip, tools/diffusion_tune f2class / f2place; profiles/r4_placement_*.txt): separately
allocated 1 GiB arrays fall into a few classes.  A streaming pass that reads one array and writes another at the same
offsets runs at ~4950 GB/s when the two are of one class and at ~5150 / ~5400 GB/s otherwise (every array alone reads at
6.5 TB/s; address-translation counters are the same) -- and the fused diffusion launch, which streams Ht and Hτ in and the new
field and the residual out at equal offsets, takes 0.775 ms on four arrays that differ and 0.85-0.91 ms on arrays of one
class.  The class comes with the physical pages an allocation happens to receive (the same virtual address is fast in one
process and slow in the next), so it cannot be computed, only measured: `alloc_fields` allocates a small pool of candidates,
times a copy between every pair, keeps the subset whose pairs copy fastest and frees the rest.  Nothing about the arrays'
contents or layout changes; a host that owns its arrays (the reference's `@zeros`) can do the same once at start-up.
"""
import itertools


def _pair_times(c, arrs, reps=2):
    """Copy time i -> j [ms] for every ordered pair of equally sized arrays (fpr_copy: 16-byte lanes, on the compute stream)."""
    import torch

    from ._lib import fptr

    n = arrs[0].numel()
    k = len(arrs)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t = [[0.0] * k for _ in range(k)]
    nd = arrs[0].dim()
    for i in range(k):
        for j in range(k):
            if i == j:
                continue
            c.call("fpr_copy", fptr(arrs[j], nd), fptr(arrs[i], nd), n)      # warm-up (clocks, first touch)
            e0.record()
            for _ in range(reps):
                c.call("fpr_copy", fptr(arrs[j], nd), fptr(arrs[i], nd), n)
            e1.record()
            e1.synchronize()
            t[i][j] = e0.elapsed_time(e1) / reps
    return t


def alloc_fields(count, *shape, pool=None, min_bytes=256 << 20, report=None, pairs=None, trial=None, trials=4, spacer_bytes=None,
                 extend_below_GBs=5050.0, extend_by=10):
    """`count` zeroed column-major float64 arrays of `shape` for kernels that stream several of them at equal offsets: the
    best-matched `count` out of `pool` candidate allocations (default count + 7, less if memory is short).  `pairs`: the
    (i, j) positions of the returned list that are streamed together (default: all); the assignment whose slowest such pair
    copies fastest wins.  `trial(arrays) -> ms` (optional): the `trials` best assignments by that measure are timed with the
    caller's own kernel and the fastest is kept.  Arrays below `min_bytes` are simply allocated.  `report` (a dict) receives
    what was measured.  A pool whose fastest pair copies below `extend_below_GBs` is all of one class (seen on one card in five:
    every pair 4700-4940 GB/s where other pools show 5140, the fused launch 0.84 ms whatever the assignment): `extend_by` more
    candidates are allocated behind further spacers, once, if memory allows."""
    import torch

    from . import ctx as _ctx
    from ._lib import fzeros

    nbytes = 8
    for s in shape:
        nbytes *= int(s)
    if report is None:
        report = {}
    report.update({"arrays": count, "bytes_per_array": nbytes, "selected": False})
    if nbytes < min_bytes or count < 2:
        return [fzeros(*shape) for _ in range(count)]
    free, _total = torch.cuda.mem_get_info()
    k = pool if pool is not None else count + 7
    k = max(count, min(k, int(0.6 * free // nbytes)))
    if k <= count:
        return [fzeros(*shape) for _ in range(count)]
    if pairs is None:
        pairs = list(itertools.combinations(range(count), 2))
    roles = sorted({i for p in pairs for i in p})          # positions that matter; the others take what is left
    c = _ctx()
    # Candidates that follow each other in one stretch of memory tend to be of one class (runs of four or five 1 GiB allocations, tools/
    # place_probe.hip); untouched spacer allocations between them spread the pool over more of the card (three classes among twelve 1 GiB candidates where the plain
    # pool showed two; ten 134 MB candidates without spacers all sat in one run and were all alike).  The spacers are reserved, never written, and freed with the rest.
    # (the label changes every four or five GiB of consecutive allocations, whatever the size of the arrays: 4 GiB spacers, 3 x the array
    # for arrays above 1.3 GiB)
    spacer = int(spacer_bytes if spacer_bytes is not None else max(4 << 30, 3 * nbytes))
    if spacer and k * nbytes + (k - 1) * spacer > 0.7 * free:
        spacer = max(0, int((0.7 * free - k * nbytes) // max(k - 1, 1)))
    cands, spacers = [], []
    for i in range(k):
        cands.append(fzeros(*shape))
        if spacer >= (64 << 20) and i + 1 < k:
            try:
                spacers.append(torch.empty(spacer, dtype=torch.uint8, device=cands[0].device))
            except RuntimeError:
                spacer = 0
    report["spacer_bytes"] = spacer if spacers else 0
    torch.cuda.synchronize()
    t = _pair_times(c, cands)
    gbs_of = lambda ms: 2.0 * nbytes / (ms * 1e-3) / 1e9
    fastest = min(0.5 * (t[i][j] + t[j][i]) for i in range(k) for j in range(i + 1, k))
    report["pool_first"] = k
    if extend_by > 0 and gbs_of(fastest) < extend_below_GBs:
        free2, _ = torch.cuda.mem_get_info()
        more = min(extend_by, int(0.7 * free2 // (nbytes + max(spacer, 0))))
        try:
            for i in range(more):
                if spacer >= (64 << 20):
                    spacers.append(torch.empty(spacer, dtype=torch.uint8, device=cands[0].device))
                cands.append(fzeros(*shape))
        except RuntimeError:
            pass
        if len(cands) > k:
            k = len(cands)
            torch.cuda.synchronize()
            t = _pair_times(c, cands)
            report["pool_extended_because_fastest_pair_GBs"] = gbs_of(fastest)
    sym = [[0.5 * (t[i][j] + t[j][i]) for j in range(k)] for i in range(k)]
    ranked = []
    for sub in itertools.permutations(range(k), len(roles)):
        where = dict(zip(roles, sub))
        ts = [sym[where[i]][where[j]] for i, j in pairs]
        ranked.append(((max(ts), sum(ts)), sub))
    ranked.sort(key=lambda x: x[0])
    # assignments that differ only by a relabelling of equivalent positions rank equal: keep distinct SETS among the best
    short, seen = [], set()
    for cost, sub in ranked:
        key = frozenset(sub)
        if key in seen:
            continue
        seen.add(key)
        short.append((cost, sub))
        if len(short) >= (trials if trial is not None else 1):
            break

    def build(sub):
        rest = [i for i in range(k) if i not in sub]
        out, it = [None] * count, iter(rest)
        for pos, i in zip(roles, sub):
            out[pos] = cands[i]
        for pos in range(count):
            if out[pos] is None:
                out[pos] = cands[next(it)]
        return out

    tried = []
    best_sub, best_ms = short[0][1], None
    if trial is not None:
        # the copy times only rank the candidates roughly; the caller's kernel decides.  Start from the best few assignments, then
        # local search: swap one position at a time for a candidate not in use, keep what is faster (a trial is a few launches).
        for cost, sub in short:
            ms = float(trial(build(sub)))
            tried.append({"arrays": list(sub), "slowest_pair_ms": cost[0], "trial_ms": ms})
            if best_ms is None or ms < best_ms:
                best_sub, best_ms = sub, ms
        def local_search(first_cand):
            nonlocal best_sub, best_ms
            budget = 6 * (k - first_cand)
            improved = True
            while improved and budget > 0:
                improved = False
                for pos in range(len(roles)):
                    for cand in range(first_cand, k):
                        if cand in best_sub or budget <= 0:
                            continue
                        sub = tuple(cand if q == pos else v for q, v in enumerate(best_sub))
                        budget -= 1
                        ms = float(trial(build(sub)))
                        tried.append({"arrays": list(sub), "trial_ms": ms})
                        if ms < 0.995 * best_ms:
                            best_sub, best_ms, improved = sub, ms, True
                first_cand = 0      # (after an improvement every candidate is worth another look)

        local_search(0)
        # Every assignment within 2.5 % of every other: the pool is of one class as far as the caller's kernel can tell (arrays that
        # fit in the Infinity Cache copy at cache speed whatever their pages: the copy times above say nothing about them).  More
        # candidates behind further spacers, once, and the search goes on among them.
        spread = max(x["trial_ms"] for x in tried) / min(x["trial_ms"] for x in tried) - 1.0
        if extend_by > 0 and "pool_extended_because_fastest_pair_GBs" not in report and spread < 0.025 and len(tried) >= 4:
            free2, _ = torch.cuda.mem_get_info()
            more = min(extend_by, int(0.7 * free2 // (nbytes + max(spacer, 0))))
            k0 = k
            try:
                for i in range(more):
                    if spacer >= (64 << 20):
                        spacers.append(torch.empty(spacer, dtype=torch.uint8, device=cands[0].device))
                    cands.append(fzeros(*shape))
            except RuntimeError:
                pass
            if len(cands) > k0:
                k = len(cands)
                torch.cuda.synchronize()
                t = _pair_times(c, cands)
                sym = [[0.5 * (t[i][j] + t[j][i]) for j in range(k)] for i in range(k)]
                report["pool_extended_because_trial_spread"] = spread
                local_search(k0)
        report["trial_ms_best"] = best_ms
        report["trial_ms_first"] = tried[0]["trial_ms"]
        report["trial_ms_worst"] = max(x["trial_ms"] for x in tried)
    flat = sorted(sym[i][j] for i in range(k) for j in range(i + 1, k))
    gbs = lambda ms: 2.0 * nbytes / (ms * 1e-3) / 1e9
    where = dict(zip(roles, best_sub))
    chosen_ts = [sym[where[i]][where[j]] for i, j in pairs]
    report.update({"selected": True, "pool": k, "chosen": list(best_sub), "pairs": [list(p) for p in pairs],
                   "pair_copy_GBs_all": {"slowest": gbs(flat[-1]), "median": gbs(flat[len(flat) // 2]), "fastest": gbs(flat[0])},
                   "pair_copy_GBs_chosen": {"slowest": gbs(max(chosen_ts)), "mean": gbs(sum(chosen_ts) / len(chosen_ts))},
                   "trials": len(tried),
                   "note": "candidate allocations timed pairwise with fpr_copy; the assignment whose slowest streamed-together pair "
                           "copies fastest is kept (finalprojectrepo.jl_amd/placement.py)"})
    out = build(best_sub)
    for a in out:
        a.zero_()
    del cands, spacers
    torch.cuda.empty_cache()
    return out
