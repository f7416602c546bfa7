#!/usr/bin/env python3
"""How the fused diffusion pair behaves when it runs for seconds instead of milliseconds: event time per pair in chunks of 50 pairs, with the
card's clock and power beside each chunk (librocm_smi64), for the exact kernel, the contracted one (fp_contract = 1) and the exact one again.
Arrays: placement.alloc_fields as bench.py does (argument `plain`: the first five allocations instead)."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd

spec = importlib.util.spec_from_file_location("legs", os.path.join(ROOT, "finalprojectrepo.jl_amd", "benchlegs.py"))
legs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(legs)
F = fpr_amd.load(0)
ctx = F.ctx()
n = 512
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
plain = len(sys.argv) > 1 and sys.argv[1] == "plain"


def trial(arrs):
    tHt, tA, tC, tR, tB = arrs
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3):
        if i == 1:
            e0.record()
        for _ in range(3 if i else 2):
            F.part1.diffusion_3D_step_τ2(tHt, tA, tB, tC, tR, *coef)
            F.part1.diffusion_3D_step_τ2(tHt, tC, tB, tA, tR, *coef)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 12.0


rep = {}
if plain:
    Ht, A, Cc, R, B = (F.fzeros(n, n, n) for _ in range(5))
else:
    Ht, A, Cc, R, B = F.placement.alloc_fields(5, n, n, n, report=rep, pairs=[(0, 1), (0, 2), (2, 3), (1, 3), (1, 2)], trial=trial)
print("placement:", {k: rep.get(k) for k in ("chosen", "trial_ms_best", "trial_ms_plain_allocation", "trial_ms_worst")}, flush=True)
F.part1.init_local_gaussian((5.0, 5.0, 5.0), dx, dx, dx, Ht)
A.copy_(Ht); Cc.copy_(Ht)
sq = torch.zeros(2, dtype=torch.float64, device="cuda")
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for label, fma, norm in (("exact", 0, True), ("fp_contract", 1, True), ("exact", 0, True), ("exact, no norm", 0, False)):
    ctx.set_option("fp_contract", fma)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    st = []
    torch.cuda.synchronize()
    time.sleep(0.5)                       # every variant starts from a card that has idled for half a second
    t0 = time.perf_counter()
    ev[0].record()
    for c in range(chunks):
        for _ in range(25):
            F.part1.diffusion_3D_step_τ2(Ht, A, B, Cc, R, *coef, 0.2 if norm else 0.0, sq if norm else None)
            F.part1.diffusion_3D_step_τ2(Ht, Cc, B, A, R, *coef, 0.2 if norm else 0.0, sq if norm else None)
        ev[c + 1].record()
        if c % 4 == 3:
            ev[c + 1].synchronize()
            d = legs.device_state(0)
            st.append((c, d.get("sclk_MHz"), d.get("power_W"), d.get("temp_junction_C"), d.get("temp_memory_C")))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[c].elapsed_time(ev[c + 1]) / 50.0 for c in range(chunks)]
    print("%-16s wall %.4f ms per pair; by chunk of 50 pairs: %s" % (label, wall / (50 * chunks) * 1e3, " ".join("%.3f" % v for v in ms)), flush=True)
    print("   clock / power / T_j / T_mem after chunks: %s" % " ".join("%d:%s/%s/%s/%s" % (c, a, b, tj, tm) for c, a, b, tj, tm in st), flush=True)
ctx.set_option("fp_contract", 0)

# ---- the bench's own probe (benchlegs.probe_under_load: a 20 ms librocm_smi64 sampler thread beside the launches, a synchronisation per 32
#      iterations) on the same arrays, in the orders exact / contracted / exact / exact-without-sampler ----
state = {"cur": A}


def pairs_n(k, norm=True):
    for _ in range(k // 2):
        out = Cc if state["cur"] is A else A
        F.part1.diffusion_3D_step_τ2(Ht, state["cur"], B, out, R, *coef, 0.2 if norm else 0.0, sq if norm else None)
        state["cur"] = out


for label, fma in (("exact", 0), ("fp_contract", 1), ("exact", 0), ("fp_contract", 1)):
    ctx.set_option("fp_contract", fma)
    r = legs.probe_under_load(torch, 0, pairs_n, 1.0)
    print("probe %-12s %.4f ms per pair, sclk %.0f MHz, %.0f W, %d samples" % (label, 2 * r["ms_per_iteration"], r["sclk_MHz_avg"], r["power_W_avg"], r["samples"]), flush=True)
ctx.set_option("fp_contract", 0)
for label, fma in (("exact", 0), ("fp_contract", 1)):
    ctx.set_option("fp_contract", fma)
    pairs_n(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nn = 0
    while time.perf_counter() - t0 < 1.0:
        pairs_n(32)
        torch.cuda.synchronize()
        nn += 32
    print("no sampler %-12s %.4f ms per pair" % (label, (time.perf_counter() - t0) / nn * 2e3), flush=True)
ctx.set_option("fp_contract", 0)
