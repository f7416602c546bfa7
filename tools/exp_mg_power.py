#!/usr/bin/env python3
"""Clocks and power under back-to-back 4097^2 solves (librocm_smi64 every 20 ms, as bench.py's power probe does for the diffusion kernels)."""
import importlib.util
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
F = fpr_amd.load(0)
mg = F.multigrid
n = 4097
h = 1.0 / (n - 1)
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        d = bench.device_state(0)
        samples.append((d.get("sclk_MHz"), d.get("power_W")))
        stop.wait(0.02)


for _ in range(3):
    x.zero_(); mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False)
F.synchronize()
th = threading.Thread(target=sampler, daemon=True)
t0 = time.perf_counter()
th.start()
k = 0
while time.perf_counter() - t0 < 1.5:
    x.zero_(); mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False); k += 1
F.synchronize()
dt = time.perf_counter() - t0
stop.set(); th.join(2.0)
late = samples[len(samples) // 3:]
print("%d solves in %.2f s = %.3f ms per solve; sclk %.0f MHz avg (min %.0f), power %.0f W avg (max %.0f) of cap %s"
      % (k, dt, dt / k * 1e3, sum(s[0] for s in late) / len(late), min(s[0] for s in late), sum(s[1] for s in late) / len(late),
         max(s[1] for s in late), bench.device_state(0).get("power_cap_W")))
