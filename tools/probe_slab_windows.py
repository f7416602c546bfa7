#!/usr/bin/env python3
"""Which windows of ONE allocation stream well together?  Copy rate (read + write, GB/s) between a 1 GiB window at offset 0 of an 8 GiB
slab and 1 GiB windows at other offsets, beside the same copy between separately allocated 1 GiB arrays (is this lease of one class?).
The seam pass of the V-cycle streams x, b and its ping-pong partner at equal offsets: 99.7 us on arrays whose placement classes differ,
115 us on one class (EXPERIMENTS 13.19, 14.8).   usage (GPU box): python3 tools/probe_slab_windows.py"""
import torch

GiB = 1 << 30
dev = torch.device("cuda:0")


def rate(dst, src, reps=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    return 2.0 * dst.numel() * 8 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


sep = [torch.zeros(GiB // 8, dtype=torch.float64, device=dev) for _ in range(6)]
print("separate allocations: copy into #0 from #k:", " ".join("%.0f" % rate(sep[0], sep[k]) for k in range(1, 6)))
print("                      copy into #1 from #k:", " ".join("%.0f" % rate(sep[1], sep[k]) for k in range(2, 6)))
del sep
torch.cuda.empty_cache()
slab = torch.zeros(8 * GiB // 8, dtype=torch.float64, device=dev)
w0 = slab[:GiB // 8]
step = 128 << 20
print("one 8 GiB allocation: copy into the window at 0 from the window at offset (MiB): GB/s")
k = GiB
while k + GiB <= 8 * GiB:
    w = slab[k // 8:(k + GiB) // 8]
    print("  %5d: %.0f" % (k >> 20, rate(w0, w)), flush=True)
    k += step
