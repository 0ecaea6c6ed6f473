#!/usr/bin/env python3
"""Per-item cost of the persistent pipeline (prologue: Q, first K/V tiles, reference; epilogue: normalise + store O): launches
with the same number of items (256 = one per CU) and 64 / 128 / 256 tiles each, then 1 / 2 / 4 items per CU at 64 tiles.
us per launch, us per tile per item, and the intercept of the linear fit = what an item costs outside its tiles."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import flashattention_kernel_project_amd as fa  # noqa: E402


def t_us(BH, N, d=64, algo=24, iters=20, rounds=5):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(1, BH, N, d, generator=g, device="cuda").half() for _ in range(3))
    o = torch.empty(1, BH, N, d, device="cuda", dtype=torch.float32)
    for _ in range(60):
        fa.fa_forward(q, k, v, out=o, algo=algo)
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fa.fa_forward(q, k, v, out=o, algo=algo)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


rows = []
for BH, N in ((32, 4096), (16, 8192), (8, 16384), (4, 32768)):
    us = t_us(BH, N)
    tiles = N // 64
    items = BH * (N // 512)
    rows.append((tiles, us))
    print(f"BH {BH:3d} N {N:6d}: {items} items x {tiles} tiles  {us:8.1f} us  {us / tiles:.3f} us/tile  {4.0 * BH * N * N * 64 / us / 1e6:7.1f} TF", flush=True)
(x0, y0), (x1, y1) = rows[0], rows[-1]
slope = (y1 - y0) / (x1 - x0)
print(f"fit: {slope:.3f} us per tile, {y0 - slope * x0:.1f} us per item outside its tiles")
for BH in (32, 64, 128, 256):
    us = t_us(BH, 4096)
    print(f"BH {BH:3d} N 4096: {BH * 8 // 256} items per CU  {us:8.1f} us  {us / (BH * 8 // 256):8.1f} us per round  {4.0 * BH * 4096 * 4096 * 64 / us / 1e6:7.1f} TF", flush=True)
