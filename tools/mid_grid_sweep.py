#!/usr/bin/env python3
"""Mid-size grids at d=64: microseconds per launch of the candidate kernels (il 256-row / 128-row workgroups, the rolling
pipeline on 512 / 256 / 128-row workgroups) over (B*H, N) shapes, to place FA_ALGO_AUTO's thresholds.
    python tools/mid_grid_sweep.py [--dtype f16]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import flashattention_kernel_project_amd as fa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f16")
ap.add_argument("--algos", default="")
ap.add_argument("--d", type=int, default=64)
args = ap.parse_args()
dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
d = args.d
algos = [int(a) for a in (args.algos or ("0,5,6,24,26,27" if d == 64 else "0,24,26")).split(",")]
shapes = [(32, 1024), (8, 1024), (16, 2048), (64, 1024), (128, 512), (64, 512), (256, 256), (16, 4096), (32, 4096), (48, 4096),
          (8, 8192), (24, 2048), (96, 1024), (128, 1024), (40, 1536), (128, 300)]
if d == 128:
    shapes = [(32, 1024), (8, 1024), (16, 2048), (64, 1024), (128, 512), (64, 512), (16, 4096), (32, 4096), (48, 2048), (8, 8192), (24, 2048),
              (96, 1024), (128, 1024), (40, 1536), (128, 300), (300, 300), (256, 256)]
g = torch.Generator(device="cuda").manual_seed(0)
print(f"# us per launch, {args.dtype}, d={d}; algos {algos}; AUTO picks fa_selected_algo")
for BH, N in shapes:
    q, k, v = (torch.randn(1, BH, N, d, generator=g, device="cuda").to(dt) for _ in range(3))
    o = torch.empty(1, BH, N, d, device="cuda", dtype=torch.float32)
    ref = None
    row = []
    for a in algos:
        for _ in range(5):
            fa.fa_forward(q, k, v, out=o, algo=a)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fa.fa_forward(q, k, v, out=o, algo=a)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        if ref is None:
            ref = o.clone()
        err = float((o - ref).abs().max())
        assert err < 5e-3, (BH, N, a, err)
        row.append(f"{a}:{statistics.median(ts):7.1f}")
    sel = fa.lib().fa_selected_algo(1, BH, N, d, 0 if args.dtype == "f16" else 1)
    tf = 4.0 * BH * N * N * d / 1e6
    print(f"BH {BH:4d} N {N:5d}  auto={sel:2d}  " + "  ".join(row) + f"   ({tf / min(float(r.split(':')[1]) for r in row):6.1f} TF best)", flush=True)
