#!/usr/bin/env python3
"""Per-launch duration of the forward right after idle (the first ~30 ms run slower while the clocks settle):
what the driver's `--steps 20 --warmup 5` window sees, against the sustained rate.

    python tools/transient_probe.py [--algo 0] [--n 80] [--pre idle|gemm|self]
Writes one CSV row per launch to stdout: pre-condition, index, microseconds.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import flashattention_kernel_project_amd as fa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--algo", type=int, default=0)
ap.add_argument("--n", type=int, default=80)
ap.add_argument("--dtype", default="f16")
args = ap.parse_args()
B, H, N, d = 8, 16, 4096, 64
dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(42)
q, k, v = (torch.randn(B, H, N, d, generator=g, device="cuda").to(dt) for _ in range(3))
o = torch.empty(B, H, N, d, device="cuda", dtype=torch.float32)
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)


def launches(n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        fa.fa_forward(q, k, v, out=o, algo=args.algo)
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]


fa.fa_forward(q, k, v, out=o, algo=args.algo)
torch.cuda.synchronize()
print("pre,index,us")
for pre in ("idle2s", "gemm50ms", "self300", "idle2s_again"):
    if pre.startswith("idle"):
        time.sleep(2.0)
    elif pre == "gemm50ms":
        time.sleep(2.0)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            (a @ a).sum().item()
    elif pre == "self300":
        time.sleep(2.0)
        for _ in range(300):
            fa.fa_forward(q, k, v, out=o, algo=args.algo)
        torch.cuda.synchronize()
    us = launches(args.n)
    for i, u in enumerate(us):
        print(f"{pre},{i},{u:.1f}")
    w = us[5:25]
    print(f"# {pre}: launches 5..24 mean {sum(w) / len(w):.1f} us; launches 40.. mean {sum(us[40:]) / len(us[40:]):.1f} us", file=sys.stderr)
