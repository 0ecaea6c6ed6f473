#!/usr/bin/env python3
"""Minimal launch loop for rocprofv3: `rocprofv3 ... -- python3 tools/profile_run.py --algo 0`."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--dtype", default="f16")
    args = ap.parse_args()
    import torch
    import flashattention_kernel_project_amd as fa
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(args.B, args.H, args.N, args.d, generator=g, device="cuda").to(dt) for _ in range(3))
    o = torch.empty(q.shape, device="cuda", dtype=torch.float32)
    for _ in range(args.iters):
        fa.fa_forward(q, k, v, out=o, algo=args.algo)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
