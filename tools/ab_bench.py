#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (HIP events on the launch stream).

    python tools/ab_bench.py --algos 2,3 --rounds 10 --iters 20 [--dtype f16] [--B 8 --H 16 --N 4096 --d 64]

Prints per variant: median / min ms and TFLOP/s, plus max-abs difference of each variant's output
against the first variant's (a sanity check, not the parity test).
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algos", default="2,3")
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--out", default="f32")
    ap.add_argument("--causal", action="store_true", help="fa_forward_causal (algos 0,1,2); FLOPs counted as 4*BH*d*N(N+1)/2")
    args = ap.parse_args()
    import torch
    import flashattention_kernel_project_amd as fa
    algos = [int(a) for a in args.algos.split(",")]
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    odt = torch.float32 if args.out == "f32" else dt
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(args.B, args.H, args.N, args.d, generator=g, device="cuda").to(dt) for _ in range(3))
    outs = {a: torch.empty(q.shape, device="cuda", dtype=odt) for a in algos}
    for a in algos:
        for _ in range(3):
            fa.fa_forward(q, k, v, out=outs[a], algo=a, causal=args.causal)
    torch.cuda.synchronize()
    times = {a: [] for a in algos}
    for _ in range(args.rounds):
        for a in algos:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fa.fa_forward(q, k, v, out=outs[a], algo=a, causal=args.causal)
            e1.record()
            torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / args.iters)
    fl = fa.attention_flops(args.B * args.H, args.N, args.d)
    if args.causal:
        fl *= (args.N + 1) / (2.0 * args.N)
    base = outs[algos[0]].float()
    for a in algos:
        med, mn = statistics.median(times[a]), min(times[a])
        diff = float((outs[a].float() - base).abs().max())
        print(f"algo {a}: median {med:.4f} ms ({fl / med / 1e9:.1f} TF)  min {mn:.4f} ms ({fl / mn / 1e9:.1f} TF)  "
              f"max|d| vs algo {algos[0]} = {diff:.2e}", flush=True)


if __name__ == "__main__":
    main()
