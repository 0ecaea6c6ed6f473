#!/usr/bin/env python3
"""Randomised shapes for ONE explicit kernel id against fp32 torch ops on the GPU -- a focused companion of tools/fuzz_gpu.py
(used for the one-wave-per-SIMD kernel, whose waves synchronise through LDS flags instead of s_barrier: many multi-item
grids, ragged N, both input types, spread inputs, the same launch repeated to catch an ordering that only sometimes loses).

    python tools/stress_algo.py --algo 28 --d 128 --seconds 120 --seed 1
"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", type=int, required=True)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--causal", action="store_true")
    args = ap.parse_args()
    import torch
    import flashattention_kernel_project_amd as fa
    rng = random.Random(args.seed)
    g = torch.Generator(device="cuda").manual_seed(args.seed)
    t0, cases, fails, worst = time.time(), 0, 0, 0.0
    while time.time() - t0 < args.seconds:
        kind = rng.random()
        if kind < 0.5:      # many items per workgroup of the persistent grid
            bh, n = rng.randint(64, 400), rng.choice([256, 300, 512, 777, 1024, 1500, 2048])
        elif kind < 0.8:
            bh, n = rng.randint(1, 48), rng.randint(1, 4500)
        else:
            bh, n = rng.randint(1, 8), rng.choice([4096, 8192, 6000])
        dt = rng.choice([torch.float16, torch.bfloat16])
        spread = rng.choice([1.0, 1.0, 1.0, 1.5, 2.0, 3.0])
        q, k, v = (torch.randn(bh, n, args.d, generator=g, device="cuda") for _ in range(3))
        q, k, v = (q * spread).to(dt), (k * spread).to(dt), v.to(dt)
        scale = 1.0 / args.d ** 0.5
        s_ = torch.einsum("bid,bjd->bij", q.float(), k.float()) * scale
        if args.causal:
            s_ = s_.masked_fill(~torch.ones(n, n, dtype=torch.bool, device="cuda").tril_(), float("-inf"))
        want = torch.softmax(s_, dim=-1) @ v.float()
        tol = 1e-2 * (1.0 if dt == torch.float16 else 2.5) * (1.0 if spread < 2.0 else 2.0)
        first = None
        for rep in range(3):   # the same launch again: results must not depend on timing
            got = fa.fa_forward(q, k, v, scale=scale, algo=args.algo, causal=args.causal)
            err = float((got - want).abs().max())
            worst = max(worst, err if err == err else float("inf"))
            if first is None:
                first = got.clone()
            same = bool((got == first).all()) or bool(torch.isnan(got).any())
            if not (err <= tol) or not same:
                fails += 1
                print(f"FAIL bh={bh} n={n} dt={dt} spread={spread} rep={rep} err={err:.3e} tol={tol:.1e} repeatable={same}", flush=True)
                break
        cases += 1
    print(f"stress algo {args.algo}{' causal' if args.causal else ''} d={args.d} seed {args.seed}: {cases} cases x 3 launches, {fails} failures, worst max-abs {worst:.3e}, {time.time() - t0:.0f} s")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
