#!/usr/bin/env python3
"""Randomised shape / dtype / kernel sweep against fp32 torch ops on the GPU (run on an MI355X):

    python tools/fuzz_gpu.py --seconds 120 --seed 1

Each case draws (BH, N, d, dtype, output type, scale sign, causal, kernel variant, input spread),
runs the C-ABI forward and compares every output element with softmax(QK^T*scale [+mask])V computed
in fp32 torch ops.  Also draws split-KV cases (Nq != Nk).  Prints every failure and a summary line;
exit code 1 if any case exceeded the tolerance (1e-2 max-abs, widened only by what a 16-bit OUTPUT format or
an A/B kernel's un-rounded bf16 row sums impose on peaked rows; see the comment at `tol`)."""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ref(torch, q, k, v, scale, causal):
    s = torch.einsum("bid,bjd->bij", q.float(), k.float()) * scale
    if causal:
        n = q.shape[1]
        s = s.masked_fill(~torch.ones(n, n, dtype=torch.bool, device=q.device).tril_(), float("-inf"))
    return torch.softmax(s, dim=-1) @ v.float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    import flashattention_kernel_project_amd as fa
    rng = random.Random(args.seed)
    g = torch.Generator(device="cuda").manual_seed(args.seed)
    exp = fa.lib().fa_mi355_has_experiments() == 1   # the A/B kernels exist only in libfa_mi355_exp.so (FA_MI355_LIB=...)
    plain = {64: (0, 1, 2, 5, 6, 23, 24, 24, 24, 26, 26, 27, 27, 29, 29) + ((13, 14, 16, 17, 18, 19, 20, 21, 22, 25) if exp else ()),
             128: (0, 1, 2, 23, 24, 24, 26, 26, 28, 28) + ((13, 14, 16, 21) if exp else ())}
    caus = {64: (0, 1, 2, 6, 24, 24) + ((13,) if exp else ()), 128: (0, 1, 2, 6, 24, 24, 28, 28) + ((13,) if exp else ())}
    t0, cases, fails, worst = time.time(), 0, 0, 0.0
    next_note = t0 + 60.0
    while time.time() - t0 < args.seconds:
        kind = rng.choice(["plain", "plain", "causal", "split"])
        d = rng.choice([16, 32, 64, 64, 64, 128, 128, 256]) if kind != "split" else rng.choice([64, 128])
        dt = rng.choice([torch.float16, torch.bfloat16])
        bh = rng.randint(1, 6)
        n = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, rng.randint(1, 2500)])
        if rng.random() < 0.08:   # grids large enough for AUTO's big-grid kernels
            bh, n = rng.choice([128, 256, 300]), rng.choice([300, 512, 640, 1000])
        spread = rng.choice([1.0, 1.0, 0.3, 2.5])
        scale = (1.0 / d ** 0.5) * rng.choice([1.0, 1.0, 1.0, -1.0, 0.5])
        out_same = rng.random() < 0.3
        odt = dt if out_same else torch.float32
        q = (torch.randn(bh, n, d, generator=g, device="cuda") * spread).to(dt)
        if kind == "split":
            nk = rng.choice([64, 65, 200, 1000, 4096, rng.randint(1, 9000)])
            k = (torch.randn(bh, nk, d, generator=g, device="cuda") * spread).to(dt)
            v = torch.randn(bh, nk, d, generator=g, device="cuda").to(dt)
            nq = rng.choice([1, 1, 2, 8, 16, 33, 130])
            q = q[:, :nq].contiguous() if nq <= n else (torch.randn(bh, nq, d, generator=g, device="cuda") * spread).to(dt)
            got = fa.fa_forward_splitkv(q[None], k[None], v[None], scale=scale, out_dtype=odt)[0]
            want = ref(torch, q, k, v, scale, False)
            algo = -1
            desc = f"split bh={bh} nq={q.shape[1]} nk={nk} d={d} {dt} out_same={out_same} scale={scale:.4f} spread={spread}"
        else:
            k = (torch.randn(bh, n, d, generator=g, device="cuda") * spread).to(dt)
            v = torch.randn(bh, n, d, generator=g, device="cuda").to(dt)
            causal = kind == "causal"
            table = caus if causal else plain
            algo = rng.choice(table.get(d, (0, 1)))
            got = fa.fa_forward(q, k, v, scale=scale, out_dtype=odt, algo=algo, causal=causal)
            want = ref(torch, q, k, v, scale, causal)
            desc = f"{kind} bh={bh} n={n} d={d} {dt} out_same={out_same} algo={algo} scale={scale:.4f} spread={spread}"
        torch.cuda.synchronize()
        err = (got.float() - want).abs().max().item()
        # 1e-2 (north-star) plus what the 16-bit formats themselves impose on peaked rows (O ~ one V row):
        # a bf16 OUTPUT rounds by |O| * 2^-9.
        vmax = v.float().abs().max().item()
        tol = 1e-2
        if dt == torch.bfloat16 and (spread > 2 or (kind == "causal")):
            # (under the mask the first rows have two or three keys only: peaked by construction)
            # sharply peaked rows with two or three comparable dominant weights: each bf16 weight carries 2^-9
            # relative rounding, so O moves by up to 2^-9 * (spread of the dominant V rows) -- the format's limit
            tol += vmax * 2.0 ** -9
        if out_same:
            tol += vmax * (2.0 ** -9 if dt == torch.bfloat16 else 2.0 ** -12)
        cases += 1
        if time.time() >= next_note:   # a silent GPU job is taken to be hung
            print(f"... {cases} cases, {fails} failures so far", flush=True)
            next_note += 60.0
        worst = max(worst, err)
        if not (err <= tol) or not torch.isfinite(got).all():
            fails += 1
            print(f"FAIL err={err:.3e} tol={tol:.0e}: {desc}", flush=True)
    from flashattention_kernel_project_amd import capi as _capi
    libname = os.path.basename(_capi.LIB_PATH)
    print(f"fuzz: seed {args.seed} library {libname} ({fa.version()}): {cases} cases in {time.time() - t0:.0f} s, {fails} failures, worst max-abs {worst:.3e}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
