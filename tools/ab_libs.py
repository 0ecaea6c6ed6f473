#!/usr/bin/env python3
"""Interleaved A/B of several builds of libfa_mi355.so in one process (same device, same data).

    python tools/ab_libs.py --libs gpurun_variants/lib_base.so,gpurun_variants/lib_x.so --algo 5
"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--algo", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--spread", type=float, default=1.0, help="scale of Q and K (the logits grow with its square)")
    args = ap.parse_args()
    import torch
    from flashattention_kernel_project_amd import capi
    capi._share_torch_hip_runtime()
    libs = []
    for path in args.libs.split(","):
        L = C.CDLL(os.path.abspath(path))
        L.fa_forward_ex.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float] + [C.c_int] * 3 + [C.c_void_p]
        L.fa_forward_ex.restype = C.c_int
        libs.append((os.path.basename(path), L))
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(args.B, args.H, args.N, args.d, generator=g, device="cuda") for _ in range(3))
    q, k, v = (q * args.spread).half(), (k * args.spread).half(), v.half()
    outs = [torch.empty(q.shape, device="cuda", dtype=torch.float32) for _ in libs]
    st = torch.cuda.current_stream().cuda_stream

    def run(i):
        rc = libs[i][1].fa_forward_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), outs[i].data_ptr(), args.B, args.H,
                                      args.N, args.d, 1.0 / args.d ** 0.5, 0, 0, args.algo, st)
        assert rc == 0, rc

    for i in range(len(libs)):
        for _ in range(3):
            run(i)
    torch.cuda.synchronize()
    times = [[] for _ in libs]
    for _ in range(args.rounds):
        for i in range(len(libs)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / args.iters)
    fl = 4.0 * args.B * args.H * args.N * args.N * args.d
    for i, (name, _) in enumerate(libs):
        med, mn = statistics.median(times[i]), min(times[i])
        diff = float((outs[i] - outs[0]).abs().max())
        print(f"{name:24s} median {med:.4f} ms ({fl / med / 1e9:.1f} TF)  min {mn:.4f} ms ({fl / mn / 1e9:.1f} TF)  max|d| vs first {diff:.1e}", flush=True)


if __name__ == "__main__":
    main()
