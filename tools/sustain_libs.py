#!/usr/bin/env python3
"""Sustained ms per launch, board power and sclk of several builds, each run back to back for --seconds (one process, one
device; energy-limited kernels need their own steady state, which short interleaved rounds do not reach):

    python tools/sustain_libs.py --libs gpurun_variants/lib_a.so,gpurun_variants/lib_b.so --algo 24 --seconds 1.5

Prints per build: ms per launch over the last 60 % of the run, mean W and sclk over the same window, joules per launch."""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--algo", type=int, default=24)
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--spread", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=1)
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
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(args.B, args.H, args.N, args.d, generator=g, device="cuda") for _ in range(3))
    q, k, v = (q * args.spread).to(dt), (k * args.spread).to(dt), v.to(dt)
    out = torch.empty(q.shape, device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    samples = []
    stop = False

    def sampler():
        while not stop:
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
                lines = [l for l in o.strip().splitlines() if l and not l.startswith("WARNING")]
                vals = lines[1].split(",") if len(lines) > 1 else []
                m = re.search(r"(\d+)Mhz", vals[7]) if len(vals) > 7 else None
                samples.append((time.time(), float(vals[-1]) if vals else float("nan"), float(m.group(1)) if m else float("nan")))
            except Exception:  # noqa: BLE001
                pass
            time.sleep(0.1)

    th = threading.Thread(target=sampler)
    th.start()
    fl = 4.0 * args.B * args.H * args.N * args.N * args.d
    for rnd in range(args.rounds):
        for name, L in libs:
            def launch():
                rc = L.fa_forward_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), args.B, args.H, args.N, args.d,
                                     1.0 / args.d ** 0.5, 0 if args.dtype == "f16" else 1, 0, args.algo, st)
                assert rc == 0, rc
            t0 = time.time()
            while time.time() - t0 < 0.4 * args.seconds:   # settle
                for _ in range(50):
                    launch()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ta = time.time()
            e0.record()
            n = 0
            while time.time() - ta < 0.6 * args.seconds:
                for _ in range(50):
                    launch()
                n += 50
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
            tb = time.time()
            ms = e0.elapsed_time(e1) / n
            win = [(w, c) for (t, w, c) in samples if ta + 0.1 <= t <= tb]
            W = sum(w for w, _ in win) / max(1, len(win))
            clk = sum(c for _, c in win) / max(1, len(win))
            print(f"{name:22s} {ms:.4f} ms ({fl / ms / 1e9:7.1f} TF)  {W:6.0f} W  sclk {clk:5.0f} MHz  {W * ms:.1f} mJ per launch  [{len(win)} samples]", flush=True)
            time.sleep(0.3)
    stop = True
    th.join()


if __name__ == "__main__":
    main()
