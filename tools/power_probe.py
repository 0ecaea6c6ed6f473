#!/usr/bin/env python3
"""Run the forward back to back for a few seconds while sampling rocm-smi (power, sclk):
is the kernel clock- / power-limited?   python tools/power_probe.py --algo 5 --seconds 4"""
import argparse
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--ablate", type=int, default=-1, help="il kernel ablation bit mask (1 no LDS reads, 2 no MFMA, 4 no VALU, 8 no staging): timing/power only")
    args = ap.parse_args()
    import torch
    import flashattention_kernel_project_amd as fa
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(8, 16, args.N, args.d, generator=g, device="cuda").half() for _ in range(3))
    o = torch.empty(q.shape, device="cuda", dtype=torch.float32)
    samples = []
    stop = False
    if args.ablate >= 0:
        import ctypes as C
        L = fa.lib()
        L.fa_debug_il_times.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
        q3, k3, v3, o3 = (t.view(128, args.N, args.d) for t in (q, k, v, o))

        def launch():
            assert L.fa_debug_il_times(q3.data_ptr(), k3.data_ptr(), v3.data_ptr(), o3.data_ptr(), 128, args.N, 0.125, None,
                                       100 + args.ablate, None) == 0
    else:
        def launch():
            fa.fa_forward(q, k, v, out=o, algo=args.algo)

    def sampler():
        while not stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"],
                                     capture_output=True, text=True, timeout=5).stdout
                samples.append((time.time(), out))
            except Exception as e:  # noqa: BLE001
                samples.append((time.time(), f"ERR {e}"))
            time.sleep(0.3)

    th = threading.Thread(target=sampler)
    th.start()
    time.sleep(1.0)
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < args.seconds:
        for _ in range(50):
            launch()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    t1 = time.time()
    time.sleep(0.7)
    stop = True
    th.join()
    print(f"algo {args.algo} ablate {args.ablate}: {ms:.4f} ms per launch over {n} launches ({fa.attention_flops(128, args.N, args.d) / ms / 1e9:.1f} TF)")
    for ts, out in samples:
        tag = "busy" if t0 <= ts <= t1 else "idle"
        lines = [l for l in out.strip().splitlines() if l and not l.startswith("WARNING")]
        vals = lines[1].split(",") if len(lines) > 1 else []
        print(tag, f"{ts - t0:+.1f}s", "sclk", vals[7] if len(vals) > 7 else "?", "power W", vals[-1] if vals else "?")


if __name__ == "__main__":
    main()
