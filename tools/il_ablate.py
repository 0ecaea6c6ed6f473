#!/usr/bin/env python3
"""Wall-clock ablation of the interleaved kernel's iteration (no in-kernel stamps): which piece of
the per-tile work sets the time.  Results of ablated builds are wrong by construction."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flashattention_kernel_project_amd as fa

B, H, N, d = 8, 16, 4096, 64
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(B * H, N, d, generator=g, device="cuda").half() for _ in range(3))
o = torch.empty(B * H, N, d, device="cuda", dtype=torch.float32)
L = fa.lib()
L.fa_debug_il_times.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
names = {0: "full", 1: "no LDS operand reads", 2: "no MFMA", 4: "no softmax VALU", 8: "no staging", 5: "MFMA + staging",
         6: "LDS reads + staging", 3: "VALU + staging", 7: "staging + barrier only", 15: "barrier only", 31: "empty loop", 11: "VALU only (no staging)", 13: "MFMA only (no staging)", 14: "LDS reads only (no staging)", 9: "MFMA + VALU (no LDS, no staging)", 10: "LDS + VALU (no MFMA, no staging)", 12: "LDS + MFMA (no VALU, no staging)"}
modes = [0, 8, 11, 13, 14, 9, 10, 12, 7, 15, 31]


def run(m, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        assert L.fa_debug_il_times(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B * H, N, 0.125, None, 100 + m, None) == 0
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


BASE = int(sys.argv[1]) if len(sys.argv) > 1 else 100   # 100: 8 waves, 2 per SIMD; 200: 4 waves, ONE per SIMD
if BASE == 200:
    modes = [0, 8, 11, 13, 14, 9, 10, 12, 31]
tiles_per_simd = 512 if BASE == 100 else 1024             # wave-tile slots a SIMD runs back to back
_run = run
run = lambda m, iters=10: _run(m - 100 + BASE, iters)
for m in modes:
    run(m, 3)
res = {m: [] for m in modes}
for _ in range(5):
    for m in modes:
        res[m].append(run(m))
for m in modes:
    t = statistics.median(res[m])
    print(f"{names[m]:28s} {t:.4f} ms   ~{t * 1e-3 * 2.0e9 / tiles_per_simd:.0f} cycles per {'tile-pair' if BASE == 100 else 'wave-tile'} per SIMD at 2.0 GHz")
