#!/usr/bin/env python3
"""Diagnostic: where an iteration of the interleaved kernel spends its cycles (s_memtime sums per wave)."""
import ctypes as C
import os
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
nt = N // 64
for W in (8, 11, 12, 14, 15, 16, 17, 18, 13, 25, 41):
    mode, W = W, (8 if W >= 10 else W)
    desc = {8: 'full', 11: 'no LDS reads', 12: 'no MFMA', 14: 'no VALU', 15: 'no LDS, no VALU (MFMA only)', 16: 'no MFMA, no VALU (LDS reads only)', 17: 'no LDS/MFMA/VALU (staging + barrier only)', 18: 'no staging', 13: 'no LDS, no MFMA (VALU only)', 25: 'nothing but the barrier', 41: 'nothing at all'}.get(mode, '')
    nwg = B * H * (N // (32 * W))
    diag = torch.zeros(nwg, W, 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        assert L.fa_debug_il_times(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B * H, N, 0.125, diag.data_ptr(), mode, None) == 0
    torch.cuda.synchronize()
    dg = diag.double().cpu()
    print(f"W={W} mode {mode}: {desc}")
    groups = (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))) if W == 8 else (("waves 0-3", slice(0, 4)),)
    for name, sl in groups:
        c_, w_, b_ = (dg[:, sl, i].mean().item() / nt for i in range(3))
        print(f"  {name}: per tile  compute {c_:.0f}  stage wait+write {w_:.0f}  barrier {b_:.0f}  total {c_ + w_ + b_:.0f} ticks")
    pro, loop, epi = (dg[:, :, i].mean().item() for i in (4, 5, 6))
    print(f"  per workgroup: prologue {pro:.0f}  main loop {loop:.0f}  drain+store {epi:.0f} ticks  (prologue+epilogue = {100 * (pro + epi) / (pro + loop + epi):.1f} %)")
    tot = dg[:, :, :3].sum(-1) / nt
    print(f"  per-workgroup total per tile: min {tot.min().item():.0f} median {tot.median().item():.0f} max {tot.max().item():.0f}")
