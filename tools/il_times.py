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
for W in (8, 4, 11, 12, 13, 14, 15, 16):
    mode, W = W, (8 if W >= 10 else W)
    nwg = B * H * (N // (32 * W))
    diag = torch.zeros(nwg, W, 4, dtype=torch.int64, device="cuda")
    for _ in range(3):
        assert L.fa_debug_il_times(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B * H, N, 0.125, diag.data_ptr(), mode, None) == 0
    torch.cuda.synchronize()
    dg = diag.double().cpu()
    print(f"W={W} waves per workgroup, mode {mode} (11: no LDS operand reads, 12: no MFMA, 13: no softmax VALU, 14: no staging, 15: MFMA operands not from LDS but reads still issued, 16: no barrier)")
    groups = (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))) if W == 8 else (("waves 0-3", slice(0, 4)),)
    for name, sl in groups:
        c_, w_, b_ = (dg[:, sl, i].mean().item() / nt for i in range(3))
        print(f"  {name}: per tile  compute {c_:.0f}  stage wait+write {w_:.0f}  barrier {b_:.0f}  total {c_ + w_ + b_:.0f} ticks")
    tot = dg[:, :, :3].sum(-1) / nt
    print(f"  per-workgroup total per tile: min {tot.min().item():.0f} median {tot.median().item():.0f} max {tot.max().item():.0f}")
