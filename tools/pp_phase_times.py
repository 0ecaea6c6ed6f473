#!/usr/bin/env python3
"""Diagnostic: where a ping-pong iteration spends its cycles (s_memtime sums per wave)."""
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
nwg = B * H * (N // 256)
diag = torch.zeros(nwg, 8, 4, dtype=torch.int64, device="cuda")
L = fa.lib()
L.fa_debug_pp_phase_times.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
for mode in (0, 1, 2):
  print("mode", mode, "(0: both groups work, 1: group 1 idles at the barriers, 2: group 0 idles)")
  diag.zero_()
  for _ in range(3):
    rc = L.fa_debug_pp_phase_times(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B * H, N, 0.125,
                                   diag.data_ptr(), mode, None)
    assert rc == 0
  torch.cuda.synchronize()
  dg = diag.double().cpu()
  nt = N // 64
  for grp, sl in (("group0 (waves 0-3)", slice(0, 4)), ("group1 (waves 4-7)", slice(4, 8))):
    m = dg[:, sl, 0].mean().item() / nt
    s = dg[:, sl, 1].mean().item() / nt
    b = dg[:, sl, 2].mean().item() / nt
    print(f"  {grp}: per tile  matrix phase {m:.0f}  vector phase {s:.0f}  barrier wait {b:.0f}  total {m + s + b:.0f} cycles (s_memtime ticks)")

