#!/usr/bin/env python3
"""Per-phase s_memtime stamps of fa_fwd_sk (measurement build in libfa_mi355_exp.so): where a tile's cycles go,
waves 0-3 and 4-7 separately.  Shares, not lengths (the stamps forbid overlaps the real kernel has)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FA_MI355_LIB", os.path.join(ROOT, "flashattention_kernel_project_amd", "libfa_mi355_exp.so"))
import torch  # noqa: E402
import flashattention_kernel_project_amd as fa  # noqa: E402

B, H, N, d = 8, 16, 4096, 64
BH = B * H
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(BH, N, d, generator=g, device="cuda").half() for _ in range(3))
o = torch.empty(BH, N, d, device="cuda", dtype=torch.float32)
L = fa.lib()
L.fa_lab_sk.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
L.fa_lab_sk.restype = C.c_int
nwg = 256
diag = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
tiles_per_wave = (BH * (N // 512) / nwg) * (N // 64)
slots = ["stage K", "QK^T", "softmax a", "barrier 1", "stage V", "softmax b", "PV", "barrier 2"]
for variant, name in ((0, "fold + skew"), (1, "exact + skew"), (2, "fold, lockstep"), (3, "exact, lockstep")):
    for rep in range(3):
        diag.zero_()
        rc = L.fa_lab_sk(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, N, 0.125, variant, diag.data_ptr(), None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    dd = diag.view(nwg, 8, 8).double().cpu()
    print(f"## {name}: ticks per tile per wave (mean over workgroups)")
    for lo, hi, wn in ((0, 4, "waves 0-3"), (4, 8, "waves 4-7")):
        m = dd[:, lo:hi, :].mean(dim=(0, 1)) / tiles_per_wave
        print(f"  {wn}: " + "  ".join(f"{s} {float(x):.0f}" for s, x in zip(slots, m)) + f"  | total {float(m.sum()):.0f}")
