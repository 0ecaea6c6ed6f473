#!/usr/bin/env python3
"""Stage / ablation breakdown of the shipped d=64 fp16 stream (fa_fwd_w64x) from its measurement build
(csrc/fa_lab_w64x.hip in libfa_mi355_exp.so, `make -C flashattention_kernel_project_amd/csrc experimental`).

  1. the un-ablated lab instances against the product kernel's output (sanity, not the parity test)
  2. wall-clock ablation (HIP events, interleaved rounds): the loop minus LDS reads / MFMA / VALU / staging / barrier
  3. in-kernel s_memtime stamps per phase, waves 0-3 and waves 4-7 separately (shares, not lengths)

Reference analogues: flashattn_stage_latency_breakdown.cu:181-207, flashattn_forward_cp_async_stall.cu:93-206,
flashattn_tensorcore_util_profile.cu:69, flashattn_forward_softmax_bottleneck.cu:66.
"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FA_MI355_LIB", os.path.join(ROOT, "flashattention_kernel_project_amd", "libfa_mi355_exp.so"))
import torch  # noqa: E402
import flashattention_kernel_project_amd as fa  # noqa: E402

B, H, N, d = 8, 16, 4096, 64
BH = B * H
FLOPS = 4.0 * BH * N * N * d
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(BH, N, d, generator=g, device="cuda").half() for _ in range(3))
o = torch.empty(BH, N, d, device="cuda", dtype=torch.float32)
L = fa.lib()
assert L.fa_mi355_has_experiments() == 1, "needs libfa_mi355_exp.so"
L.fa_lab_w64x.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.fa_lab_w64x.restype = C.c_int
SCALE = 1.0 / d ** 0.5


def lab(kstruct, abl, diag=None):
    rc = L.fa_lab_w64x(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, N, SCALE, kstruct, abl,
                       diag.data_ptr() if diag is not None else None, None)
    assert rc == 0, (kstruct, abl, rc)


def product(algo):
    fa.fa_forward(q.view(B, H, N, d), k.view(B, H, N, d), v.view(B, H, N, d), out=o.view(B, H, N, d), algo=algo)


def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    print(f"# B{B} H{H} N{N} d{d} fp16 -> fp32; {fa.version()}")
    # ---- 1. sanity ----
    product(16)
    ref = o.clone()
    for ks, name in ((0, "struct 0 (shipped order)"), (1, "struct 1 (two halves, lockstep)"), (3, "struct 1 + skew")):
        o.zero_()
        lab(ks, 0)
        torch.cuda.synchronize()
        print(f"sanity {name:34s} max|lab - product| = {float((o - ref).abs().max()):.2e}")

    # ---- 2. wall-clock ablation ----
    names = {0: "full", 1: "no LDS operand reads", 2: "no MFMA", 4: "no softmax VALU", 8: "no staging", 16: "no barrier",
             3: "VALU + staging (no LDS reads, no MFMA)", 5: "MFMA + staging (no LDS reads, no VALU)",
             6: "LDS reads + staging (no MFMA, no VALU)", 7: "staging + barrier only", 9: "MFMA + VALU (no LDS reads, no staging)",
             11: "VALU only", 13: "MFMA only", 14: "LDS reads only", 15: "barrier only", 31: "empty loop"}
    cases = [("product algo 16 (w64x)", lambda: product(16)), ("product algo 13 (w64, 32x32x16)", lambda: product(13)),
             ("product algo 5 (il)", lambda: product(5))]
    for ks, nm in ((0, "s0"), (1, "s1"), (3, "s1+skew")):
        cases.append((f"lab {nm} full", (lambda ks=ks: lab(ks, 0))))
    for a in (1, 2, 4, 8, 16, 3, 5, 6, 7, 9, 11, 13, 14, 15, 31):
        cases.append((f"lab s0 {names[a]}", (lambda a=a: lab(0, a))))
    for a in (1, 2, 4, 8):
        cases.append((f"lab s1+skew {names[a]}", (lambda a=a: lab(3, a))))
    for _, fn in cases:
        timed(fn, 3)
    res = [[] for _ in cases]
    for _ in range(5):
        for i, (_, fn) in enumerate(cases):
            res[i].append(timed(fn))
    print("\n## wall-clock (median of 5 interleaved rounds x 10 launches); cycles per tile-pair per SIMD assume 2.0 GHz")
    for (name, _), r in zip(cases, res):
        t = statistics.median(r)
        print(f"{name:52s} {t:.4f} ms  {FLOPS / t / 1e9:7.1f} TF-equivalent  ~{t * 1e-3 * 2.0e9 / 512:.0f} cyc/tile-pair")

    # ---- 3. stamps ----
    nwg = 256
    diag = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
    tiles_per_wave = (BH * (N // 512) / nwg) * (N // 64)
    for ks, name, slots in (
            (0, "struct 0 (shipped order)", ["load issue", "QK^T", "softmax", "PV a", "stage write", "PV b", "barrier", "-"]),
            (1, "struct 1 (two halves, lockstep)", ["stage K", "QK^T", "softmax 0,1", "barrier 1", "stage V", "softmax 2,3", "PV", "barrier 2"]),
            (3, "struct 1 + skew", ["stage K", "QK^T", "softmax 0,1", "barrier 1", "stage V", "softmax 2,3", "PV", "barrier 2"])):
        diag.zero_()
        for _ in range(3):
            lab(ks, 0, diag)
        diag.zero_()
        lab(ks, 0, diag)
        torch.cuda.synchronize()
        dd = diag.view(nwg, 8, 8).double().cpu()
        print(f"\n## stamps, {name}: s_memtime ticks per tile per wave (mean over workgroups)")
        for lo, hi, wn in ((0, 4, "waves 0-3"), (4, 8, "waves 4-7")):
            m = dd[:, lo:hi, :].mean(dim=(0, 1)) / tiles_per_wave
            parts = "  ".join(f"{s} {float(x):.0f}" for s, x in zip(slots, m) if s != "-")
            print(f"  {wn}: {parts}  | total {float(m.sum()):.0f}")


if __name__ == "__main__":
    main()
