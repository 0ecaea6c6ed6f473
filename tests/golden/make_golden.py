#!/usr/bin/env python3
"""Generate the committed golden vectors in tests/golden/*.npz.

Run in the build container only (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py

Expected outputs come from the REFERENCE'S OWN CPU oracle -- `flashattn_cpu_ref`,
GEMM/FlashAttention Forward Fused/flashattn_forward_fused_5_4_2.cu:224-272 -- compiled from
where it lies by oracle/build_ref.sh and executed here.  Inputs are our portable counter-based
stream (oracle/attention_cpu.c: fa_oracle_fill), rounded through fp16/bf16, drawn Q->K->V as
the reference drivers do (flashattn_streaming_16x16_mw.cu:332-349).  A fixture is data only:
uint16 encodings of Q,K,V, the format id, and the fp32 expected O.

The reference holds no golden vectors of its own; these pin our oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name, BH, N, D, fmt, seed, dist
GENERAL = [
    ("cfg1_bh1_n128_d64_f16_normal", 1, 128, 64, o.F16, 42, o.NORMAL),      # BASELINE configs[0]/[1]
    ("ref542_bh2_n128_d64_f16_uniform", 2, 128, 64, o.F16, 42, o.UNIFORM),  # 5_4_2.cu:276-306 driver shape
    ("bh2_n128_d64_bf16_normal", 2, 128, 64, o.BF16, 7, o.NORMAL),
    ("ragged_bh3_n100_d64_f16_normal", 3, 100, 64, o.F16, 11, o.NORMAL),    # N % 64 != 0, N % 16 != 0
    ("ragged_bh2_n321_d64_bf16_normal", 2, 321, 64, o.BF16, 12, o.NORMAL),  # > one 256-row block, odd tail
    ("bh2_n192_d128_f16_normal", 2, 192, 128, o.F16, 13, o.NORMAL),
    ("bh1_n77_d128_bf16_normal", 1, 77, 128, o.BF16, 14, o.NORMAL),
    ("bh2_n64_d16_f16_normal", 2, 64, 16, o.F16, 15, o.NORMAL),             # generic-kernel head dims
    ("bh2_n50_d32_f16_uniform", 2, 50, 32, o.F16, 16, o.UNIFORM),
    ("memprofile_bh4_n512_d64_f16_normal", 4, 512, 64, o.F16, 17, o.NORMAL),  # memprofile.cu:409-411 shape
]

# 16x16 streaming family: name, B, L, seed (reference driver: B=1024, L=128, N(0,1) seed 42,
# flashattn_streaming_16x16_mw.cu:322-349; a B=8 slice is committed)
STREAM16 = [
    ("s16_b8_l128_normal", 8, 128, 42),
    ("s16_b4_l16_normal", 4, 16, 43),
    ("s16_b3_l80_normal", 3, 80, 44),
]


def main():
    o.build()
    assert o.have_ref(), "oracle/_ref/libref_cpu.so missing: /root/reference not available?"
    for name, bh, n, d, fmt, seed, dist in GENERAL:
        (q, k, v), (qb, kb, vb) = o.make_qkv(bh, n, d, fmt, seed, dist)
        expected = o.reference_forward(q, k, v)   # the reference's own function
        np.savez_compressed(os.path.join(HERE, name + ".npz"), q=qb, k=kb, v=vb,
                            fmt=np.int32(fmt), expected=expected, kind="general")
        print(name, expected.shape, float(np.abs(expected).max()))
    for name, b, l, seed in STREAM16:
        # Q [B,16,16], K [B,16,L], V [B,L,16], one stream in the order Q -> K -> V
        nq, nk = b * 256, b * 16 * l
        qb = o.encode16(o.fill(nq, seed, 0), o.F16).reshape(b, 16, 16)
        kb = o.encode16(o.fill(nk, seed, nq), o.F16).reshape(b, 16, l)
        vb = o.encode16(o.fill(nk, seed, nq + nk), o.F16).reshape(b, l, 16)
        q, k, v = (o.decode16(x, o.F16) for x in (qb, kb, vb))
        # Expected values from the reference's general oracle at D=16 (scale = 1/sqrt(16) = 0.25 =
        # the 16x16 drivers' scale, mw.cu:329): rows 0..15 of an L-row self-attention whose first
        # 16 query rows are Q and whose keys are K^T.  Differs from the 16x16 family's own
        # normalisation only by its EPS = 1e-6 in the denominator (relative 1e-6/l, l >= 1).
        nn = max(l, 16)
        qq = np.zeros((b, nn, 16), np.float32)
        kk = np.zeros((b, nn, 16), np.float32)
        vv = np.zeros((b, nn, 16), np.float32)
        qq[:, :16] = q
        kk[:, :l] = np.transpose(k, (0, 2, 1))
        vv[:, :l] = v
        assert nn == l, "L >= 16 for every committed case"
        expected = o.reference_forward(qq, kk, vv)[:, :16].copy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), q=qb, k=kb, v=vb,
                            fmt=np.int32(o.F16), expected=expected, kind="stream16")
        print(name, expected.shape)


if __name__ == "__main__":
    main()
