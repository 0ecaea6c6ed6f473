"""CPU tier: the oracle against the committed golden vectors (which came from the reference's
own CPU oracle), against oracle/_ref when it is present, and its helpers against numpy."""
import glob
import os

import numpy as np
import pytest


def _load(path):
    z = np.load(path)  # allow_pickle=False (default)
    return z


def _general_cases(golden_dir):
    return sorted(p for p in glob.glob(os.path.join(golden_dir, "*.npz")) if not os.path.basename(p).startswith("s16_"))


def test_golden_present(golden_dir):
    assert len(_general_cases(golden_dir)) >= 10
    assert len(glob.glob(os.path.join(golden_dir, "s16_*.npz"))) >= 3


def test_oracle_matches_golden_general(oracle, golden_dir):
    for path in _general_cases(golden_dir):
        z = _load(path)
        fmt = int(z["fmt"])
        q, k, v = (oracle.decode16(z[n], fmt) for n in ("q", "k", "v"))
        exp = z["expected"]
        got64 = oracle.forward(q, k, v, accum=1)
        got32 = oracle.forward(q, k, v, accum=0, nthreads=2)
        # double-accumulator mode restates the reference's function exactly
        assert oracle.max_abs(got64, exp) <= 1e-7, path
        # fp32-accumulator mode (north-star's naive fp32 reference) agrees to fp32 rounding
        assert oracle.max_abs(got32, exp) <= 5e-6, path


def test_oracle_matches_reference_build(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(0)
    for bh, n, d in [(1, 1, 16), (2, 33, 48), (1, 130, 64), (3, 64, 128)]:
        q, k, v = (rng.standard_normal((bh, n, d)).astype(np.float32) for _ in range(3))
        ref = oracle.reference_forward(q, k, v)
        assert oracle.max_abs(oracle.forward(q, k, v, accum=1), ref) <= 1e-7
        assert oracle.max_abs(oracle.forward(q, k, v, accum=0), ref) <= 5e-6


def test_oracle_rows_subset_and_threads(oracle):
    (q, k, v), _ = oracle.make_qkv(3, 70, 32, oracle.F16, seed=5)
    full = oracle.forward(q, k, v)
    part = oracle.forward(q, k, v, nthreads=3, bh_range=(1, 3), row_range=(10, 40))
    assert np.array_equal(part[1:3, 10:40], full[1:3, 10:40])
    assert not part[0].any() and not part[1:3, :10].any() and not part[1:3, 40:].any()


def test_causal_oracle_is_prefix_of_plain_oracle(oracle):
    """The causal oracle is not a reference function; it is pinned through this identity with the
    pinned plain oracle: row i over keys 0..i, and an explicit numpy masked softmax."""
    (q, k, v), _ = oracle.make_qkv(2, 45, 32, seed=3)
    c = oracle.forward(q, k, v, causal=True)
    for i in (0, 1, 17, 44):
        p = oracle.forward(q[:, :i + 1], k[:, :i + 1], v[:, :i + 1])
        assert np.array_equal(c[:, i], p[:, i])
    s = np.einsum("bid,bjd->bij", q.astype(np.float64), k.astype(np.float64)) / np.sqrt(32.0)
    s = np.where(np.tril(np.ones((45, 45), bool)), s, -np.inf)
    w = np.exp(s - s.max(-1, keepdims=True))
    want = np.einsum("bij,bjd->bid", w / w.sum(-1, keepdims=True), v.astype(np.float64))
    assert np.abs(c - want).max() < 2e-6
    assert np.array_equal(oracle.forward(q, k, v, causal=True, accum=1, nthreads=4, row_range=(10, 20))[:, 10:20],
                          oracle.forward(q, k, v, causal=True, accum=1)[:, 10:20])


def test_oracle_scale_argument(oracle):
    (q, k, v), _ = oracle.make_qkv(1, 40, 16, oracle.F16, seed=6)
    a = oracle.forward(q, k, v, scale=0.5)
    b = oracle.forward(q * 2.0, k, v, scale=0.25)
    assert oracle.max_abs(a, b) <= 1e-6


def test_streaming16_oracle_matches_golden(oracle, golden_dir):
    for path in sorted(glob.glob(os.path.join(golden_dir, "s16_*.npz"))):
        z = _load(path)
        q, k, v = (oracle.decode16(z[n], oracle.F16) for n in ("q", "k", "v"))
        got = oracle.streaming_16x16(q, k, v, scale=0.25)
        # expected came from the reference's general oracle (no EPS): EPS = 1e-6 in the
        # denominator changes O by at most 1e-6 relative (l >= 1)
        assert oracle.max_abs(got, z["expected"]) <= 5e-6, path


def test_streaming16_equals_general_oracle(oracle):
    b, l = 5, 48
    q = oracle.decode16(oracle.encode16(oracle.fill(b * 256, 3, 0), 0), 0).reshape(b, 16, 16)
    k = oracle.decode16(oracle.encode16(oracle.fill(b * 16 * l, 3, b * 256), 0), 0).reshape(b, 16, l)
    v = oracle.decode16(oracle.encode16(oracle.fill(b * 16 * l, 3, b * 256 + b * 16 * l), 0), 0).reshape(b, l, 16)
    got = oracle.streaming_16x16(q, k, v, 0.25)
    qq = np.zeros((b, l, 16), np.float32)
    qq[:, :16] = q
    gen = oracle.forward(qq, oracle.transpose_k_16(k), v, scale=0.25)[:, :16]
    assert oracle.max_abs(got, gen) <= 5e-6


def test_transpose_k(oracle):
    k = np.arange(2 * 16 * 5, dtype=np.float32).reshape(2, 16, 5)
    assert np.array_equal(oracle.transpose_k_16(k), np.transpose(k, (0, 2, 1)))


def test_fp16_encoder_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    parts = [rng.standard_normal(50000).astype(np.float32) * s for s in (1.0, 1e-3, 1e-6, 3e-8, 100.0, 7e4)]
    parts.append(np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25,
                           2.0 ** -25 * 1.0001, 2.0 ** -14, np.inf, -np.inf], np.float32))
    x = np.concatenate(parts)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(oracle.encode16(x, oracle.F16), want)
    allb = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    f = allb.view(np.float16).astype(np.float32)
    got = oracle.decode16(allb, oracle.F16)
    ok = ~np.isnan(f)
    assert np.array_equal(got[ok].view(np.uint32), f[ok].view(np.uint32))
    assert np.isnan(got[~ok]).all()


def test_bf16_encoder(oracle):
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(100000) * 10).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    want = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    assert np.array_equal(oracle.encode16(x, oracle.BF16), want)
    back = oracle.decode16(want, oracle.BF16)
    assert np.array_equal(back.view(np.uint32), want.astype(np.uint32) << 16)
    assert np.abs(back - x).max() <= np.abs(x).max() * 2.0 ** -8


def test_fill_is_counter_based_and_plausible(oracle):
    a = oracle.fill(1000, 42, 0)
    b = oracle.fill(500, 42, 500)
    assert np.array_equal(a[500:], b)
    big = oracle.fill(200000, 42, 0)
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01
    u = oracle.fill(200000, 123, 0, oracle.UNIFORM)
    assert u.min() >= -1.0 and u.max() <= 1.0 and abs(float(u.mean())) < 0.01
    assert not np.array_equal(oracle.fill(10, 1), oracle.fill(10, 2))


def test_metrics(oracle):
    a = np.array([1.0, 2.0, 3.0], np.float32)
    b = np.array([1.0, 2.5, 3.0], np.float32)
    assert abs(oracle.max_abs(a, b) - 0.5) < 1e-7
    assert abs(oracle.rel_l2(a, b) - 0.5 / np.sqrt(1 + 6.25 + 9)) < 1e-6
    c = a.copy()
    c[1] = np.nan
    assert np.isnan(oracle.max_abs(c, b))
