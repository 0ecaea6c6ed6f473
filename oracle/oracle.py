"""ctypes bindings for oracle/liboracle_cpu.so (our C restatement) and oracle/_ref/libref_cpu.so
(the reference's own `flashattn_cpu_ref`, compiled from /root/reference by build_ref.sh).

TEST INFRASTRUCTURE ONLY -- see oracle/attention_cpu.h.  numpy in, numpy out.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_cpu.so")
_REF = os.path.join(_HERE, "_ref", "libref_cpu.so")

F16, BF16 = 0, 1
NORMAL, UNIFORM = 0, 1

_fp = C.POINTER(C.c_float)
_u16p = C.POINTER(C.c_uint16)


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "attention_cpu.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_cpu.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir(os.environ.get("FA_REFERENCE_ROOT", "/root/reference")) and (
            force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.fa_oracle_forward.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        L.fa_oracle_forward_rows.argtypes = L.fa_oracle_forward.argtypes + [C.c_int] * 4
        L.fa_oracle_forward_causal_rows.argtypes = L.fa_oracle_forward_rows.argtypes
        L.fa_oracle_forward_cross.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        L.fa_oracle_streaming_16x16.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_float]
        L.fa_oracle_transpose_k_16.argtypes = [_fp, _fp, C.c_int, C.c_int]
        L.fa_oracle_fill.argtypes = [_fp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
        L.fa_oracle_round_through.argtypes = [_fp, C.c_size_t, C.c_int]
        L.fa_oracle_encode16.argtypes = [_fp, _u16p, C.c_size_t, C.c_int]
        L.fa_oracle_decode16.argtypes = [_u16p, _fp, C.c_size_t, C.c_int]
        L.fa_oracle_rel_l2.argtypes = [_fp, _fp, C.c_size_t]
        L.fa_oracle_rel_l2.restype = C.c_double
        L.fa_oracle_max_abs.argtypes = [_fp, _fp, C.c_size_t]
        L.fa_oracle_max_abs.restype = C.c_double
        _lib = L
    return _lib


def have_ref() -> bool:
    return os.path.exists(_REF)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        R.ref_flashattn_cpu_ref.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int]
        _ref = R
    return _ref


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_fp)


def fill(n: int, seed: int = 42, offset: int = 0, dist: int = NORMAL) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().fa_oracle_fill(_f(out), n, seed, offset, dist)
    return out


def make_qkv(bh: int, n: int, d: int, fmt: int = F16, seed: int = 42, dist: int = NORMAL):
    """Q,K,V [bh,n,d] drawn in the order Q->K->V from one stream and rounded through fp16/bf16.
    Returns (fp32 values, uint16 encodings)."""
    cnt = bh * n * d
    vals, bits = [], []
    for i in range(3):
        x = fill(cnt, seed, i * cnt, dist)
        u = encode16(x, fmt)
        vals.append(decode16(u, fmt).reshape(bh, n, d))
        bits.append(u.reshape(bh, n, d))
    return vals, bits


def encode16(x: np.ndarray, fmt: int) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().fa_oracle_encode16(_f(x), out.ctypes.data_as(_u16p), x.size, fmt)
    return out


def decode16(u: np.ndarray, fmt: int) -> np.ndarray:
    u = np.ascontiguousarray(u, np.uint16)
    out = np.empty(u.shape, np.float32)
    lib().fa_oracle_decode16(u.ctypes.data_as(_u16p), _f(out), u.size, fmt)
    return out


def forward(q, k, v, scale=None, accum: int = 0, nthreads: int = 1, bh_range=None, row_range=None, causal: bool = False):
    """Naive 3-loop attention forward on [BH,N,D] fp32 arrays (already 16-bit-rounded)."""
    q = np.ascontiguousarray(q, np.float32)
    k = np.ascontiguousarray(k, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    bh, n, d = q.shape
    if scale is None:
        scale = 1.0 / np.sqrt(np.float32(d))
    o = np.zeros_like(q)
    b0, b1 = bh_range if bh_range else (0, bh)
    r0, r1 = row_range if row_range else (0, n)
    fn = lib().fa_oracle_forward_causal_rows if causal else lib().fa_oracle_forward_rows
    fn(_f(q), _f(k), _f(v), _f(o), bh, n, d, float(scale), accum, nthreads, b0, b1, r0, r1)
    return o


def forward_cross(q, k, v, scale=None, accum: int = 0, nthreads: int = 1):
    """q [BH,Nq,D], k/v [BH,Nk,D] fp32 (already 16-bit-rounded) -> o [BH,Nq,D]; no mask."""
    q = np.ascontiguousarray(q, np.float32)
    k = np.ascontiguousarray(k, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    bh, nq, d = q.shape
    nk = k.shape[1]
    assert k.shape == (bh, nk, d) and v.shape == (bh, nk, d)
    if scale is None:
        scale = 1.0 / np.sqrt(np.float32(d))
    o = np.zeros_like(q)
    lib().fa_oracle_forward_cross(_f(q), _f(k), _f(v), _f(o), bh, nq, nk, d, float(scale), accum, nthreads)
    return o


def reference_forward(q, k, v):
    """The reference's own flashattn_cpu_ref (scale = 1/sqrt(D), double accumulators)."""
    q = np.ascontiguousarray(q, np.float32)
    k = np.ascontiguousarray(k, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    bh, n, d = q.shape
    o = np.zeros_like(q)
    ref().ref_flashattn_cpu_ref(_f(q), _f(k), _f(v), _f(o), bh, n, d)
    return o


def streaming_16x16(q, k, v, scale=0.25):
    """16x16 family: q [B,16,16], k [B,16,L], v [B,L,16] -> o [B,16,16] (fp32)."""
    q = np.ascontiguousarray(q, np.float32)
    k = np.ascontiguousarray(k, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    b, l = q.shape[0], v.shape[1]
    assert q.shape == (b, 16, 16) and k.shape == (b, 16, l) and v.shape == (b, l, 16)
    o = np.zeros((b, 16, 16), np.float32)
    lib().fa_oracle_streaming_16x16(_f(q), _f(k), _f(v), _f(o), b, l, float(scale))
    return o


def transpose_k_16(k):
    k = np.ascontiguousarray(k, np.float32)
    b, _, l = k.shape
    kt = np.empty((b, l, 16), np.float32)
    lib().fa_oracle_transpose_k_16(_f(k), _f(kt), b, l)
    return kt


def rel_l2(got, ref_) -> float:
    got = np.ascontiguousarray(got, np.float32)
    ref_ = np.ascontiguousarray(ref_, np.float32)
    return float(lib().fa_oracle_rel_l2(_f(got), _f(ref_), got.size))


def max_abs(got, ref_) -> float:
    got = np.ascontiguousarray(got, np.float32)
    ref_ = np.ascontiguousarray(ref_, np.float32)
    return float(lib().fa_oracle_max_abs(_f(got), _f(ref_), got.size))
