"""ctypes binding of include/fa_mi355.h (libfa_mi355.so, built in-tree by csrc/Makefile)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FA_MI355_LIB") or os.path.join(_HERE, "libfa_mi355.so")   # override: A/B of experimental builds
CSRC = os.path.join(_HERE, "csrc")

F16, BF16 = 0, 1
OUT_F32, OUT_SAME = 0, 1
ALGO_AUTO, ALGO_GENERIC, ALGO_TILED, ALGO_INTERLEAVED, ALGO_INTERLEAVED_2WG = 0, 1, 2, 5, 6
ALGO_W64, ALGO_W64P, ALGO_W64X = 13, 14, 16
ALGO_SK, ALGO_RP, ALGO_RP_FOLD, ALGO_RP16, ALGO_RP16_FOLD = 17, 21, 22, 23, 24
ALGO_RP16_FOLD_HALF, ALGO_RP16_FOLD_QUARTER, ALGO_RP16_FOLD_1W, ALGO_RP16_FOLD_KS2 = 26, 27, 28, 29

# every symbol include/fa_mi355.h declares
SYMBOLS = (
    "flashattn_forward_wmma",
    "fa_forward",
    "fa_forward_ex",
    "fa_forward_causal",
    "fa_forward_splitkv_workspace_bytes",
    "fa_forward_splitkv",
    "fa_debug_stage",
    "flashattn_streaming_16x16_mw",
    "flashattn_streaming_16x16_mw_kt",
    "fa_mi355_version",
    "fa_mi355_has_experiments",
    "fa_selected_algo",
    "fa_selected_kernel",
)


class FaError(RuntimeError):
    """A launcher returned a non-zero hipError_t."""

    def __init__(self, fn: str, code: int):
        super().__init__(f"{fn} failed with hipError_t {code}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile libfa_mi355.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _share_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64 (soname libamdhip64.so.7).  Device
    pointers and stream handles are only meaningful inside ONE HIP runtime, so that copy must be
    in the process before libfa_mi355.so is loaded: its DT_NEEDED libamdhip64.so.7 then binds to
    it by soname instead of pulling /opt/rocm's copy in as a second runtime (which sees no device:
    hipErrorNoDevice).  A pure C/C++ host (bench/fa_bench) links /opt/rocm's runtime directly."""
    try:
        import torch
    except ImportError:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Load the HIP library.  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C flashattention_kernel_project_amd/csrc`")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.flashattn_forward_wmma.argtypes = [vp, vp, vp, vp, i, i, i, f, vp]
        L.fa_forward.argtypes = [vp, vp, vp, vp, i, i, i, i, f, i, i, vp]
        L.fa_forward_ex.argtypes = [vp, vp, vp, vp, i, i, i, i, f, i, i, i, vp]
        L.fa_forward_causal.argtypes = [vp, vp, vp, vp, i, i, i, i, f, i, i, i, vp]
        L.fa_forward_splitkv.argtypes = [vp, vp, vp, vp, i, i, i, i, i, f, i, i, vp, C.c_size_t, vp]
        L.fa_forward_splitkv_workspace_bytes.argtypes = [i, i, i, i, i]
        L.fa_debug_stage.argtypes = [i, vp, vp, vp, i, i, i, f, i, vp]
        L.flashattn_streaming_16x16_mw.argtypes = [vp, vp, vp, vp, i, i, f, vp]
        L.flashattn_streaming_16x16_mw_kt.argtypes = [vp, vp, vp, vp, i, i, f, vp]
        for s in SYMBOLS[:9]:
            getattr(L, s).restype = C.c_int
        L.fa_mi355_has_experiments.argtypes = []
        L.fa_mi355_has_experiments.restype = C.c_int
        L.fa_selected_algo.argtypes = [i, i, i, i, i]
        L.fa_selected_algo.restype = C.c_int
        L.fa_selected_kernel.argtypes = [i, i, i, i, i, i]
        L.fa_selected_kernel.restype = C.c_char_p
        L.fa_forward_splitkv_workspace_bytes.restype = C.c_size_t
        L.fa_mi355_version.argtypes = []
        L.fa_mi355_version.restype = C.c_char_p
        _lib = L
    return _lib


def version() -> str:
    return lib().fa_mi355_version().decode()


def check(fn: str, code: int) -> None:
    if code != 0:
        raise FaError(fn, code)
