"""flashattention_kernel_project_amd -- MI355X-native FlashAttention forward behind the
launch signatures of jeehun98/FlashAttention_Kernel_Project.

The product is `libfa_mi355.so` (hand-written HIP for gfx950, C ABI in include/fa_mi355.h).
This package is the Python host side: a ctypes binding of that ABI (`capi`), a torch-tensor
front end with the reference's argument lists (`ops`), and the B x H sharding helper used for
multi-GPU runs (`shard`).  There is no CPU or PyTorch fallback: if the HIP library is missing
or fails to load, every op raises.
"""
from .capi import build, lib, version, FaError  # noqa: F401
from .ops import (  # noqa: F401
    fa_forward,
    fa_forward_splitkv,
    splitkv_workspace_bytes,
    flashattn_forward_wmma,
    flashattn_streaming_16x16_mw,
    flashattn_streaming_16x16_mw_kt,
    attention_flops,
    attention_min_bytes,
)
from .shard import shard_range  # noqa: F401

__all__ = [
    "build", "lib", "version", "FaError",
    "fa_forward", "fa_forward_splitkv", "splitkv_workspace_bytes", "flashattn_forward_wmma",
    "flashattn_streaming_16x16_mw", "flashattn_streaming_16x16_mw_kt",
    "attention_flops", "attention_min_bytes", "shard_range",
]
