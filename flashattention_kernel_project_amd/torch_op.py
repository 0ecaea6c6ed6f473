"""torch.ops.fa_mi355.forward -- the C-ABI forward as a PyTorch custom op.

Not part of the reference (it has no Python); a convenience for callers that live in PyTorch
graphs.  The op body is the same ctypes call as ops.fa_forward: HIP library or an exception, no
eager fallback.  Registered on first use:

    from flashattention_kernel_project_amd.torch_op import register
    register()
    o = torch.ops.fa_mi355.forward(q, k, v, scale, causal, out_fp32)
"""
from __future__ import annotations

import math

_registered = False


def register() -> None:
    """Define torch.ops.fa_mi355.forward (idempotent)."""
    global _registered
    if _registered:
        return
    import torch
    from . import ops

    @torch.library.custom_op("fa_mi355::forward", mutates_args=(), device_types="cuda",
                             schema="(Tensor q, Tensor k, Tensor v, float scale, bool causal, bool out_fp32) -> Tensor")
    def forward(q, k, v, scale, causal, out_fp32):
        stream = torch.cuda.current_stream(q.device)
        return ops.fa_forward(q.contiguous(), k.contiguous(), v.contiguous(), scale=scale,
                              out_dtype=torch.float32 if out_fp32 else q.dtype, causal=causal, stream=stream)

    @forward.register_fake
    def _(q, k, v, scale, causal, out_fp32):
        return q.new_empty(q.shape, dtype=torch.float32 if out_fp32 else q.dtype)

    _registered = True


def sdpa_like(q, k, v, is_causal: bool = False, scale: float | None = None):
    """Same call shape as torch.nn.functional.scaled_dot_product_attention for [B,H,N,d] fp16/bf16
    self-attention without dropout or an explicit mask; output in the input dtype."""
    import torch
    register()
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    return torch.ops.fa_mi355.forward(q, k, v, float(scale), bool(is_causal), False)
