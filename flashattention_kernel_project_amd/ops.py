"""torch-tensor front end over the C ABI, keeping the reference's argument lists.

PyTorch is plumbing only (device memory + streams); every op goes through libfa_mi355.so.
"""
from __future__ import annotations

import math

from . import capi


def attention_flops(bh: int, n: int, d: int) -> float:
    """4*BH*N^2*D (QK^T + PV, non-causal) -- the reference's own count,
    FlashAttention/flashattn_forward_memory_bound/flashattn_forward_wmma_memprofile.cu:508."""
    return 4.0 * bh * n * n * d


def attention_min_bytes(bh: int, n: int, d: int, in_bytes: int = 2, out_bytes: int = 4) -> float:
    """3*BH*N*D*sizeof(in) + BH*N*D*sizeof(out): memprofile.cu:518-520."""
    return 3.0 * bh * n * d * in_bytes + 1.0 * bh * n * d * out_bytes


def _stream_ptr(stream):
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    return getattr(stream, "cuda_stream", stream)


def _one_device(*tensors):
    """All tensors on one GPU; returns it.  The C side launches on (and reads the CU count / sets the LDS
    attribute of) the CURRENT device, so every entry point below makes the tensors' device current."""
    dev = tensors[0].device
    for t in tensors[1:]:
        if t is not None and t.device != dev:
            raise ValueError("all tensors of one call must live on one device")
    return dev


def _dev_ptr(t, name: str, dtypes):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if t.dtype not in dtypes:
        raise ValueError(f"{name} has dtype {t.dtype}, expected one of {dtypes}")
    return t.data_ptr()


def flashattn_forward_wmma(Q, K, V, O, BH: int, N: int, D: int, scale: float, stream=None) -> None:
    """(Q,K,V,O,BH,N,D,scale) of flashattn_forward_wmma_kernel
    (FlashAttention/flashattn_forward_wmma/flashattn_forward_wmma.cu:49-58).
    Q,K,V: fp16 [BH,N,D]; O: fp32 [BH,N,D], fully overwritten."""
    import torch
    for name, t in (("Q", Q), ("K", K), ("V", V)):
        if t.numel() != BH * N * D:
            raise ValueError(f"{name} has {t.numel()} elements, expected BH*N*D = {BH * N * D}")
    if O.numel() != BH * N * D:
        raise ValueError("O has the wrong number of elements")
    ptrs = (_dev_ptr(Q, "Q", (torch.float16,)), _dev_ptr(K, "K", (torch.float16,)),
            _dev_ptr(V, "V", (torch.float16,)), _dev_ptr(O, "O", (torch.float32,)))
    with torch.cuda.device(_one_device(Q, K, V, O)):
        code = capi.lib().flashattn_forward_wmma(*ptrs, BH, N, D, float(scale), _stream_ptr(stream))
    capi.check("flashattn_forward_wmma", code)


def fa_forward(q, k, v, scale: float | None = None, out_dtype=None, algo: int = capi.ALGO_AUTO,
               out=None, stream=None, causal: bool = False):
    """Attention forward on [B,H,N,d] (or [BH,N,d]) fp16/bf16 device tensors.
    out_dtype: torch.float32 (the reference's output type, default) or the input dtype.
    causal: query row i attends to keys 0..i (fa_forward_causal; algo AUTO / GENERIC / TILED / RP16_FOLD, RP16_FOLD_1W at d=128)."""
    import torch
    if q.dim() == 3:
        B, (H, N, d) = 1, q.shape
    elif q.dim() == 4:
        B, H, N, d = q.shape
    else:
        raise ValueError("q must be [B,H,N,d] or [BH,N,d]")
    if k.shape != q.shape or v.shape != q.shape:
        raise ValueError("q, k, v must have identical shapes (self-attention, Nq == Nk)")
    if q.dtype not in (torch.float16, torch.bfloat16) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError("q, k, v must all be fp16 or all bf16")
    in_dt = capi.F16 if q.dtype == torch.float16 else capi.BF16
    if out_dtype is None:
        out_dtype = torch.float32 if out is None else out.dtype
    if out_dtype == torch.float32:
        out_dt = capi.OUT_F32
    elif out_dtype == q.dtype:
        out_dt = capi.OUT_SAME
    else:
        raise ValueError("out_dtype must be torch.float32 or the input dtype")
    if out is None:
        out = torch.empty(q.shape, dtype=out_dtype, device=q.device)
    elif out.shape != q.shape or out.dtype != out_dtype:
        raise ValueError("out has the wrong shape or dtype")
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    dts = (torch.float16, torch.bfloat16)
    fn_name = "fa_forward_causal" if causal else "fa_forward_ex"
    with torch.cuda.device(_one_device(q, k, v, out)):   # the launch goes to the CURRENT device: make that the tensors' device
        code = getattr(capi.lib(), fn_name)(
            _dev_ptr(q, "q", dts), _dev_ptr(k, "k", dts), _dev_ptr(v, "v", dts),
            _dev_ptr(out, "out", (out_dtype,)), B, H, N, d, float(scale), in_dt, out_dt, algo,
            _stream_ptr(stream))
    capi.check(fn_name, code)
    return out


def fa_forward_splitkv(q, k, v, scale: float | None = None, out_dtype=None, workspace=None, stream=None):
    """q [B,Hq,Nq,d], k/v [B,Hkv,Nk,d] fp16/bf16 device tensors, no mask, d in {64,128}: the key axis is
    cut into chunks that run in parallel and are merged (fa_forward_splitkv, "flash-decoding").
    Hq may be a multiple of Hkv (grouped-query attention): the G = Hq/Hkv query heads of a group are
    contiguous in q, so the group is passed to the C ABI as ONE head with G*Nq query rows and its K/V is
    streamed once for all of them -- no copy, same kernel.
    workspace: optional uint8 device tensor of at least splitkv_workspace_bytes(B, Hkv, G*Nq, Nk, d)."""
    import torch
    if q.dim() != 4 or k.dim() != 4 or v.shape != k.shape or q.shape[0] != k.shape[0] or q.shape[3] != k.shape[3]:
        raise ValueError("q must be [B,Hq,Nq,d] and k, v [B,Hkv,Nk,d]")
    B, Hq, Nq, d = q.shape
    H, Nk = k.shape[1], k.shape[2]
    if Hq % H != 0:
        raise ValueError("the number of query heads must be a multiple of the number of K/V heads")
    rows = (Hq // H) * Nq   # query rows that share one K/V head
    if q.dtype not in (torch.float16, torch.bfloat16) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError("q, k, v must all be fp16 or all bf16")
    in_dt = capi.F16 if q.dtype == torch.float16 else capi.BF16
    out_dtype = out_dtype or torch.float32
    if out_dtype not in (torch.float32, q.dtype):
        raise ValueError("out_dtype must be torch.float32 or the input dtype")
    out = torch.empty(q.shape, dtype=out_dtype, device=q.device)
    with torch.cuda.device(q.device):   # the split count follows the CU count of the device that will run it
        need = splitkv_workspace_bytes(B, H, rows, Nk, d)
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
    ws_ptr, ws_len = (workspace.data_ptr(), workspace.numel() * workspace.element_size()) if workspace is not None else (None, 0)
    if scale is None:
        scale = 1.0 / math.sqrt(d)
    dts = (torch.float16, torch.bfloat16)
    ptrs = (_dev_ptr(q, "q", dts), _dev_ptr(k, "k", dts), _dev_ptr(v, "v", dts), _dev_ptr(out, "out", (out_dtype,)))
    with torch.cuda.device(_one_device(q, k, v, out, workspace)):
        code = capi.lib().fa_forward_splitkv(
            *ptrs, B, H, rows, Nk, d, float(scale), in_dt, capi.OUT_F32 if out_dtype == torch.float32 else capi.OUT_SAME,
            ws_ptr, ws_len, _stream_ptr(stream))
    capi.check("fa_forward_splitkv", code)
    return out


def splitkv_workspace_bytes(B: int, H: int, Nq: int, Nk: int, d: int) -> int:
    return int(capi.lib().fa_forward_splitkv_workspace_bytes(B, H, Nq, Nk, d))


def _streaming(fn_name: str, Q, K, V, O, num_batches: int, seq_len: int, scale: float, stream):
    import torch
    if Q.numel() != num_batches * 256 or K.numel() != num_batches * 16 * seq_len \
            or V.numel() != num_batches * 16 * seq_len or O.numel() != num_batches * 256:
        raise ValueError("tensor sizes do not match num_batches/seq_len")
    ptrs = (_dev_ptr(Q, "Q", (torch.float16,)), _dev_ptr(K, "K", (torch.float16,)),
            _dev_ptr(V, "V", (torch.float16,)), _dev_ptr(O, "O", (torch.float32,)))
    with torch.cuda.device(_one_device(Q, K, V, O)):
        code = getattr(capi.lib(), fn_name)(*ptrs, num_batches, seq_len, float(scale), _stream_ptr(stream))
    capi.check(fn_name, code)


def flashattn_streaming_16x16_mw(Q, K, V, O, num_batches: int, seq_len: int, scale: float, stream=None) -> None:
    """(Q,K,V,O,num_batches,seq_len,scale) of flashattn_streaming_16x16_kernel_mw
    (Streaming_FlashAttention_Forward_Kernel/flashattn_streaming_16x16_mw.cu:73-81).
    Q [B,16,16], K [B,16,L], V [B,L,16] fp16; O [B,16,16] fp32."""
    _streaming("flashattn_streaming_16x16_mw", Q, K, V, O, num_batches, seq_len, scale, stream)


def flashattn_streaming_16x16_mw_kt(Q, K_T, V, O, num_batches: int, seq_len: int, scale: float, stream=None) -> None:
    """v8+ ABI: K_T [B,L,16] (flashattn_warp_spc/flashattn_streaming_16x16_mw_v8.cu:103-111)."""
    _streaming("flashattn_streaming_16x16_mw_kt", Q, K_T, V, O, num_batches, seq_len, scale, stream)
