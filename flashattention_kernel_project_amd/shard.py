"""Batch x head sharding across the GPUs of a node.

Every (b,h) pair is an independent attention problem (grid.y = BH in
FlashAttention/flashattn_forward_wmma/flashattn_forward_wmma.cu:109,390), so the flattened BH
axis is split contiguously over ranks and there is no data-path collective.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) share of `total` items for `rank` of `world`; sizes differ by <= 1."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    q, r = divmod(total, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)
