"""One-process-per-GPU plumbing for bench.py: rank discovery, barrier-bracketed timing and the
max-over-ranks reduction.  The attention path itself has no collective: (b,h) problems are
independent and are sharded contiguously over ranks (shard.py); RCCL ("nccl" backend on ROCm)
is used only for the timing barrier / reduction, gloo in the CPU tests.
"""
from __future__ import annotations

import os
import time


class Ranks:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self._pg = False
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl":
                    import torch
                    kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
                dist.init_process_group(backend=backend or "gloo", rank=self.rank, world_size=self.world, **kw)
                self._pg = True

    def barrier(self) -> None:
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(self, value: float, device=None) -> float:
        if self.world == 1:
            return value
        import torch
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64, device=device if self.backend == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float, device=None) -> float:
        if self.world == 1:
            return value
        import torch
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64, device=device if self.backend == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, value: float, device=None) -> list:
        """Every rank's value, in rank order, on every rank (the per-GPU figures of the bench line)."""
        if self.world == 1:
            return [value]
        import torch
        import torch.distributed as dist
        mine = torch.tensor([value], dtype=torch.float64, device=device if self.backend == "nccl" else None)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        return [float(t.item()) for t in out]

    def close(self) -> None:
        if self._pg:
            import torch.distributed as dist
            dist.destroy_process_group()
            self._pg = False


def timed_region(ranks: Ranks, run_steps, sync) -> float:
    """barrier + sync | run_steps() | sync + barrier; returns this rank's wall seconds."""
    sync()
    ranks.barrier()
    t0 = time.perf_counter()
    run_steps()
    sync()
    t1 = time.perf_counter()
    ranks.barrier()
    return t1 - t0
