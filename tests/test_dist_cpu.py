"""CPU tier: the N>1 plumbing of bench.py (rank discovery, (b,h) sharding, barrier-bracketed timing,
max-over-ranks reduction) with world_size 2 over gloo.  Each rank computes its shard of a small
attention problem with the CPU oracle (test infrastructure standing in for the GPU launch) and the
gathered shards must equal the single-process result: the path has no data collective, only the
timing reduction."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import time
    from flashattention_kernel_project_amd.dist import Ranks, timed_region
    from flashattention_kernel_project_amd.shard import shard_range
    from oracle import oracle as o

    ranks = Ranks(backend="gloo")
    assert ranks.world == world and ranks.rank == rank
    bh, n, d = 7, 48, 32                      # 7 heads over 2 ranks: uneven shards 4 + 3
    (q, k, v), _ = o.make_qkv(bh, n, d, o.F16, seed=9)
    b0, b1 = shard_range(bh, rank, world)
    result = {}

    def run_steps():
        result["o"] = o.forward(q[b0:b1], k[b0:b1], v[b0:b1])
        time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one

    wall = timed_region(ranks, run_steps, lambda: None)
    worst = ranks.max_over_ranks(wall)
    total = ranks.sum_over_ranks(float(b1 - b0))
    assert worst >= wall - 1e-9 and worst >= 0.1 - 1e-3     # everybody sees the slowest rank's time
    assert total == bh
    per_rank = ranks.gather(float(b1 - b0))                 # the bench line's per-GPU list: every rank's own figure, rank order
    assert per_rank == [4.0, 3.0]
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), result["o"])
    np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array([b0, b1]))
    ranks.close()


def test_two_rank_gloo_sharding(tmp_path, oracle):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    (q, k, v), _ = oracle.make_qkv(7, 48, 32, oracle.F16, seed=9)
    want = oracle.forward(q, k, v)
    parts, ranges = [], []
    for r in range(2):
        parts.append(np.load(tmp_path / f"shard{r}.npy"))
        ranges.append(tuple(np.load(tmp_path / f"range{r}.npy")))
    assert ranges == [(0, 4), (4, 7)]
    assert np.array_equal(np.concatenate(parts), want)      # shards tile the batch exactly, bit for bit


def test_single_rank_needs_no_process_group(fa):
    from flashattention_kernel_project_amd.dist import Ranks, timed_region
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k_, None)
    r = Ranks()
    assert (r.rank, r.world) == (0, 1)
    assert r.max_over_ranks(1.5) == 1.5
    assert r.gather(2.5) == [2.5]
    t = timed_region(r, lambda: None, lambda: None)
    assert 0 <= t < 0.5
