#!/usr/bin/env python3
"""What the folded fast pass costs when its gates refuse the data: ms per launch of FA_ALGO_RP16_FOLD (24) and of the exact
pipeline (23) at B8 H16 N4096 d64 for inputs of growing spread (the logits grow with the square of it).  Interleaved
rounds after a warm-up (the first ~30 ms after idle run slower)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import flashattention_kernel_project_amd as fa  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
o = torch.empty(8, 16, 4096, 64, device="cuda", dtype=torch.float32)
for dt in (torch.float16, torch.bfloat16):
    for spread in (1.0, 1.5, 2.0, 3.0, 4.0):
        q, k, v = (torch.randn(8, 16, 4096, 64, generator=g, device="cuda") for _ in range(3))
        q, k, v = (q * spread).to(dt), (k * spread).to(dt), v.to(dt)
        for _ in range(60):
            fa.fa_forward(q, k, v, out=o, algo=24)
        res = {24: [], 23: []}
        for _ in range(4):
            for algo in (24, 23):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fa.fa_forward(q, k, v, out=o, algo=algo)
                e1.record()
                torch.cuda.synchronize()
                res[algo].append(e0.elapsed_time(e1) / 20)
        mx = float((q[0, 0].float() @ k[0, 0].float().T).abs().max()) * 0.125 * 1.4427
        print(f"{str(dt):15s} spread {spread:3.1f}  max |logit| {mx:6.1f} log2 units   folded-first (24) {statistics.median(res[24]):.4f} ms   exact (23) {statistics.median(res[23]):.4f} ms", flush=True)
