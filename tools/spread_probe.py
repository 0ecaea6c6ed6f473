import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
import flashattention_kernel_project_amd as fa
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.empty(8, 16, 4096, 64, device="cuda", dtype=torch.float32)
q0, k0, v0 = (torch.randn(8, 16, 4096, 64, generator=g, device="cuda") for _ in range(3))
cases = {"base": (1, 1, 1), "qk1.5": (1.5, 1.5, 1), "q2.25": (2.25, 1, 1), "k2.25": (1, 2.25, 1), "v4": (1, 1, 4), "qk1.25": (1.25, 1.25, 1), "q0.5k0.5":(0.5,0.5,1), "zeroqk":(0,0,1)}
for name, (a, b, c) in cases.items():
    q, k, v = (q0 * a).half(), (k0 * b).half(), (v0 * c).half()
    for _ in range(60):
        fa.fa_forward(q, k, v, out=o, algo=24)
    res = {24: [], 23: []}
    for _ in range(4):
        for algo in (24, 23):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fa.fa_forward(q, k, v, out=o, algo=algo)
            e1.record(); torch.cuda.synchronize()
            res[algo].append(e0.elapsed_time(e1) / 20)
    print(f"{name:10s} 24: {statistics.median(res[24]):.4f}  23: {statistics.median(res[23]):.4f}", flush=True)
