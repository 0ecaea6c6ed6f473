# (run on the GPU box from the repo root) fa_forward_causal at d=128: parity of ids 24 / 28 against fp32 torch ops, then timings
python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
import flashattention_kernel_project_amd as fa
torch.manual_seed(0)
for dt in (torch.float16, torch.bfloat16):
    for (bh, n) in ((3, 1000), (2, 2048), (5, 333), (1, 4500)):
        q, k, v = (torch.randn(1, bh, n, 128, device="cuda").to(dt) for _ in range(3))
        s = (q.float() @ k.float().transpose(-1, -2)) / 128 ** 0.5
        s = s.masked_fill(~torch.ones(n, n, dtype=torch.bool, device="cuda").tril_(), float("-inf"))
        ref = torch.softmax(s, -1) @ v.float()
        for algo in (24, 28):
            o = fa.fa_forward(q, k, v, algo=algo, causal=True)
            print(dt, bh, n, "algo", algo, "max err %.2e" % float((o - ref).abs().max()))
PY
for shp in "8 16 8192" "1 32 16384" "4 16 4096"; do set -- $shp; for dt in f16 bf16; do echo "== causal B$1 H$2 N$3 d128 $dt"; python tools/ab_bench.py --causal --algos 24,28 --rounds 6 --iters 6 --B $1 --H $2 --N $3 --d 128 --dtype $dt 2>&1 | grep algo; done; done
