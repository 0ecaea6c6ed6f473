#!/usr/bin/env python3
"""bench.py -- attention-forward TFLOP/s of the HIP path on MI355X, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one forward launch over the whole per-GPU batch (B=8,H=16,N=4096,d=64 fp16 in,
fp32 out: the configuration BASELINE.json's metric is quoted on), inputs resident in HBM.
For N>1 the driver starts one rank per GPU (torch.distributed.run); every rank runs the same
per-GPU workload on its own (b,h) shard of a global batch of 8*N (weak scaling, no data-path
collective -- SURVEY.md 8(e)); the barrier-bracketed wall time is max-reduced over ranks.

Rank 0 prints ONE JSON line with the contract keys plus `roofline` (dominant kernel, HIP events
on the launch stream) and `cpu_baseline` (the oracle's naive 3-loop fp32 port, timed on this
host's cores over a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (2.5 PF)
CONFIGS = {"cfg3": (4, 8, 1024, 64, "f16"), "cfg4": (8, 16, 4096, 64, "f16"), "cfg4bf16": (8, 16, 4096, 64, "bf16"),
           "cfg5": (8, 16, 8192, 128, "f16")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--dtype", choices=["f16", "bf16"], default="f16")
    ap.add_argument("--out", choices=["f32", "same"], default="f32")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json shorthand: cfg3 (B4 H8 N1024 d64 fp16), cfg4 (B8 H16 N4096 d64 fp16, the metric config), "
                         "cfg4bf16 (config 4 as written), cfg5 (one GPU's shard of B64 H16 N8192 d128 fp16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sustained", type=int, default=300,
                    help="launches run AFTER the timed region and reported under `sustained` (the first ~30 ms after idle run "
                         "slower while the clocks settle; never part of `value`); 0 disables")
    args = ap.parse_args()
    if args.config:
        args.B, args.H, args.N, args.d, args.dtype = CONFIGS[args.config]

    import torch
    import flashattention_kernel_project_amd as fa
    from flashattention_kernel_project_amd.dist import Ranks, timed_region

    fa.lib()   # fail loudly if the HIP library is missing: there is no fallback path
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # FA_BENCH_BACKEND=gloo + FA_BENCH_ONE_DEVICE=1: rehearsal of the N>1 path on a one-GPU box
    # (all ranks on cuda:0, gloo carries the barrier / max reduction).  The driver's runs use RCCL.
    backend = os.environ.get("FA_BENCH_BACKEND", "nccl") if world > 1 else None
    local = 0 if os.environ.get("FA_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ranks = Ranks(backend=backend)

    B, H, N, d = args.B, args.H, args.N, args.d
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    odt = torch.float32 if args.out == "f32" else dt
    # this rank's shard of the global (B*world) x H problem: B x H heads, generated in place
    g = torch.Generator(device=dev).manual_seed(42 + ranks.rank)
    q, k, v = (torch.randn(B, H, N, d, generator=g, device=dev, dtype=torch.float32).to(dt) for _ in range(3))
    o = torch.empty(B, H, N, d, device=dev, dtype=odt)
    stream = torch.cuda.current_stream()

    kernel_name = fa.lib().fa_selected_kernel(B, H, N, d, 0 if args.dtype == "f16" else 1, 0).decode()

    def step():
        fa.fa_forward(q, k, v, out=o, stream=stream)

    for _ in range(args.warmup):
        step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run_steps():
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)

    wall = timed_region(ranks, run_steps, torch.cuda.synchronize)
    wall = ranks.max_over_ranks(wall, dev)
    kern_own = ev0.elapsed_time(ev1) / args.steps         # avg launch duration, events on the launch stream
    kern_ms = ranks.max_over_ranks(kern_own, dev)
    per_rank_ms = ranks.gather(kern_own, dev)             # "per-GPU and aggregate": every rank's own figure, rank order

    flops_per_gpu = fa.attention_flops(B * H, N, d)
    total_flops = flops_per_gpu * world * args.steps
    value = total_flops / wall / 1e12
    achieved = flops_per_gpu / (kern_ms * 1e-3) / 1e12

    # after the timed region, outside `value`: the same launch back to back until the clocks have settled
    sustained = None
    if args.sustained > 0:
        sv0, sv1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(args.sustained // 3):
            step()
        sv0.record(stream)
        for _ in range(args.sustained):
            step()
        sv1.record(stream)
        torch.cuda.synchronize()
        s_ms = ranks.max_over_ranks(sv0.elapsed_time(sv1) / args.sustained, dev)
        sustained = {"launches": args.sustained, "after_launches": args.warmup + args.steps + args.sustained // 3,
                     "ms_per_step": round(s_ms, 5), "tflops_per_gpu": round(flops_per_gpu / (s_ms * 1e-3) / 1e12, 3),
                     "frac": round(flops_per_gpu / (s_ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4)}
    # N > 1: BASELINE config 5's per-GPU shard (B8 H16 N8192 d128) as an extra figure, a few launches
    cfg5 = None
    if world > 1 and (B, H, N, d) != (8, 16, 8192, 128):
        q5, k5, v5 = (torch.randn(8, 16, 8192, 128, generator=g, device=dev, dtype=torch.float32).to(dt) for _ in range(3))
        o5 = torch.empty(8, 16, 8192, 128, device=dev, dtype=odt)
        # (run behind the sustained block, 12 + 30 launches of 3.7 ms: the post-idle clock clamp -- the first ~30 ms of a
        # sudden load, profiles/r02_transient.txt -- is over before the timed launches start)
        n5w, n5 = 12, 30
        for _ in range(n5w):
            fa.fa_forward(q5, k5, v5, out=o5, stream=stream)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(n5):
            fa.fa_forward(q5, k5, v5, out=o5, stream=stream)
        c1.record(stream)
        torch.cuda.synchronize()
        c_own = c0.elapsed_time(c1) / n5
        c_ms = ranks.max_over_ranks(c_own, dev)
        f5 = fa.attention_flops(128, 8192, 128)
        cfg5 = {"workload": "B=8 H=16 N=8192 d=128 per GPU (config 5 shard)", "launches": n5, "after_launches": n5w,
                "ms_per_step": round(c_ms, 4),
                "tflops_per_gpu": round(f5 / (c_ms * 1e-3) / 1e12, 2), "tflops_all_gpus": round(world * f5 / (c_ms * 1e-3) / 1e12, 2),
                "per_rank_tflops": [round(f5 / (m * 1e-3) / 1e12, 2) for m in ranks.gather(c_own, dev)]}
        del q5, k5, v5, o5

    cpu = None
    if ranks.rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(q, k, v, o, N, d, args.cpu_seconds)

    if ranks.rank == 0:
        line = {
            "metric": f"attention-fwd TFLOP/s (and %MFMA-peak) at B{B} H{H} N{N} d{d} {'fp16' if args.dtype == 'f16' else 'bf16'}",
            "value": round(value, 3),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic N(0,1) Q/K/V rounded to %s, resident in HBM" % args.dtype,
            "config": {"workload": f"attention forward B={B} H={H} N={N} d={d} per GPU, {args.dtype} in / "
                                   f"{'fp32' if odt == torch.float32 else args.dtype} out, non-causal, "
                                   f"global batch {B * world} sharded over (b,h)",
                       "B_per_gpu": B, "H": H, "N": N, "d": d, "flops_per_step_per_gpu": flops_per_gpu},
            "pct_mfma_peak": round(100.0 * value / (PEAK_TFLOPS * world), 2),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_TFLOPS, 4),
                         "traffic": measured_traffic(B, H, N, d, args.dtype, args.out, kernel_name),
                         "kernel": kernel_name, "avg_launch_ms": round(kern_ms, 5),
                         "algorithmic_bytes": fa.attention_min_bytes(B * H, N, d, 2, 4 if odt == torch.float32 else 2),
                         "hbm_GBps_algorithmic": round(fa.attention_min_bytes(B * H, N, d, 2, 4 if odt == torch.float32 else 2)
                                                       / (kern_ms * 1e-3) / 1e9, 1)},
            "cpu_baseline": cpu,
            "sustained": sustained,
        }
        line["per_rank_tflops"] = [round(flops_per_gpu / (m * 1e-3) / 1e12, 3) for m in per_rank_ms]
        if cfg5 is not None:
            line["cfg5_per_gpu"] = cfg5
        print(json.dumps(line), flush=True)
    ranks.close()


def cpu_threads():
    """Threads for the CPU baseline: this process's CPU share, capped at 16 (one GPU's share of the box)."""
    env = os.environ.get("FA_CPU_THREADS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(q, k, v, o_gpu, N, d, budget_s):
    """The oracle (naive 3-loop fp32 C port of the reference's CPU function) on a bounded sample
    of the same workload: whole heads (all N query rows x all N keys) starting at (b=0,h=0), or a
    row range of head 0 when one head exceeds the budget.  Also cross-checks the GPU output on
    that sample (the oracle is the checker here, never the measured path)."""
    from oracle import oracle as orc
    threads = cpu_threads()
    H = q.shape[1]
    flops_row = 4.0 * N * d
    q0, k0, v0 = (t[0, 0].float().cpu().numpy()[None] for t in (q, k, v))
    cal_rows = min(N, 8 * threads)
    t0 = time.perf_counter()
    orc.forward(q0, k0, v0, accum=0, nthreads=threads, row_range=(0, cal_rows))
    rate = cal_rows / max(time.perf_counter() - t0, 1e-5)          # rows / s
    heads = int(budget_s * rate / N)
    if heads >= 1:
        heads = min(heads, q.shape[0] * H)
        idx = [(i // H, i % H) for i in range(heads)]
        qs, ks, vs = (torch_stack(t, idx) for t in (q, k, v))
        t0 = time.perf_counter()
        want = orc.forward(qs, ks, vs, accum=0, nthreads=threads)
        dt_ = time.perf_counter() - t0
        got = torch_stack(o_gpu, idx)
        rows = heads * N
        what = f"{heads} whole (b,h) heads x {N} query rows x {N} keys"
    else:
        rows = max(threads, int(budget_s * rate))
        t0 = time.perf_counter()
        want = orc.forward(q0, k0, v0, accum=0, nthreads=threads, row_range=(0, rows))[:, :rows]
        dt_ = time.perf_counter() - t0
        got = o_gpu[0, 0, :rows].float().cpu().numpy()[None]
        what = f"{rows} query rows x {N} keys of head (b=0,h=0)"
    err = orc.max_abs(got, want)
    # the same port on ONE thread (SURVEY 8(d) asks for both), on a row range sized for ~2 s
    rows1 = max(8, min(N, int(2.0 * rate / max(threads, 1))))
    t0 = time.perf_counter()
    orc.forward(q0, k0, v0, accum=0, nthreads=1, row_range=(0, rows1))
    dt1 = time.perf_counter() - t0
    # the reference's OWN CPU function (oracle/_ref: flashattn_cpu_ref of flashattn_forward_fused_5_4_2.cu:224-272, compiled
    # from /root/reference in the build container; single-threaded, double accumulators) on a prefix of head (b=0,h=0) sized
    # for a few seconds: it takes whole [BH,N,D] problems, so the sample is that head cut to its first n_ref rows AND keys
    reference = None
    if orc.have_ref():
        n_ref = 2048 if N >= 2048 else N
        qr, kr, vr = (np_c(t[:, :n_ref]) for t in (q0, k0, v0))
        t0 = time.perf_counter()
        o_ref = orc.reference_forward(qr, kr, vr)
        dtr = time.perf_counter() - t0
        o_port = orc.forward(qr, kr, vr, accum=1, nthreads=threads)
        reference = {"value": round(4.0 * n_ref * n_ref * d / dtr / 1e12, 6), "unit": "TFLOP/s", "cores": 1, "kind": "reference",
                     "sample": f"flashattn_cpu_ref on head (b=0,h=0) cut to {n_ref} rows x {n_ref} keys, d={d}: "
                               f"{4.0 * n_ref * n_ref * d / 1e9:.2f} GFLOP in {dtr:.2f} s on 1 thread",
                     "port_vs_reference_max_abs": float(orc.max_abs(o_port, o_ref))}
    return {"value": round(rows * flops_row / dt_ / 1e12, 6), "unit": "TFLOP/s", "cores": threads, "kind": "port",
            "reference": reference,
            "single_thread": {"value": round(rows1 * flops_row / dt1 / 1e12, 6), "unit": "TFLOP/s", "cores": 1,
                              "sample": f"{rows1} query rows x {N} keys of head (b=0,h=0) in {dt1:.2f} s"},
            "sample": f"{what}, d={d}: {rows * flops_row / 1e9:.1f} GFLOP in {dt_:.2f} s on {threads} threads "
                      f"(oracle/attention_cpu.c, naive 3-loop fp32, OpenMP over rows)",
            "gpu_vs_cpu_max_abs_on_sample": float(err)}


def np_c(a):
    import numpy as np
    return np.ascontiguousarray(a, dtype=np.float32)


def torch_stack(t, idx):
    import numpy as np
    return np.stack([t[b, h].float().cpu().numpy() for (b, h) in idx])


def measured_traffic(B, H, N, d, dtype, out, kernel_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json,
    written by tools/collect_profiles.sh with the gfx950 FETCH_SIZE x2 correction), if they were
    taken on this exact workload AND on the kernel this run dispatches; else None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        cfg = rec.get("config", {})
        same = (cfg.get("B"), cfg.get("H"), cfg.get("N"), cfg.get("d"), cfg.get("dtype"), cfg.get("out")) == (B, H, N, d, dtype, out)
        if same and kernel_name and str(rec.get("kernel", "")).replace("void ", "").startswith(kernel_name):
            return rec.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


if __name__ == "__main__":
    main()
