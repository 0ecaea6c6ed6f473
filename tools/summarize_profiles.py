#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh) into the committed
profiles/<tag>_* files: the rocprofv3 kernel stats, per-kernel PMC averages with derived MFMA
utilisation, and the corrected HBM traffic per launch (profiles/hbm_traffic.json, read by bench.py).

    python tools/summarize_profiles.py r01 [--B 8 --H 16 --N 4096 --d 64 --dtype f16 --out f32]
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None   # newest (gpurun merges runs into one tree)


def is_redo(name):
    """The redo kernel of the full-width pipeline (fa_fwd_rp16_kernel<..., kScan = true, 8>): launched behind every forward of
    those shapes, ends after one look at the marker words unless a row block was left to it.  Reported separately."""
    if "fa_fwd_rp16_kernel<" not in name:
        return False
    args = [a.strip() for a in name.split("fa_fwd_rp16_kernel<", 1)[1].split(">", 1)[0].split(",")]
    return len(args) > 7 and args[7] == "true"   # <T, D, X, out, fold, dma, causal, kScan, ...>


def pmc_avgs(d, kernel_substr="fa_fwd", redo=False):
    f = one(os.path.join(d, "**", "*_counter_collection.csv"))
    if not f:
        return {}
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"] and is_redo(r["Kernel_Name"]) == redo:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                      "Grid_Size", "Workgroup_Size", "Kernel_Name") if k in r}
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    out["_dispatches"] = max((len(v) for v in acc.values()), default=0)
    out["_meta"] = meta
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--out", default="f32")
    ap.add_argument("--config", default=None, help="cfg3 | cfg4 | cfg4bf16 | cfg5 (sets B, H, N, d, dtype)")
    a = ap.parse_args()
    if a.config:
        a.B, a.H, a.N, a.d, a.dtype = {"cfg3": (4, 8, 1024, 64, "f16"), "cfg4": (8, 16, 4096, 64, "f16"),
                                       "cfg4bf16": (8, 16, 4096, 64, "bf16"), "cfg5": (8, 16, 8192, 128, "f16")}[a.config]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{a.tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)

    ks = one(os.path.join(src, "kt", "**", "*_kernel_stats.csv"))
    kernel_ms, redo_us = None, None
    if ks:
        shutil.copy(ks, os.path.join(dst, f"{a.tag}_kernel_stats.csv"))
        for r in csv.DictReader(open(ks)):
            if "fa_fwd" in r["Name"] and not is_redo(r["Name"]):
                kernel_ms = float(r["AverageNs"]) / 1e6
            elif "fa_fwd" in r["Name"]:
                redo_us = float(r["AverageNs"]) / 1e3
    line = os.path.join(src, "bench_line_under_kernel_trace.json")
    bench = json.load(open(line)) if os.path.exists(line) and os.path.getsize(line) else None

    sq, mf = pmc_avgs(os.path.join(src, "pmc_sq")), pmc_avgs(os.path.join(src, "pmc_mfma"))
    fe, wr, l2 = (pmc_avgs(os.path.join(src, x)) for x in ("pmc_fetch", "pmc_write", "pmc_l2"))
    flops = 4.0 * a.B * a.H * a.N * a.N * a.d
    alg_bytes = 3.0 * a.B * a.H * a.N * a.d * 2 + a.B * a.H * a.N * a.d * (4 if a.out == "f32" else 2)
    summ = {"tag": a.tag, "config": vars(a), "kernel_avg_ms_kernel_trace": kernel_ms, "bench_line_same_run": bench,
            "flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes, "pmc": {}}
    for name, d in (("sq", sq), ("mfma", mf), ("fetch", fe), ("write", wr), ("l2", l2)):
        summ["pmc"][name] = {k: v for k, v in d.items() if not k.startswith("_")}
    if sq.get("_meta"):
        summ["kernel_resources"] = sq["_meta"]
    # derived figures
    der = {}
    if "GRBM_GUI_ACTIVE" in mf and "SQ_VALU_MFMA_BUSY_CYCLES" in sq:
        cyc = mf["GRBM_GUI_ACTIVE"] / 8.0                    # counter sums the 8 XCDs
        der["gpu_cycles_per_launch"] = cyc
        der["mfma_busy_frac_of_simd_time"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc   # 1024 SIMDs
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in mf:
            der["mfma_coexec_over_mfma_busy"] = mf["SQ_VALU_MFMA_COEXEC_CYCLES"] / sq["SQ_VALU_MFMA_BUSY_CYCLES"]
        if kernel_ms:
            der["effective_clock_GHz_profiled"] = cyc / (kernel_ms * 1e-3) / 1e9
    if "SQ_WAVE_CYCLES" in sq:
        w = sq["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in sq:
                der[k + "_frac_of_wave_cycles"] = sq[k] / w
    # HBM traffic: FETCH_SIZE / WRITE_SIZE are in KiB units of 64-B... rocprofv3 reports KB; on gfx950
    # FETCH_SIZE reads exactly half of a wide coalesced stream (MI355X_MICROARCH.md, HBM): double it.
    if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
        rd = fe["FETCH_SIZE"] * 1024.0 * 2.0
        wb = wr["WRITE_SIZE"] * 1024.0
        der["hbm_read_bytes_per_launch_corrected"] = rd
        der["hbm_write_bytes_per_launch"] = wb
        der["hbm_bytes_per_launch"] = rd + wb
        der["hbm_bytes_over_algorithmic"] = (rd + wb) / alg_bytes
        traffic_file = "hbm_traffic.json" if (a.B, a.H, a.N, a.d, a.dtype) == (8, 16, 4096, 64, "f16") else f"{a.tag}_hbm_traffic.json"
        json.dump({"config": {"B": a.B, "H": a.H, "N": a.N, "d": a.d, "dtype": a.dtype, "out": a.out},
                   "kernel": (sq.get("_meta") or {}).get("Kernel_Name", ""),
                   "hbm_bytes_per_launch": rd + wb, "read_bytes_fetch_size_x2": rd, "write_bytes": wb,
                   "source": f"profiles/{a.tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"},
                  open(os.path.join(dst, traffic_file), "w"), indent=1)
    if "TCC_HIT_sum" in l2 and "TCC_MISS_sum" in l2:
        der["l2_hit_rate"] = l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"])
    if kernel_ms:
        der["tflops_from_kernel_trace"] = flops / (kernel_ms * 1e-3) / 1e12
        der["frac_of_2500_TF_peak"] = der["tflops_from_kernel_trace"] / 2500.0
    if redo_us is not None:
        der["redo_kernel_avg_us"] = redo_us   # (nothing marked on the bench data: the cost of looking)
        der["kernel_plus_redo_ms"] = (kernel_ms or 0.0) + redo_us / 1e3
    summ["derived"] = der
    # per-dispatch durations of the long kernel-trace run: the post-idle transient the driver's 5+20 window sits in
    kl = one(os.path.join(src, "kt_long", "**", "*_kernel_trace.csv"))
    if kl:
        rows = [r for r in csv.DictReader(open(kl)) if "fa_fwd" in r["Kernel_Name"] and not is_redo(r["Kernel_Name"])]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        with open(os.path.join(dst, f"{a.tag}_dispatch_times.csv"), "w") as f:
            f.write("dispatch,start_us_since_first,duration_us\n")
            t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
            for i, r in enumerate(rows):
                f.write(f"{i},{(int(r['Start_Timestamp']) - t0) / 1e3:.1f},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}\n")
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        if len(durs) >= 60:
            summ["dispatch_us"] = {"first": durs[0], "mean_5_to_24": sum(durs[5:25]) / 20.0, "slowest": max(durs), "mean_last_50": sum(durs[-50:]) / 50.0}
    json.dump(summ, open(os.path.join(dst, f"{a.tag}_summary.json"), "w"), indent=1)
    print(json.dumps(der, indent=1))


if __name__ == "__main__":
    main()
