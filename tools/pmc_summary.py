#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (tools/pmc_algos.sh)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
assert fs, "no counter_collection.csv under " + d
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
    name = r["Kernel_Name"]
    if "fa_" not in name:
        continue
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    a = {k: sum(v) / len(v) for k, v in cs.items()}
    short = name.split("(")[0].replace("void ", "")
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    simd_cycles = 1024.0 * cyc
    out = [f"{short[:70]:70s} n={len(next(iter(cs.values())))}", f"cycles/launch {cyc / 1e6:.3f} M"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and cyc:
        out.append(f"MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.3f}")
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in a:
            out.append(f"coexec/busy {a['SQ_VALU_MFMA_COEXEC_CYCLES'] / a['SQ_VALU_MFMA_BUSY_CYCLES']:.3f}")
    if "SQ_WAVE_CYCLES" in a:
        wc = a["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if k in a:
                out.append(f"{k[3:]} {a[k] / wc:.3f}")
    if "SQ_INSTS_VALU" in a:
        out.append(f"VALU insts {a['SQ_INSTS_VALU'] / 1e6:.1f} M")
    if "--lds" in sys.argv:
        out = out[:2]
        for k in ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_MFMA", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU"):
            if k in a:
                out.append(f"{k[3:]} {a[k] / 1e6:.2f} M")
    print("  ".join(out))
