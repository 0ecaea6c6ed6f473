#!/usr/bin/env python3
"""Energy per MFMA slot at the package power cap: runs tools/microbench/slot_model in `sustain` mode (one instruction
mix, two waves per SIMD, every CU, ~3 s) while sampling rocm-smi, and prints watts, sclk and nanojoules per slot per SIMD.
What a joule buys decides the d=64 forward, which sits at the cap."""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "microbench", "slot_model")
KINDS = [(1, "MFMA 32x32x16 only"), (2, "2 x MFMA 16x16x32 only"), (10, "vector only: folded"), (20, "vector only: scalar fma"),
         (30, "vector only: packed"), (11, "32x32x16 + folded"), (21, "32x32x16 + scalar fma"), (31, "32x32x16 + packed"),
         (12, "2 x 16x16x32 + folded"), (22, "2 x 16x16x32 + scalar fma"), (32, "2 x 16x16x32 + packed")]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
    lines = [l for l in out.strip().splitlines() if l and not l.startswith("WARNING")]
    vals = lines[1].split(",") if len(lines) > 1 else []
    try:
        m = re.search(r"(\d+)Mhz", vals[7])
        return float(vals[-1]), float(m.group(1)) if m else float("nan")
    except Exception:  # noqa: BLE001
        return float("nan"), float("nan")


print(f"{'mix':32s} {'W':>7s} {'sclk MHz':>9s} {'ns/slot':>8s} {'cyc/slot':>9s} {'nJ/slot/SIMD':>13s}  (1024 SIMDs; idle floor not subtracted)")
for kind, name in KINDS:
    p = subprocess.Popen([BIN, "sustain", str(kind), str(secs)], stdout=subprocess.PIPE, text=True)
    samples = []
    t0 = time.time()
    while p.poll() is None:
        if time.time() - t0 > 1.0:
            samples.append(smi())
        time.sleep(0.25)
    out = p.stdout.read()
    m = re.search(r"([\d.]+) ns per slot per SIMD, ([\d.]+) cycles", out)
    ns, cyc = (float(m.group(1)), float(m.group(2))) if m else (float("nan"), float("nan"))
    ws = [w for w, _ in samples if w == w]
    cs = [c for _, c in samples if c == c]
    w = sum(ws) / max(len(ws), 1)
    c = sum(cs) / max(len(cs), 1)
    print(f"{name:32s} {w:7.0f} {c:9.0f} {ns:8.2f} {cyc:9.2f} {w * ns / 1024:13.2f}", flush=True)
    time.sleep(1.0)
