#!/usr/bin/env python3
"""Kernel resource table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
    python tools/kres.py flashattention_kernel_project_amd/csrc/fa_fwd_rp16.hip [-DFLAG ...] [--filter SUBSTR]"""
import re, subprocess, sys
args = sys.argv[1:]
flt = None
if "--filter" in args:
    i = args.index("--filter"); flt = args[i + 1]; del args[i:i + 2]
src, extra = args[0], args[1:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-fvisibility=hidden",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, r in rows.items():
    if flt and flt not in k:
        continue
    print(f"{k:90s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} SGPR {r.get('TotalSGPRs','?'):>4} sspill {r.get('SGPRs Spill','?'):>3} vspill {r.get('VGPRs Spill','?'):>3} scratch {r.get('ScratchSize','?'):>4} occ {r.get('Occupancy','?')}")
