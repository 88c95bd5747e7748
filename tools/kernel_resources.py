#!/usr/bin/env python3
"""Compile the HIP sources with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "relationnetworks-clevr_amd", "csrc")
files = sys.argv[1:] or [f for f in sorted(os.listdir(src)) if f.endswith(".hip")]
for f in files:
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(src, f), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?):\s*(\S+)", line)
        if "Function Name" in line:
            cur = {"name": line.split("Function Name:")[1].split()[0]}
        elif m and cur:
            cur[m.group(1).strip()] = m.group(2)
            if m.group(1).strip().startswith("LDS Size"):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
                print("%-62s vgpr %-4s agpr %-4s spill %s scratch %-4s occ %s lds %s" % (
                    name[:62], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("VGPR Spill", cur.get("VGPRs Spill")),
                    cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
    if p.returncode:
        print(p.stderr[-2000:])
