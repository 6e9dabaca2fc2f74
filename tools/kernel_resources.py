#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR/AGPR/spill/occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  Usage: python tools/kernel_resources.py"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-s", "-C", os.path.join(root, "pilot_amd", "csrc"), "resources"],
                     capture_output=True, text=True).stdout
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    txt = m.group(1)
    if txt.startswith("Function Name:"):
        name = txt.split(":", 1)[1].strip()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"\(.*", "", d).replace("void pilot::", "")
        cur = {"name": d}
        rows.append(cur)
    elif cur is not None and ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
cols = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs", "LDS Size [bytes/block]"]
print("%-60s %6s %6s %8s %5s %6s %8s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "SGPR", "LDS"))
for r in rows:
    print("%-60s %6s %6s %8s %5s %6s %8s" % tuple([r["name"][:60]] + [r.get(c, "?") for c in cols]))
