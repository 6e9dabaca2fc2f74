#!/usr/bin/env python3
"""Per-kernel HBM traffic of the pre-pass from the rocprofv3 PMC passes of tools/prepass_pmc.sh: FETCH_SIZE and WRITE_SIZE are
in KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md "HBM"), WRITE_SIZE is taken
as is.  Prints the summary and writes <out>/prepass_traffic.json = HBM bytes of ONE pre-pass call (sum over its kernels)."""
import collections, csv, glob, json, os, subprocess, sys
d, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?").split("(")[0].replace("void ", "").replace("pilot::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
calls, avg = {}, {}
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Name"].split("(")[0].replace("void ", "").replace("pilot::", "")
        calls[k] = int(row["Calls"]); avg[k] = float(row["AverageNs"]) / 1e3
# one pre-pass call of the probe = proportions (count, prior, proportions) + medians (type_count, prep, group, 4 x (hist, pick));
# launches per call from the stats run: calls / calls of group_rows_kernel (the medians) resp. count_kernel (the proportions)
per_call = {}
n_med = calls.get("group_rows_kernel<float>", 1)
n_cnt = calls.get("count_kernel", 1)
total_b, total_us, rows = 0.0, 0.0, []
for k in sorted(acc, key=lambda k: -avg.get(k, 0) * calls.get(k, 0)):
    if k.startswith("__amd") or k not in calls:
        continue
    f = acc[k].get("FETCH_SIZE"); w = acc[k].get("WRITE_SIZE")
    fetch = 2.0 * 1024.0 * sum(f) / len(f) if f else 0.0
    write = 1024.0 * sum(w) / len(w) if w else 0.0
    base = n_cnt if k in ("count_kernel", "prior_kernel", "proportions_kernel") else n_med
    launches = calls[k] / base
    per_call[k] = {"launches_per_call": launches, "avg_us": round(avg[k], 2), "hbm_read_bytes_per_launch": int(fetch), "hbm_write_bytes_per_launch": int(write)}
    total_b += launches * (fetch + write); total_us += launches * avg[k]
    rows.append("%-34s x%-4g avg %8.1f us   read %8.1f MB   write %8.1f MB   -> %6.0f GB/s" % (k, launches, avg[k], fetch / 1e6, write / 1e6, (fetch + write) / avg[k] / 1e3))
print("pre-pass at 1 800 000 cells x 30 dims (float32), 50 types, 600 samples; per launch: rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE, separate passes;")
print("durations: rocprofv3 --kernel-trace --stats of the same command")
print("\n".join(rows))
print("one call (medians + proportions): %.1f MB of HBM traffic, %.1f us of kernels -> %.0f GB/s" % (total_b / 1e6, total_us, total_b / total_us / 1e3))
sha = ""
try:
    sha = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    pass
if not sha:
    try:
        sha = open(os.path.join(root, "tools", ".git_sha")).read().strip()
    except OSError:
        sha = "unknown"
json.dump({"cells": 1800000, "dims": 30, "dtype": "float32", "types": 50, "samples": 600, "hbm_bytes_per_call": int(total_b),
           "kernel_us_per_call": round(total_us, 1), "per_kernel": per_call, "git": sha,
           "how": "rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE per launch summed over the launches of one call, tools/prepass_pmc.sh"},
          open(os.path.join(out, "prepass_traffic.json"), "w"), indent=1)
