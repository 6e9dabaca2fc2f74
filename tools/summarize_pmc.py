#!/usr/bin/env python3
"""Per-kernel mean of every PMC counter found under <dir>/*/**/*counter_collection.csv."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            k = k.split("(")[0].replace("void pilot::", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0].replace("void pilot::", "")
            dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0]))):
    if not k.startswith("sinkhorn") and not k.startswith("emd") and not k.startswith("cell_w2"):
        continue
    print("== %s   dispatches=%d  mean duration (profiled) = %.1f us   max %.1f us" % (k, len(dur.get(k, [])), sum(dur[k]) / max(1, len(dur[k])), max(dur.get(k, [0]))))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s mean %.6g   (n=%d)   max %.6g" % (c, sum(v) / len(v), len(v), max(v)))
