#!/usr/bin/env python3
"""Kernel timeline of the last bench step from a rocprofv3 --kernel-trace csv: start, duration and the gap to the previous kernel.
usage: step_timeline.py <dir given to rocprofv3 -d>"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
preps = [i for i, r in enumerate(rows) if 'sinkhorn_prep_kernel' in r['Kernel_Name']]
lo, hi = preps[-2], preps[-1]
t0 = int(rows[lo]['Start_Timestamp']); prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-60s start %8.1f us  dur %8.1f us  gap %6.1f" % (r['Kernel_Name'].replace('void pilot::', '')[:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("step span %.1f us (prep start -> next prep start: %.1f us)" % ((prev_end - t0) / 1e3, (int(rows[hi]['Start_Timestamp']) - t0) / 1e3))
