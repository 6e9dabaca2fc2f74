#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace csv (gaps between dispatches)."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows = [r for r in rows if 'copyBuffer' not in r['Kernel_Name'] or True]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
vals = [i for i, r in enumerate(rows) if 'value_kernel' in r['Kernel_Name']]
lo, hi = vals[-2] + 1, vals[-1] + 1
t0 = int(rows[lo]['Start_Timestamp']); prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-48s q=%s start %8.1f us  dur %8.1f us  gap %6.1f" % (r['Kernel_Name'][:48], r['Queue_Id'], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("step span %.1f us" % ((prev_end - t0) / 1e3))
