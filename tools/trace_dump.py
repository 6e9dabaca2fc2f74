"""Print the last N kernel records of a rocprofv3 --kernel-trace csv with the gaps between them. usage: trace_dump.py DIR N"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp']); prev = t0
for r in rows[-int(sys.argv[2]):]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-40s start %10.1f us dur %9.1f gap %9.1f" % (r['Kernel_Name'][:40], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3)); prev = e
