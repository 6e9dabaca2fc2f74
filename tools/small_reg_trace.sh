#!/bin/bash
# kernel breakdown of c3 at reg 0.01 (AUTO: two exponent bands + f64 hand-over); run on the GPU box via gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sr; rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/small_reg_trace.py > $O/out.txt 2>/dev/null
cat $O/out.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-100s calls %4s  total %10.3f ms  avg %10.3f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
