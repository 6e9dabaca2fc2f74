#!/bin/bash
# HBM traffic of the pre-pass kernels (GPU): rocprofv3 PMC passes, FETCH_SIZE and WRITE_SIZE in SEPARATE runs with --kernel-trace
# only (MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC slots"), over tools/prepass_probe.py at BASELINE c3's cohort shape, plus a
# --kernel-trace --stats run for the durations.  Writes <out>/rocprofv3_kernel_stats_prepass_c3.csv,
# <out>/rocprofv3_pmc_summary_prepass_c3.txt and <out>/prepass_traffic.json (what bench.py's prepass.roofline.traffic quotes).
#   tools/prepass_pmc.sh <out dir>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/$1
mkdir -p $O/tmp_prepass
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/tmp_prepass/$c -- python3 $R/tools/prepass_probe.py 1800000 30 50 float32 5 > $O/tmp_prepass/$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_prepass/stats -- python3 $R/tools/prepass_probe.py 1800000 30 50 float32 10 > $O/prepass_probe_c3.txt 2>&1
cd $R
cp $(find $O/tmp_prepass/stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_prepass_c3.csv
python3 tools/prepass_traffic.py $O/tmp_prepass $O > $O/rocprofv3_pmc_summary_prepass_c3.txt 2>&1
cat $O/rocprofv3_pmc_summary_prepass_c3.txt
rm -rf $O/tmp_prepass
