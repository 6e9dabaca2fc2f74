#!/bin/bash
# A/B of pre-pass tuning constants (GPU): every build_variants/lib_*.so (pilot_ot.hip rebuilt with -DPILOT_GROUP_ROWS / -DPILOT_SELECT_*)
# under rocprofv3 --kernel-trace --stats with tools/prepass_probe.py; prints the kernels' average times per variant.
#   tools/prepass_variants.sh <out dir> [probe args...]
O=$1; shift
mkdir -p $O
for lib in "" build_variants/lib_*.so; do
  name=$(basename "${lib:-base}" .so)
  export PILOT_AMD_LIB=${lib:+$PWD/$lib}
  [ -z "$lib" ] && unset PILOT_AMD_LIB
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 tools/prepass_probe.py "$@" > $O/$name.txt 2>&1
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1)
  echo "== $name: $(grep -h 'bit-exact' $O/$name.txt | tr '\n' ' ')"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for key in ("select_hist", "group_rows", "count_kernel", "select_pick"):
        if key in n and "type_count" not in n:
            print("   %-14s avg %8.1f us (%s calls)" % (key, float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
