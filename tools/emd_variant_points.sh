#!/bin/bash
# A/B builds of the exact-OT kernel timed with tools/emd_point.py: POINTS="4|real|50" tools/emd_variant_points.sh "<-D flags>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
KEEP=$(mktemp /tmp/libpilot_ot.keep.XXXXXX.so)
cp ../libpilot_ot.so "$KEEP"
# (put the installed library back on ANY exit: an interrupted run must not leave a diagnostic build behind)
trap 'cp "$KEEP" "$R/pilot_amd/libpilot_ot.so"; rm -f "$KEEP"' EXIT
for v in "" "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $v -c -o /tmp/pilot_ot_var.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_var.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/pilot_ot_labels.o build/sk_wide.o build/sk_inst_*.o -ldl -lpthread
  for pt in ${POINTS:-4 real 50}; do
    echo "[$v] $(timeout 120 python3 $R/tools/emd_point.py $pt 2>&1 | tail -1)"
  done
done
