#!/bin/bash
# Diagnostic builds of the exact-OT kernel (EMD_PROF = n, emd_kernels.hpp): a wave's clock ticks per pair, by section.  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
KEEP=$(mktemp /tmp/libpilot_ot.keep.XXXXXX.so)
cp ../libpilot_ot.so "$KEEP"
# (put the installed library back on ANY exit: an interrupted run must not leave a diagnostic build behind)
trap 'cp "$KEEP" "$R/pilot_amd/libpilot_ot.so"; rm -f "$KEEP"' EXIT
for st in 1 2 3 4 5 6 7 8 9; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DEMD_PROF=$st $EXTRA -c -o /tmp/pilot_ot_prof.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_prof.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/pilot_ot_labels.o build/sk_wide.o build/sk_inst_*.o -ldl -lpthread
  (cd $R; python3 tools/emd_stats.py ${1:-c3} | sed "s/^/EMD_PROF=$st (ticks\/16): /" | head -2 | tr '\n' ' '; echo)
done
