#!/bin/bash
# Diagnostic builds of the exact-OT kernel that report a per-pair event counter instead of the augmentation count
# (EMD_STAT = 1 Dijkstra steps, 2 tied-row relaxations, 3 path hops, 4 source rows visited by A rebuilds, 5 searches).  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
KEEP=$(mktemp /tmp/libpilot_ot.keep.XXXXXX.so)
cp ../libpilot_ot.so "$KEEP"
# (put the installed library back on ANY exit: an interrupted run must not leave a diagnostic build behind)
trap 'cp "$KEEP" "$R/pilot_amd/libpilot_ot.so"; rm -f "$KEEP"' EXIT
python3 $R/tools/emd_stats.py ${1:-c3} | sed 's/^/augmentations: /' | head -1
for st in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DEMD_STAT=$st -c -o /tmp/pilot_ot_stat.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_stat.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/pilot_ot_labels.o build/sk_wide.o build/sk_inst_*.o -ldl -lpthread
  python3 $R/tools/emd_stats.py ${1:-c3} | sed "s/^/EMD_STAT=$st: /" | head -1
done
