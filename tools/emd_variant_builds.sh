#!/bin/bash
# A/B builds of the exact-OT kernel: each argument is one set of -D flags (e.g. -DEMD_ULAB=0).  GPU box.
# (the installed library is put back on ANY exit: an interrupted run must not leave a variant build behind)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
KEEP=$(mktemp /tmp/libpilot_ot.keep.XXXXXX.so)
cp ../libpilot_ot.so "$KEEP"
trap 'cp "$KEEP" "$R/pilot_amd/libpilot_ot.so"; rm -f "$KEEP"' EXIT
for v in "" "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $v -c -o /tmp/pilot_ot_var.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_var.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/pilot_ot_labels.o build/sk_wide.o build/sk_inst_*.o -ldl -lpthread
  for cfg in ${EMD_CFGS:-c3 c4}; do
    timeout 120 python3 $R/bench.py --mode emd --config $cfg --steps $([ $cfg = c4 ] && echo 3 || echo 20) --warmup 2 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] $cfg: %.3f ms' % d['ms_per_step'])"
  done
done
