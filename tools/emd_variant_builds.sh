#!/bin/bash
# A/B builds of the exact-OT kernel (EMD_LAZY: lazy restarts; EMD_WPE: waves-per-EU register caps).  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
cp ../libpilot_ot.so /tmp/libpilot_ot.keep.so
for v in "0 0" "0 1" "1 0" "1 1"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DEMD_LAZY=$1 -DEMD_WPE=$2 $EMD_EXTRA -c -o /tmp/pilot_ot_var.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_var.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/sk_inst_*.o -ldl
  for cfg in c3 c4 c2; do
    python3 $R/bench.py --mode emd --config $cfg --steps $([ $cfg = c4 ] && echo 3 || echo 20) --warmup 2 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('EMD_LAZY=$1 EMD_WPE=$2 $cfg: %.3f ms' % d['ms_per_step'])"
  done
done
cp /tmp/libpilot_ot.keep.so ../libpilot_ot.so
