#!/bin/bash
# A/B builds of the POT-literal kernel: each argument is one set of -D flags for pilot_ot.hip (e.g. -DPILOT_GENERIC_WG=512).  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
cp ../libpilot_ot.so /tmp/libpilot_ot.keep.so
for v in "" "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $v -c -o /tmp/pilot_ot_var.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_var.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/sk_inst_*.o -ldl
  echo "== [$v]"; (cd $R && timeout 120 python3 tools/big_k_probe.py) | sed 's/  exact.*//'
done
cp /tmp/libpilot_ot.keep.so ../libpilot_ot.so
