#!/bin/bash
# A/B builds of the Sinkhorn stream kernels: each argument is one set of -D flags for the bf16-split units (and f64 unit).  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
cp ../libpilot_ot.so /tmp/libpilot_ot.keep.so
run() { for cfg in ${K2_CFGS:-c2 c3}; do timeout 120 python3 $R/bench.py --config $cfg --no-extras --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $cfg: %.4f ms  kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done; }
run baseline
for v in "$@"; do
  for part in 6 7; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function $v -DSK_PART=$part -c -o /tmp/sk_var_$part.o sk_inst.hip 2>/dev/null & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so build/pilot_ot.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/sk_inst_0.o build/sk_inst_1.o build/sk_inst_2.o build/sk_inst_3.o build/sk_inst_4.o build/sk_inst_5.o /tmp/sk_var_6.o /tmp/sk_var_7.o -ldl
  run "$v"
done
cp /tmp/libpilot_ot.keep.so ../libpilot_ot.so
