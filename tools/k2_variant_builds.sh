#!/bin/bash
# A/B builds of the Sinkhorn stream kernels: each argument is one set of -D flags for the translation units K2_PARTS
# (default "8 9": the fp16-split configuration; "6 7": bf16-split).  GPU box.  K2_CFGS: bench configs (default "c2 c3").
# K2_HOST=1: pilot_ot.hip is rebuilt with the same flags.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/pilot_amd/csrc
PARTS=${K2_PARTS:-8 9}
KEEP=$(mktemp /tmp/libpilot_ot.keep.XXXXXX.so)
cp ../libpilot_ot.so "$KEEP"
# (put the installed library back on ANY exit: an interrupted run must not leave a diagnostic build behind)
trap 'cp "$KEEP" "$R/pilot_amd/libpilot_ot.so"; rm -f "$KEEP"' EXIT
run() { for cfg in ${K2_CFGS:-c2 c3}; do timeout 120 python3 $R/bench.py --config $cfg --no-extras --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $cfg: %.4f ms  kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done; }
run baseline
for v in "$@"; do
  objs=""
  for part in 0 1 2 3 4 5 6 7 8 9; do
    if [[ " $PARTS " == *" $part "* ]]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function $v -DSK_PART=$part -c -o /tmp/sk_var_$part.o sk_inst.hip 2>/dev/null &
      objs="$objs /tmp/sk_var_$part.o"
    else objs="$objs build/sk_inst_$part.o"; fi
  done
  host=build/pilot_ot.o
  if [ -n "$K2_HOST" ]; then      # the flags also change constants the host mirrors (occupancy rules)
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $v -c -o /tmp/pilot_ot_var.o pilot_ot.hip 2>/dev/null &
    host=/tmp/pilot_ot_var.o
  fi
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so $host build/pilot_ot_multi.o build/pilot_ot_consumers.o build/sk_wide.o $objs -ldl
  run "$v"
done
