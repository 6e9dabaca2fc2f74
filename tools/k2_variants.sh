#!/bin/bash
# A/B of the headline Sinkhorn kernel (GPU): the installed library and every build_variants/lib_*.so (sk_inst parts 8 / 9 rebuilt with
# -D switches) on bench.py's default workload (c3, reg 0.1): pairs/s, kernel ms (HIP events) and the whole-grid parity check.
#   tools/k2_variants.sh <out dir> [bench args...]
O=$1; shift
mkdir -p $O
for lib in "" build_variants/lib_*.so; do
  name=$(basename "${lib:-base}" .so)
  export PILOT_AMD_LIB=${lib:+$PWD/$lib}
  [ -z "$lib" ] && unset PILOT_AMD_LIB
  for rep in 1 2; do
    timeout 300 python3 bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline "$@" > $O/$name.$rep.json 2> $O/$name.err
    python3 - $O/$name.$rep.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-12s value %.4g pairs/s  ms/step %.4f  kernel_ms %.4f  mean updates %.2f" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["mean_updates_per_pair"]))
PY
  done
  timeout 300 python3 tools/sinkhorn_full_grid_check.py c3:0.1 2>&1 | grep -v "^make" | cut -c1-230
done
