#!/bin/bash
# A/B of the refill gate of the fast Sinkhorn kernel at one / two row-tiles (PILOT_REFILL_GATE_RT1 / _RT2; GPU): the installed library and
# every build_variants/lib_*.so on the reference test's cohort (634 x 14), K = 2 .. 32 at N = 600, c1 and c2.
#   tools/gate_variants.sh <out file>
O=$1
for lib in "" build_variants/lib_*.so; do
  name=$(basename "${lib:-base}" .so)
  export PILOT_AMD_LIB=${lib:+$PWD/$lib}
  [ -z "$lib" ] && unset PILOT_AMD_LIB
  echo -n "$name  "; timeout 200 python3 tools/real_cohort_probe.py 2>&1 | grep "pair grid" | cut -c1-150
  for K in 2 4 8 12 14 16 20 24 30 32; do echo -n "$name  "; timeout 120 python3 tools/k_point.py $K; done
  for c in c1 c2; do echo -n "$name  $c: "; timeout 200 python3 bench.py --config $c --no-extras --no-cpu-baseline | python3 -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done
  timeout 300 python3 tools/sinkhorn_full_grid_check.py 600x14:0.1 2>&1 | grep -v "^make" | cut -c1-200
done 2>&1 | tee $O
