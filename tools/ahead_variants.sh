#!/bin/bash
# A/B of the LDS prefetch distance of the LDS-image products (PILOT_LDS_AHEAD; GPU): the installed library and every
# build_variants/lib_*.so on the K sweep points with an LDS image (K = 80, 96, 100, 112) and on c4.
#   tools/ahead_variants.sh <out file>
O=$1
for lib in "" build_variants/lib_*.so; do
  name=$(basename "${lib:-base}" .so)
  export PILOT_AMD_LIB=${lib:+$PWD/$lib}
  [ -z "$lib" ] && unset PILOT_AMD_LIB
  for K in 72 80 96 100 112; do echo -n "$name  "; timeout 120 python3 tools/k_point.py $K; done
  echo -n "$name  c4: "; timeout 300 python3 bench.py --config c4 --steps 5 --warmup 1 --no-extras --no-cpu-baseline | python3 -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.3f kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  echo -n "$name  c3 reg 0.01: "; timeout 300 python3 tools/k_point.py 50 0.01 | tail -1
  timeout 300 python3 tools/sinkhorn_full_grid_check.py c4:0.1:16 2>&1 | grep -v "^make" | cut -c1-200
done 2>&1 | tee $O
