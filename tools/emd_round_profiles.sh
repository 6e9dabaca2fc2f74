#!/bin/bash
# rocprofv3 PMC passes of the exact-OT kernels (c3, the reference test's cohort, the K = 64/65 and 128/129 steps) and the HBM
# traffic of the cell-level kernel at BASELINE config 5 -> <out>/rocprofv3_pmc_summary_emd_*.txt, emd_instr.json, cellw2_traffic.json
# usage (GPU box): tools/emd_round_profiles.sh gpurun_out/emd_profiles [nocell]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1
O=$R/$OUT
mkdir -p $O
cd $R
one() {   # name, BENCH_PY or "", args...
  name=$1; py=$2; shift 2
  BENCH_PY=$py bash tools/profile_pmc.sh $OUT/tmp_$name "$@" > /dev/null 2>&1
  BENCH_PY=$py bash tools/profile_pmc_scalar.sh $OUT/tmp_$name "$@" > /dev/null 2>&1
  cp $O/tmp_$name/summary.txt $O/rocprofv3_pmc_summary_emd_$name.txt
  rm -rf $O/tmp_$name
}
one c3 "" --mode emd
one kidney tools/emd_point.py real
for k in 64 65 128 129; do one k$k tools/emd_point.py $k; done
if [ "${2:-}" != "nocell" ]; then
  export TMPDIR=/tmp
  (cd /tmp && for c in FETCH_SIZE WRITE_SIZE; do
     rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/tmp_cell/$c -- python3 $R/bench.py --mode cellw2 --no-cpu-baseline > /dev/null 2>&1
   done)
  python3 tools/summarize_pmc.py $OUT/tmp_cell > $O/rocprofv3_pmc_summary_cellw2_200x5000x30.txt 2>&1
  rm -rf $O/tmp_cell
fi
python3 tools/make_emd_instr_json.py $OUT
