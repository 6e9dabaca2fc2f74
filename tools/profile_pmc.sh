#!/bin/bash
# rocprofv3 PMC passes over a short bench.py run (GPU box).  Each pass is its own run with --kernel-trace only
# (never combined with sys/hip/hsa tracing).  Usage: tools/profile_pmc.sh <outdir> [bench args...]
# BENCH_PY=<script relative to the repo>: profile that script (with the given args) instead of bench.py
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
run_pass() {
  name=$1; shift
  if [ -n "${BENCH_PY:-}" ]; then
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/$OUT/$name" -- python3 "$R/$BENCH_PY" $BENCH_ARGS > "$R/$OUT/$name.log" 2>&1
  else
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/$OUT/$name" -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline $BENCH_ARGS > "$R/$OUT/$name.log" 2>&1
  fi
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
cd "$R"
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
