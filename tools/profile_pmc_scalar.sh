#!/bin/bash
# One more rocprofv3 PMC pass: scalar / branch / message instruction counts (the exact-OT kernel is control-flow heavy).
# Usage: tools/profile_pmc_scalar.sh <outdir> [bench args...]   (same conventions as tools/profile_pmc.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH_ARGS="$*"
# BENCH_PY=<script relative to the repo>: profile that script (with the given args) instead of bench.py
if [ -n "${BENCH_PY:-}" ]; then PROG="$R/$BENCH_PY"; else PROG="$R/bench.py"; BENCH_ARGS="--steps 4 --warmup 1 --no-cpu-baseline $BENCH_ARGS"; fi
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d "$R/$OUT/sca" -- python3 "$PROG" $BENCH_ARGS > "$R/$OUT/sca.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VSKIPPED --output-format csv -d "$R/$OUT/sca2" -- python3 "$PROG" $BENCH_ARGS > "$R/$OUT/sca2.log" 2>&1
cd "$R"
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
