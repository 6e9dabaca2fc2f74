# A/B of run-time experiment switches (PILOT_OT_DEBUG values given as arguments) on c3 / c2 / c3 at reg 1.0
for cfg in "c3 0.1" "c2 0.1" "c3 1.0"; do set -- $cfg; for d in ${DEBUGS:-0 4096 8192 12288}; do
  echo -n "$1 reg $2 PILOT_OT_DEBUG=$d: "; PILOT_OT_DEBUG=$d python bench.py --config $1 --reg $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
