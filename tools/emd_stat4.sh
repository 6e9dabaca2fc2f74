R=${GRAFT_REPO_ROOT}
cd $R/pilot_amd/csrc
KEEP=$(mktemp /tmp/libpilot_ot.keep.XXXXXX.so)
cp ../libpilot_ot.so "$KEEP"
# (put the installed library back on ANY exit: an interrupted run must not leave a diagnostic build behind)
trap 'cp "$KEEP" "$R/pilot_amd/libpilot_ot.so"; rm -f "$KEEP"' EXIT
for st in 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DEMD_STAT=$st -c -o /tmp/pilot_ot_stat.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_stat.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/sk_wide.o build/sk_inst_*.o -ldl
  python3 $R/tools/emd_stats.py c3 | sed "s/^/EMD_STAT=$st: /" | head -1
done
