R=${GRAFT_REPO_ROOT}
cd $R/pilot_amd/csrc
cp ../libpilot_ot.so /tmp/libpilot_ot.keep.so
for st in 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DEMD_STAT=$st -c -o /tmp/pilot_ot_stat.o pilot_ot.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpilot_ot.so /tmp/pilot_ot_stat.o build/pilot_ot_multi.o build/pilot_ot_consumers.o build/sk_wide.o build/sk_inst_*.o -ldl
  python3 $R/tools/emd_stats.py c3 | sed "s/^/EMD_STAT=$st: /" | head -1
done
cp /tmp/libpilot_ot.keep.so ../libpilot_ot.so
