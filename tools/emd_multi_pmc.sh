set -u
cd $GRAFT_REPO_ROOT
for mode in 0 1; do
  export TOOL_SWITCHES=PILOT_OT_EMD_MULTI=$mode      # (read by tools/emd_point.py, tools/switches.py)
  BENCH_PY=tools/emd_point.py bash tools/profile_pmc.sh gpurun_out/pmc_kidney_m$mode real > /dev/null 2>&1
  BENCH_PY=tools/emd_point.py bash tools/profile_pmc_scalar.sh gpurun_out/pmc_kidney_m$mode real > /dev/null 2>&1
  cp gpurun_out/pmc_kidney_m$mode/summary.txt gpurun_out/pmc_kidney_m$mode.txt
  rm -rf gpurun_out/pmc_kidney_m$mode
done
cat gpurun_out/pmc_kidney_m0.txt gpurun_out/pmc_kidney_m1.txt
