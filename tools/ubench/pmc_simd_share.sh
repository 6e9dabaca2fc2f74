#!/bin/bash
# PMC view of tools/ubench/simd_share.bin: are the time-based "no overlap" numbers cycles or clock throttling?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04/pmc_simd_share
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/p -- $R/tools/ubench/simd_share.bin > $O/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/p/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {"wg": r["Workgroup_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
print("dispatch wg_size  cycles(GRBM/8)  mfma_busy  valu_busy  coexec  insts_mfma insts_valu")
for d, c in by.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    simd = cyc * 1024
    print("%4s %5s %12.0f   %.3f   %.3f   %.3f   %.3g %.3g" % (d, c["wg"], cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4 * c["SQ_ACTIVE_INST_VALU"] / simd, c["SQ_VALU_MFMA_COEXEC_CYCLES"] / simd, c["SQ_INSTS_MFMA"], c["SQ_INSTS_VALU"]))
PY
cat $O/run.log | tail -25
