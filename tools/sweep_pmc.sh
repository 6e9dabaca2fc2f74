#!/bin/bash
# PMC passes (tools/profile_pmc.sh) of the sweep rows bench.py reports beside the headline: c3 at reg 0.01 and 1.0, c4 at reg 0.1, and
# the headline itself; writes <out>/rocprofv3_pmc_summary_{c3_reg0.01,c3_reg1,c4,bench_c3}.txt and <out>/sweep_rooflines.json
#   tools/sweep_pmc.sh <out dir (relative to the repo)>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$1
cd $R
bash tools/profile_pmc.sh $O/pmc_c3_reg001 --reg 0.01 --no-extras > /dev/null 2>&1
bash tools/profile_pmc.sh $O/pmc_c3_reg1 --reg 1.0 --no-extras > /dev/null 2>&1
bash tools/profile_pmc.sh $O/pmc_c4 --config c4 --no-extras > /dev/null 2>&1
bash tools/profile_pmc.sh $O/pmc_c3 --no-extras > /dev/null 2>&1
for w in c3_reg001:c3_reg0.01 c3_reg1:c3_reg1 c4:c4 c3:bench_c3; do
  cp $O/pmc_${w%%:*}/summary.txt $O/rocprofv3_pmc_summary_${w##*:}.txt
done
python3 tools/make_sweep_rooflines.py $O/sweep_rooflines.json "c3|0.01=$O/rocprofv3_pmc_summary_c3_reg0.01.txt" "c3|1=$O/rocprofv3_pmc_summary_c3_reg1.txt" \
  "c4|0.1=$O/rocprofv3_pmc_summary_c4.txt" "c3|0.1=$O/rocprofv3_pmc_summary_bench_c3.txt"
rm -rf $O/pmc_c3_reg001 $O/pmc_c3_reg1 $O/pmc_c4 $O/pmc_c3
