#!/bin/bash
# One GPU-box pass that produces everything the round commits under profiles/: usage tools/gpu_round_report.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/report_$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python tools/host_api_rate.py > $O/host_api_rate.txt 2>&1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
cd $R
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
bash tools/profile_pmc.sh gpurun_out/report_$TAG/pmc > /dev/null 2>&1
cat $O/pytest_gpu.txt $O/smoke.txt $O/host_api_rate.txt; cat $O/bench.json; head -8 $O/kernel_stats.csv; head -22 $O/pmc/summary.txt
