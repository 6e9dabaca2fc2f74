#!/bin/bash
# One GPU-box pass that produces everything the round commits under profiles/: usage tools/gpu_round_report.sh <tag>
TAG=${1:-r06}
# (before sending this to the GPU box: git rev-parse --short HEAD > tools/.git_sha -- the box has no .git, the PMC summaries are stamped from that file)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/report_$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench_c3_reg0.1_n1.json 2> $O/bench.err
python bench.py --mode emd > $O/bench_emd_c3.json 2>> $O/bench.err
python bench.py --mode emd --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_emd_c4.json 2>> $O/bench.err
python bench.py --config c4 --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_c4_reg0.1_n1.json 2>> $O/bench.err
python bench.py --config c2 --no-extras --no-cpu-baseline > $O/bench_c2_reg0.1_n1.json 2>> $O/bench.err
python bench.py --precision fp32 --no-extras --no-cpu-baseline > $O/bench_c3_reg0.1_fp32_mfma.json 2>> $O/bench.err
python bench.py --precision bf16x3 --no-extras --no-cpu-baseline > $O/bench_c3_reg0.1_bf16x3.json 2>> $O/bench.err
python bench.py --precision fp64 --no-extras --no-cpu-baseline > $O/bench_c3_reg0.1_fp64_mfma.json 2>> $O/bench.err
python bench.py --mode cellw2 > $O/bench_cellw2_c5.json 2>> $O/bench.err
python bench.py --gpus 2 --logical-shards --no-cpu-baseline > $O/bench_two_logical_shards_one_process.json 2>> $O/bench.err
python bench.py --gpus 8 --logical-shards --no-cpu-baseline > $O/bench_eight_logical_shards_one_gpu.json 2>> $O/bench.err
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29544 python bench.py --gpus 1 --force-comm --no-cpu-baseline > $O/bench_one_rank_rccl_comm.json 2>> $O/bench.err
python tools/e2e_profile.py c3 > $O/e2e_tl_wasserstein_distance_c3.txt 2>&1
python tools/cellw2_parity_c5.py 4 8 $O/cellw2_parity_c5.json > $O/cellw2_parity_c5.txt 2>&1
python tools/shard_floor.py $O/shard_floor.json > $O/shard_floor_one_gpu.txt 2>&1
python tools/host_enqueue_time.py > $O/host_enqueue_time.txt 2>&1
python tools/host_to_host_probe.py > $O/host_to_host_probe.txt 2>&1
python tools/k2_occupancy_probe.py c3 c4 > $O/k2_occupancy_probe.txt 2>&1
python tools/small_k_occupancy_probe.py > $O/small_k_occupancy_probe.txt 2>&1
python tools/tail_probe.py c3 c4 > $O/tail_probe.txt 2>&1
python tools/solo_probe.py > $O/solo_probe.txt 2>&1
python tools/update_latency_probe.py > $O/update_latency_probe.txt 2>&1
python tools/k_sweep.py > $O/k_sweep.txt 2>&1
python tools/consumer_rate.py > $O/consumer_rate.txt 2>&1
python tools/small_reg_probe.py > $O/small_reg_c3_reg0.01.txt 2>&1
python tools/small_k_probe.py 2 3 4 5 6 7 8 12 > $O/small_k_probe.txt 2>&1
python tools/real_cohort_probe.py > $O/real_cohort_kidney_igan_g.txt 2>&1
python tools/big_k_probe.py > $O/big_k_probe.txt 2>&1
python tools/mid_reg_probe.py > $O/mid_reg_probe.txt 2>&1
python tools/sinkhorn_full_grid_check.py 2>&1 | grep -v "^make" > $O/sinkhorn_full_grid_check.txt
python tools/emd_full_grid_check.py c2 c3 c4 2>&1 | grep -v "^make" > $O/emd_full_grid_check.txt
for p in 2 4 8 12 real 14 16 17 30 50 64 65 100 128 129 160 256; do python tools/emd_point.py $p; done > $O/emd_points.txt 2>&1
python tools/emd_multi_probe.py 2 4 8 12 real 14 15 16 > $O/emd_multi_probe.txt 2>&1
python tools/e2e_profile.py real > $O/e2e_tl_wasserstein_distance_kidney.txt 2>&1
python tools/emd_point.py 100 2000 >> $O/emd_points.txt 2>&1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_emd -- python3 $R/bench.py --mode emd --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_emd_c4 -- python3 $R/bench.py --mode emd --config c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cellw2 -- python3 $R/bench.py --mode cellw2 --cell-patients 48 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cons -- python3 $R/tools/consumer_rate.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k80 -- python3 $R/tools/k_point.py 80 > $O/k_point_80.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k96 -- python3 $R/tools/k_point.py 96 > $O/k_point_96.txt 2>/dev/null
cd $R
cp $O/stats/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_bench_c3.csv 2>/dev/null
cp $O/stats_emd/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_emd_c3.csv 2>/dev/null
cp $O/stats_emd_c4/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_emd_c4.csv 2>/dev/null
cp $O/stats_cellw2/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_cellw2_48x5000.csv 2>/dev/null
cp $O/stats_cons/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_consumers.csv 2>/dev/null
cp $O/stats_k80/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_k80.csv 2>/dev/null
cp $O/stats_k96/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_k96.csv 2>/dev/null
bash tools/profile_pmc.sh gpurun_out/report_$TAG/pmc --no-extras > /dev/null 2>&1
cp $O/pmc/summary.txt $O/rocprofv3_pmc_summary_bench_c3.txt
# the pre-pass (kernel stats, FETCH / WRITE per kernel, prepass_traffic.json) and the sweep rows (reg 0.01 / 1.0, c4: sweep_rooflines.json)
bash tools/prepass_pmc.sh gpurun_out/report_$TAG > /dev/null 2>&1
bash tools/sweep_pmc.sh gpurun_out/report_$TAG > /dev/null 2>&1
# exact-OT kernels (c3, the reference test's cohort, the K = 64/65 and 128/129 steps) and the cell-level kernel's traffic:
# rocprofv3_pmc_summary_emd_*.txt, emd_instr.json, cellw2_traffic.json
bash tools/emd_round_profiles.sh gpurun_out/report_$TAG > /dev/null 2>&1
BENCH_PY=tools/k_point.py bash tools/profile_pmc.sh gpurun_out/report_$TAG/pmc_k80 80 > /dev/null 2>&1
cp $O/pmc_k80/summary.txt $O/rocprofv3_pmc_summary_k80.txt
BENCH_PY=tools/k_point.py bash tools/profile_pmc.sh gpurun_out/report_$TAG/pmc_k96 96 > /dev/null 2>&1
cp $O/pmc_k96/summary.txt $O/rocprofv3_pmc_summary_k96.txt
bash tools/emd_prof_builds.sh c3 > $O/emd_prof_c3.txt 2>&1
python tools/make_traffic_json.py $O > $O/traffic.json
# the other rungs of the precision ladder: kernel-trace average + the PMC passes, merged into traffic.json
for prec in fp32 bf16x3 fp64; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$prec -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --precision $prec > /dev/null 2>&1)
  cp $O/stats_$prec/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_bench_c3_$prec.csv 2>/dev/null
  bash tools/profile_pmc.sh gpurun_out/report_$TAG/pmc_$prec --no-extras --precision $prec > /dev/null 2>&1
  cp $O/pmc_$prec/summary.txt $O/rocprofv3_pmc_summary_bench_c3_$prec.txt
  TRAFFIC_SUFFIX=_$prec python tools/make_traffic_json.py $O > $O/traffic.json.new && mv $O/traffic.json.new $O/traffic.json
  rm -rf $O/stats_$prec $O/pmc_$prec
done
rm -rf $O/stats $O/stats_emd $O/stats_emd_c4 $O/stats_cellw2 $O/stats_cons $O/stats_k80 $O/stats_k96 $O/pmc $O/pmc_k80 $O/pmc_k96
# fuzz campaigns of the round (random shapes / options / degenerate inputs against the oracle, scipy and scikit-learn)
{ for f in "fuzz_emd.py 150 51" "fuzz_sinkhorn.py 80 52" "fuzz_prepass.py 60 53" "fuzz_consumers.py 60 54" "fuzz_multi.py 40 55" "fuzz_cellw2.py 20 56"; do
    echo "== tools/$f"; timeout 900 python tools/$f 2>&1 | tail -3; done; } > $O/fuzz_campaigns.txt 2>&1
cat $O/pytest_gpu.txt $O/smoke.txt; cut -c1-700 $O/bench_c3_reg0.1_n1.json; head -6 $O/rocprofv3_kernel_stats_bench_c3.csv; cat $O/traffic.json
