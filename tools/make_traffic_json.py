#!/usr/bin/env python3
"""traffic.json for bench.py's roofline.traffic: HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE in separate passes, KB; FETCH_SIZE doubled per MI355X_MICROARCH.md) and its average duration from
the rocprofv3 --kernel-trace --stats run, keyed by workload and stamped with the git revision they were measured at."""
import csv, json, os, re, subprocess, sys
d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = json.load(open(os.path.join(d, "bench_under_rocprof.json")))
prec = {"f64": "fp64"}.get(bench["dtype"], "f16x2" if "fp16" in bench["dtype"] else ("bf16x3" if "bf16" in bench["dtype"] else "fp32"))
SUFFIX = os.environ.get("TRAFFIC_SUFFIX", "")          # "_fp32", ...: the ladder's other precisions (same workload)
if SUFFIX:
    prec = SUFFIX[1:]
kern, fetch, write = None, None, None
cnt = {}
for line in open(os.path.join(d, "rocprofv3_pmc_summary_bench_c3%s.txt" % SUFFIX)):
    m = re.match(r"== (sinkhorn_stream_kernel<[^>]*>)", line)
    if m and kern is None:
        kern = m.group(1)
        cur = True
    elif line.startswith("=="):
        cur = False
    elif kern and cur:
        m = re.match(r"\s+(\w+)\s+mean ([0-9.e+]+)", line)
        if m and m.group(1) not in cnt: cnt[m.group(1)] = float(m.group(2))
        if m and m.group(1) == "FETCH_SIZE" and fetch is None: fetch = float(m.group(2))
        if m and m.group(1) == "WRITE_SIZE" and write is None: write = float(m.group(2))
# fractions of the launch's SIMD cycles: GRBM_GUI_ACTIVE counts the 8 XCDs, a launch spans GRBM / 8 cycles on each of 1024 SIMDs
simd = cnt.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
frac = lambda k, mul=1.0: round(mul * cnt[k] / simd, 4) if simd and k in cnt else None
avg_ns = None
for row in csv.DictReader(open(os.path.join(d, "rocprofv3_kernel_stats_bench_c3%s.csv" % SUFFIX))):
    if "sinkhorn_stream_kernel" in row["Name"] and (avg_ns is None or float(row["AverageNs"]) > avg_ns):
        avg_ns = float(row["AverageNs"])
sha = ""
try:
    sha = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    pass
if not sha:      # the GPU box gets a snapshot without .git: the revision travels in tools/.git_sha (written before the run)
    try:
        sha = open(os.path.join(root, "tools", ".git_sha")).read().strip()
    except OSError:
        sha = "unknown"
N, K = bench["config"]["n_patients"], bench["config"]["n_cell_types"]
s = 8 if prec == "fp64" else 4
out = {
    "_comment": "HBM traffic of the dominant kernel per launch from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE "
                "passes, tools/profile_pmc.sh; counters in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is) and its "
                "average duration from rocprofv3 --kernel-trace --stats (tools/gpu_round_report.sh), 1x MI355X.",
    "c3|%g|%s" % (bench["config"]["reg"], prec): {
        "kernel": kern, "fetch_size_kb": fetch, "write_size_kb": write,
        "traffic_bytes": int(1024 * (2 * fetch + write)) if fetch is not None and write is not None else None,
        "algorithmic_bytes": N * N * (2 * K * s + s),
        "kernel_ms_rocprofv3": round(avg_ns / 1e6, 4) if avg_ns else None,
        "mfma_busy": frac("SQ_VALU_MFMA_BUSY_CYCLES"), "valu_busy": frac("SQ_ACTIVE_INST_VALU", 4.0), "coexec": frac("SQ_VALU_MFMA_COEXEC_CYCLES"),
        "sq_insts_mfma": cnt.get("SQ_INSTS_MFMA"), "sq_insts_valu": cnt.get("SQ_INSTS_VALU"),
        "pmc_note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES, coexec = SQ_VALU_MFMA_COEXEC_CYCLES, valu_busy = 4 x SQ_ACTIVE_INST_VALU (the "
                    "sum over the resident waves of the cycles a vector instruction of theirs was in flight: exceeds 1 with more than "
                    "two waves per SIMD), each over the launch's SIMD cycles GRBM_GUI_ACTIVE / 8 x 1024",
        "git": sha,
    },
}
if SUFFIX:            # merge into the file the default precision wrote
    base = json.load(open(os.path.join(d, "traffic.json")))
    base.update({k: v for k, v in out.items() if k != "_comment"})
    out = base
print(json.dumps(out, indent=1))
