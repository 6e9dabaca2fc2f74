#!/usr/bin/env python3
"""sweep_rooflines.json for bench.py's reg_sweep / c4 rows: matrix-pipe occupancy of the DOMINANT kernel of each workload from the
rocprofv3 PMC passes of tools/profile_pmc.sh (one summary.txt per workload).
  python tools/make_sweep_rooflines.py <out.json> <key>=<summary.txt> ...          key = "c3|0.01", "c4|0.1", ..."""
import json, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"_comment": "per workload: the kernel with the largest total time in the PMC run; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES, coexec = "
                   "SQ_VALU_MFMA_COEXEC_CYCLES, valu_busy = 4 x SQ_ACTIVE_INST_VALU, each over the launch's SIMD cycles GRBM_GUI_ACTIVE / 8 x 1024 "
                   "(tools/profile_pmc.sh: counters in separate passes, --kernel-trace only); 1x MI355X"}
sha = ""
try:
    sha = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    pass
if not sha:
    try:
        sha = open(os.path.join(root, "tools", ".git_sha")).read().strip()
    except OSError:
        sha = "unknown"
for arg in sys.argv[2:]:
    key, path = arg.split("=", 1)
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"== (\S.*?)\s+dispatches=(\d+)\s+mean duration \(profiled\) = ([0-9.]+) us", line)
        if m:
            cur = {"dispatches": int(m.group(2)), "mean_us": float(m.group(3)), "cnt": {}}
            kernels[m.group(1)] = cur
            continue
        m = re.match(r"\s+(\w+)\s+mean ([0-9.e+\-]+)", line)
        if m and cur is not None:
            cur["cnt"][m.group(1)] = float(m.group(2))
    if not kernels:
        continue
    name = max(kernels, key=lambda k: kernels[k]["dispatches"] * kernels[k]["mean_us"])
    k = kernels[name]
    c = k["cnt"]
    simd = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
    frac = lambda n, mul=1.0: round(mul * c[n] / simd, 4) if simd and n in c else None
    out[key] = {"kernel": name, "kernel_us_profiled": k["mean_us"], "sq_insts_mfma": c.get("SQ_INSTS_MFMA"), "sq_insts_valu": c.get("SQ_INSTS_VALU"),
                "mfma_busy": frac("SQ_VALU_MFMA_BUSY_CYCLES"), "valu_busy": frac("SQ_ACTIVE_INST_VALU", 4.0), "coexec": frac("SQ_VALU_MFMA_COEXEC_CYCLES"),
                "lds_insts": c.get("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
                "fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"), "git": sha,
                "share_of_gpu_time": round(k["dispatches"] * k["mean_us"] / sum(v["dispatches"] * v["mean_us"] for v in kernels.values()), 4)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
