#!/usr/bin/env python3
"""emd_instr.json / cellw2_traffic.json for bench.py: per-pair instruction counts and HBM traffic of the exact-OT kernels, and the
HBM traffic of the cell-level kernel, from the rocprofv3 PMC summaries tools/profile_pmc.sh / profile_pmc_scalar.sh wrote
(FETCH_SIZE and WRITE_SIZE in separate passes, KB; FETCH_SIZE doubled per MI355X_MICROARCH.md), stamped with the git revision.
usage: make_emd_instr_json.py <dir with rocprofv3_pmc_summary_emd_<cfg>.txt [+ rocprofv3_pmc_scalar_emd_<cfg>.txt], rocprofv3_pmc_summary_cellw2_<shape>.txt>"""
import glob, json, os, re, subprocess, sys
d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"== (\S.*?)\s+dispatches=(\d+)\s+mean duration \(profiled\) = ([0-9.]+) us(?:\s+max ([0-9.]+) us)?", line)
        if m:
            cur = out.setdefault(m.group(1), {"duration_us": float(m.group(3)), "max_duration_us": float(m.group(4) or m.group(3))})
            continue
        m = re.match(r"\s+(\w+)\s+mean ([0-9.e+-]+)(?:\s+\(n=\d+\)\s+max ([0-9.e+-]+))?", line)
        if m and cur is not None:
            cur.setdefault(m.group(1), float(m.group(2)))
            cur.setdefault("max_" + m.group(1), float(m.group(3) or m.group(2)))
    return out

def sha():
    try:
        s = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        s = ""
    if not s:
        try:
            s = open(os.path.join(root, "tools", ".git_sha")).read().strip()
        except OSError:
            s = "unknown"
    return s

SOLVED = {"c3": 600 * 601 // 2, "c4": 2000 * 2001 // 2, "c2": 100 * 101 // 2, "kidney": 634 * 635 // 2}
emd = {"_comment": "exact-OT pair grid (symmetric cost: the j >= i pairs are solved, the rest mirrored): rocprofv3 --pmc means per launch "
                   "divided by the solved pairs; traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB counters); 1x MI355X"}
for path in sorted(glob.glob(os.path.join(d, "rocprofv3_pmc_summary_emd_*.txt"))):
    cfg = re.match(r"rocprofv3_pmc_summary_emd_(\w+)\.txt", os.path.basename(path)).group(1)
    if cfg not in SOLVED:
        continue
    ks = parse(path)
    sca = os.path.join(d, "rocprofv3_pmc_scalar_emd_%s.txt" % cfg)
    ks2 = parse(sca) if os.path.exists(sca) else {}
    name = max((k for k in ks if k.startswith("emd_") and "mirror" not in k), key=lambda k: ks[k]["duration_us"], default=None)
    if not name:
        continue
    c = dict(ks2.get(name, {}), **ks[name])
    n = SOLVED[cfg]
    per = lambda k: round(c[k] / n, 1) if k in c else None
    simd = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
    emd[cfg] = {"kernel": name, "solved_pairs": n, "kernel_us_rocprofv3": c["duration_us"],
                "valu_per_pair": per("SQ_INSTS_VALU"), "salu_per_pair": per("SQ_INSTS_SALU"), "branch_per_pair": per("SQ_INSTS_BRANCH"),
                "lds_per_pair": per("SQ_INSTS_LDS"), "waves": c.get("SQ_WAVES"),
                "valu_busy": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / simd, 4) if simd and "SQ_ACTIVE_INST_VALU" in c else None,
                "fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"),
                "traffic_bytes": int(1024 * (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"])) if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
                "git": sha()}
json.dump(emd, open(os.path.join(d, "emd_instr.json"), "w"), indent=1)
cw = {"_comment": "cell-level W2 grid (BASELINE config 5, an extension): HBM traffic of cell_w2_kernel per launch, 2 x FETCH_SIZE + WRITE_SIZE "
                  "(rocprofv3 --pmc, separate passes, KB counters); 1x MI355X"}
for path in sorted(glob.glob(os.path.join(d, "rocprofv3_pmc_summary_cellw2_*.txt"))):
    shape = re.match(r"rocprofv3_pmc_summary_cellw2_(\w+)\.txt", os.path.basename(path)).group(1)
    ks = parse(path)
    name = max((k for k in ks if k.startswith("cell_w2_kernel")), key=lambda k: ks[k]["duration_us"], default=None)
    if name and "FETCH_SIZE" in ks[name] and "WRITE_SIZE" in ks[name]:
        c = ks[name]      # (the run has a three-update warm-up launch of one row beside the grid: the grid is the MAXIMUM over the dispatches)
        cw[shape] = {"kernel": name, "fetch_size_kb": c["max_FETCH_SIZE"], "write_size_kb": c["max_WRITE_SIZE"],
                     "traffic_bytes": int(1024 * (2 * c["max_FETCH_SIZE"] + c["max_WRITE_SIZE"])), "kernel_us_rocprofv3": c["max_duration_us"],
                     "note": "FETCH_SIZE counts what the L2s fetched from the fabric -- HBM or the 256 MB Infinity Cache, which holds the whole 128 MB "
                             "cohort of pieces: an upper bound of the HBM bytes",
                     "git": sha()}
if len(cw) > 1:
    json.dump(cw, open(os.path.join(d, "cellw2_traffic.json"), "w"), indent=1)
print(json.dumps(emd, indent=1)); print(json.dumps(cw, indent=1))
