"""Every ordered pair of a full Sinkhorn grid against the fp64 oracle (default precision): usage sinkhorn_full_grid_check.py [config|NxK:reg[:row_step] ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from pilot_amd import _lib, engine
from pilot_amd.synthetic import CONFIGS, make_problem
for spec in (sys.argv[1:] or ["c3:1.0", "c3:0.1", "c3:0.01", "c4:0.1:16"]):
    f = spec.split(":")
    cfg, reg, step = f[0], float(f[1]), int(f[2]) if len(f) > 2 else 1
    if "x" in cfg:          # "NxK": the cohorts of tools/k_sweep.py (8 PCA dims, 200 cells per patient)
        n_, k_ = (int(t) for t in cfg.split("x"))
        P, M = make_problem(n_, k_, 8, seed=k_, cells_per_patient=200)
    else:
        P, M = make_problem(**CONFIGS[cfg])
    N, K = P.shape
    Eg, ig = engine.sinkhorn_grid(P, M, reg, row_step=step, return_info=True)
    t = time.perf_counter()
    Eo, io = O.sinkhorn_grid(P, M, reg, row_step=step, n_threads=(16 if Eg.size <= 400000 else min(64, __import__('os').cpu_count() or 16)), return_info=True)
    dt = time.perf_counter() - t
    last_o, last_g = (io["flags"] & O.FLAG_ABSORB_ON_LAST) > 0, (ig["flags"] & _lib.FLAG_ABSORB_LAST) > 0
    last = last_o | last_g
    d = np.abs(Eg - Eo)
    capped = io["iters"] >= 1000
    print("%s N=%d K=%d reg %g: %d pairs (oracle %.1f s on the host): max|gpu - oracle| %.3e (mean %.1e) outside the %d pairs POT returns "
          "scaled by 1/K^2; oracle capped %d, absorbed %d; updates gpu %.1f / oracle %.1f; gpu at the oracle's check or earlier: %s; f64 pairs %d"
          % (cfg, N, K, reg, Eo.size, dt, d[~last].max(), d[~last].mean(), last.sum(), capped.sum(),
             ((io["flags"] & O.FLAG_ABSORBED) > 0).sum(), ig["iters"].mean(), io["iters"].mean(), bool(np.all(ig["iters"] <= io["iters"])),
             ((ig["flags"] & _lib.FLAG_F64) > 0).sum()), flush=True)
    if last.any():
        both, only_o, only_g = last_o & last_g, last_o & ~last_g, last_g & ~last_o
        same = ig["iters"] == io["iters"]
        print("   ... of those %d pairs: flagged on both sides %d (max|gpu - oracle| there %.3e), by the oracle alone %d, by the GPU alone %d; "
              "one-sided flags among pairs with the same update count on both sides: %d"
              % (last.sum(), both.sum(), np.abs(Eg - Eo)[both].max() if both.any() else 0.0, only_o.sum(), only_g.sum(),
                 ((last_o ^ last_g) & same).sum()), flush=True)
        # a one-sided flag: one side returned cost / K^2 (absorption on ITS last update), the other the cost -- accounted for as such
        if only_o.any():
            print("   ... oracle alone (the GPU stopped at an earlier check and returned the unscaled cost): max|gpu - K^2 oracle| = %.3e over %d pairs"
                  % (np.abs(Eg - K * K * Eo)[only_o].max(), only_o.sum()), flush=True)
        if only_g.any():
            print("   ... GPU alone: max|K^2 gpu - oracle| = %.3e over %d pairs" % (np.abs(K * K * Eg - Eo)[only_g].max(), only_g.sum()), flush=True)
        print("   ... one-sided share of the grid: %.2e (tests bound it at 1e-4)" % ((only_o.sum() + only_g.sum()) / Eo.size), flush=True)
    later = ig["iters"] > io["iters"]
    if later.any():
        absorbed = (io["flags"] & O.FLAG_ABSORBED) > 0
        print("   ... %d pairs stop at a LATER check than the oracle (%d of them tau-absorbing pairs on the tracking kernel; the latest by %d updates); "
              "max|gpu - oracle| among them %.3e" % (later.sum(), (later & absorbed).sum(), (ig["iters"] - io["iters"])[later].max(), d[later & ~last].max()), flush=True)
