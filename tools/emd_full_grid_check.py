"""Every pair of a full exact-OT grid against the CPU network simplex: usage emd_full_grid_check.py [config|NxK[:row step] ...]
(default row step: 1 up to 600 patients, 16 beyond; `c4:1` checks all 4 000 000 pairs of c4, ~1 min on 64 host threads)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
import os
for spec in (sys.argv[1:] or ["c2", "c3"]):
    cfg, _, step_arg = spec.partition(":")
    if "x" in cfg:          # "NxK": the cohorts of tools/k_sweep.py (8 PCA dims, 200 cells per patient)
        n_, k_ = (int(t) for t in cfg.split("x"))
        P, M = make_problem(n_, k_, 8, seed=k_, cells_per_patient=200)
    else:
        P, M = make_problem(**CONFIGS[cfg])
    N, K = P.shape
    step = int(step_arg) if step_arg else (1 if N <= 600 else 16)
    n_thr = 16 if step > 1 or N <= 600 else min(64, os.cpu_count() or 16)
    Eg = engine.emd_grid(P, M)
    t = time.perf_counter()
    Eo = O.emd_grid(P, M, row_step=step, n_threads=n_thr, fast="ns")
    dt = time.perf_counter() - t
    d = np.abs(Eg[::step] - Eo)
    print("%s N=%d K=%d: %d pairs against the network simplex (%.1f s on %d host threads): max|d| %.2e, mean %.2e; symmetric %s, zero diagonal %.1e" % (
        cfg, N, K, Eo.size, dt, n_thr, d.max(), d.mean(), np.array_equal(Eg, Eg.T), np.abs(np.diag(Eg)).max()), flush=True)
    assert d.max() <= 1e-12
