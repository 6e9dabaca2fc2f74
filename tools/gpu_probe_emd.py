#!/usr/bin/env python3
"""Dev probe (GPU box): exact-EMD kernel vs CPU oracle + timing."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c1", "c2", "c3"]
for cfg in cfgs:
    P, M = make_problem(**CONFIGS[cfg])
    N, K = P.shape
    step = max(1, N // 10)
    t = time.time(); Eo = O.emd_grid(P, M, row_step=step, n_threads=os.cpu_count()); to = time.time() - t
    t = time.time(); Eg, info = engine.emd_grid(P, M, row_step=step, mode="all", return_info=True); tg = time.time() - t
    print("%s rows/%d: max|d|=%.3e  n_aug mean %.1f max %d  oracle %.2fs (%d thr)  gpu %.3fs" % (
        cfg, step, np.abs(Eg - Eo).max(), info["n_aug"].mean(), info["n_aug"].max(), to, os.cpu_count(), tg))
    if N <= 700:
        t = time.time(); E = engine.emd_grid(P, M); tg = time.time() - t
        print("%s FULL (mirror): %.3fs  %.3e pairs/s  asym %.1e diag %.1e" % (cfg, tg, N * N / tg, np.abs(E - E.T).max(), np.abs(np.diag(E)).max()))
        t = time.time(); E2 = engine.emd_grid(P, M, mode="all"); tg = time.time() - t
        print("%s FULL (all):    %.3fs  %.3e pairs/s  |all-mirror| %.1e" % (cfg, tg, N * N / tg, np.abs(E - E2).max()))
