"""Pairs/s of the Sinkhorn grid (reg 0.1) and of the exact grid across the cohort size N at K = 50 / 30: looks for cliffs.
usage: python tools/n_sweep.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem
for K in (30, 50):
    for N in (16, 50, 100, 200, 400, 600, 1000, 2000, 4000):
        P, M = make_problem(N, K, 8, seed=N + K, cells_per_patient=200)
        plan = engine.DevicePlan(P, M)
        reps = 20 if N <= 1000 else 3
        for _ in range(3): plan.run(0.1)
        plan.sync(); t = time.perf_counter()
        for _ in range(reps): plan.run(0.1)
        plan.sync(); dt = (time.perf_counter() - t) / reps
        upd = plan.fetch()[1]["iters"].mean()
        L = plan.L
        def emd(): _lib.check(L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, 2, 0, N, 1, plan.dE, plan.dIt, None))
        emd(); plan.sync(); t = time.perf_counter()
        for _ in range(max(1, reps // 4)): emd()
        plan.sync(); de = (time.perf_counter() - t) / max(1, reps // 4)
        print("K=%2d N=%4d  sinkhorn %8.3f ms  %.3e pairs/s (%.1f updates/pair)   exact %8.3f ms  %.3e pairs/s" % (
            K, N, dt * 1e3, N * N / dt, upd, de * 1e3, N * N / de), flush=True)
        plan.close()
