"""Pairs/s of the Sinkhorn grid (reg 0.1, N = 600) and of the exact grid across K: looks for cliffs between kernel variants."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine
from pilot_amd.synthetic import make_problem
N = 600
for K in (2, 3, 4, 5, 6, 7, 8, 12, 16, 17, 20, 24, 30, 32, 33, 36, 40, 48, 49, 50, 52, 56, 64, 65, 68, 72, 80, 81, 96, 100, 112, 113, 128, 129, 160, 192, 256):
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
    plan = engine.DevicePlan(P, M)
    for _ in range(20): plan.run(0.1)
    plan.sync()
    t = time.perf_counter()
    for _ in range(10): plan.run(0.1)
    plan.sync(); dt = (time.perf_counter() - t) / 10
    _, info = plan.fetch()
    upd = info["iters"].mean()
    L = plan.L
    from pilot_amd import _lib
    mode = 2
    def emd(): _lib.check(L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, mode, 0, N, 1, plan.dE, plan.dIt, None))
    for _ in range(3): emd()      # (the first exact-mode launches of a process pay one-time costs)
    plan.sync()
    t = time.perf_counter()
    for _ in range(3): emd()
    plan.sync(); de = (time.perf_counter() - t) / 3
    flop = upd * (4 * K * K + 2 * K)
    print("K=%3d  sinkhorn %.3f ms (%.1f updates/pair, %.1f TF/s alg.)   exact %.2f ms" % (K, dt * 1e3, upd, N * N * flop / dt / 1e12, de * 1e3), flush=True)
    plan.close()
