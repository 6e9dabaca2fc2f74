import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem
N = 600
for K in (65, 68, 81, 84, 101, 112, 113, 128):
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
    plan = engine.DevicePlan(P, M)
    for _ in range(20): plan.run(0.1)
    plan.sync(); t = time.perf_counter()
    for _ in range(10): plan.run(0.1)
    plan.sync(); dt = (time.perf_counter() - t) / 10
    L = plan.L
    def emd(): _lib.check(L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, 2, 0, N, 1, plan.dE, plan.dIt, None))
    emd(); plan.sync(); t = time.perf_counter()
    for _ in range(3): emd()
    plan.sync(); de = (time.perf_counter() - t) / 3
    print("K=%3d sinkhorn %.3f ms  exact %.2f ms" % (K, dt * 1e3, de * 1e3), flush=True)
    plan.close()
