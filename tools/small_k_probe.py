"""Small K (pathomics cohorts have single-digit K): per-kernel times, update counts and flags of the Sinkhorn grid, and the
exact grid, at N = 600.  usage: small_k_probe.py [K ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem
N = 600
Ks = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6, 7, 8, 12]
for K in Ks:
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
    plan = engine.DevicePlan(P, M); plan.enable_timing(True)
    for _ in range(10): plan.run(0.1)
    plan.sync()
    a, b = plan.kernel_times_ms(10)
    plan.enable_timing(False)
    t = time.perf_counter()
    for _ in range(10): plan.run(0.1)
    plan.sync(); dt = (time.perf_counter() - t) / 10
    _, info = plan.fetch()
    it = info["iters"]; fl = info["flags"]
    L = plan.L
    def emd(): _lib.check(L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, 2, 0, N, 1, plan.dE, plan.dIt, None))
    for _ in range(3): emd()      # (the first exact-mode launches of a process pay one-time costs)
    plan.sync()
    t = time.perf_counter()
    for _ in range(3): emd()
    plan.sync(); de = (time.perf_counter() - t) / 3
    hist = {int(f): int((fl == f).sum()) for f in np.unique(fl)}
    print("K=%2d call %.3f ms main %.3f ms track %.3f ms | updates mean %.1f max %d | capped %d of %d | flags %s | exact %.2f ms | max(M)/reg %.1f" % (
        K, dt * 1e3, a.mean(), b.mean(), it.mean(), it.max(), (it >= 1000).sum(), it.size, hist, de * 1e3, M.max() / 0.1), flush=True)
    plan.close()
