import sys, time
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem
N, K = 600, 2
P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
plan = engine.DevicePlan(P, M)
L = plan.L
def emd(): _lib.check(L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, 2, 0, N, 1, plan.dE, plan.dIt, None))
def t_emd(tag):
    ts = []
    for _ in range(6):
        t = time.perf_counter(); emd(); plan.sync(); ts.append((time.perf_counter() - t) * 1e3)
    print(tag, ["%.3f" % x for x in ts])
t_emd("fresh plan      ")
for _ in range(5): plan.run(0.1)
plan.sync()
t_emd("after sinkhorn  ")
