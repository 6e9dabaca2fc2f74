#!/usr/bin/env python3
"""The exact-OT grid at one shape, a few calls (workload of the small-K exact-mode profiles).
usage: emd_point.py K [N]   |   emd_point.py real   (the Kidney_IgAN_G cohort of tests/golden)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from pilot_amd import engine, _lib
sys.path.insert(0, os.path.join(R, "tools"))
import switches
switches.apply_from_env()                      # TOOL_SWITCHES="PILOT_OT_EMD_MULTI=0" (tools/emd_multi_pmc.sh)
if sys.argv[1] == "real":
    from conftest import GOLDEN_REAL, load_golden
    g = load_golden(GOLDEN_REAL)
    P = g["proportions"]; M = g["cost"] / g["cost"].max()
else:
    from pilot_amd.synthetic import make_problem
    K = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
N, K = P.shape
plan = engine.DevicePlan(P, M)
def emd(): _lib.check(plan.L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, 2, 0, N, 1, plan.dE, plan.dIt, None))
for _ in range(3): emd()
plan.sync()
t = time.perf_counter()
for _ in range(5): emd()
plan.sync(); dt = (time.perf_counter() - t) / 5
n_aug = np.empty((N, N), dtype=np.int32)
_lib.check(plan.L.pilot_ot_memcpy_d2h(n_aug.ctypes.data, plan.dIt, 4 * N * N))
iu = np.triu_indices(N)
print("exact grid N=%d K=%d: %.3f ms per matrix, %.1f augmentations per solved pair (max %d)" % (N, K, dt * 1e3, n_aug[iu].mean(), n_aug[iu].max()))
plan.close()
