#!/usr/bin/env python3
"""One K of the K sweep (N = 600, reg 0.1, precision auto), a few calls: the workload the K = 80 / 96 profiles are taken on.
usage: k_point.py K [reg] [precision]"""
import sys, time
sys.path.insert(0, ".")
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem
K = int(sys.argv[1]); reg = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
prec = sys.argv[3] if len(sys.argv) > 3 else "auto"
N = 600
P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
plan = engine.DevicePlan(P, M)
plan.enable_timing(True)
for _ in range(10): plan.run(reg, precision=prec)
plan.sync()
t = time.perf_counter()
for _ in range(10): plan.run(reg, precision=prec)
plan.sync(); dt = (time.perf_counter() - t) / 10
m, tr = plan.kernel_times_ms(10)
it = plan.fetch()[1]["iters"]
flop = float(it.sum()) * (4 * K * K + 2 * K)
print("K=%d reg %g %s: step %.3f ms, kernel %.3f ms, mean updates %.1f, %.1f TF/s algorithmic" % (K, reg, prec, dt * 1e3, m.mean(), it.mean(), flop / (m.mean() * 1e-3) / 1e12))
plan.close()
