import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pilot_amd import engine
from pilot_amd.synthetic import make_problem
for K in (2, 3, 4, 5, 8):
    P, M = make_problem(600, K, 8, seed=K, cells_per_patient=200)
    plan = engine.DevicePlan(P, M); plan.enable_timing(True)
    for _ in range(10): plan.run(0.1)
    plan.sync()
    a, b = plan.kernel_times_ms(10)
    _, info = plan.fetch()
    it = info["iters"]; fl = info["flags"]
    print("K=%d main %.3f ms track %.3f ms | updates mean %.1f max %d | absorbed %d capped %d of %d" % (K, a.mean(), b.mean(), it.mean(), it.max(), ((fl & 8) > 0).sum(), (it >= 1000).sum(), it.size))
    plan.close()
