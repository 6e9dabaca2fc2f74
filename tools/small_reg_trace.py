"""c3 at reg 0.01 under AUTO, a few calls: run under rocprofv3 --kernel-trace --stats to see where the time goes."""
import sys
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
P, M = make_problem(**CONFIGS["c3"])
plan = engine.DevicePlan(P, M)
for _ in range(6):
    plan.run(0.01)
plan.sync()
E, info = plan.fetch()
it = info["iters"]
print("updates per pair: mean %.1f, capped %d of %d; f64 pairs %d" % (it.mean(), (it >= 1000).sum(), it.size, ((info["flags"] & 16) > 0).sum()))
print("histogram of updates:", np.histogram(it, bins=[0, 100, 200, 400, 600, 800, 999, 1001])[0])
plan.close()
