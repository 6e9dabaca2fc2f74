"""GPU probe: how far beyond max(M)/reg = 11.5 does the fp16-split configuration stay accurate?  Forces f16x2 (raw precision)
at smaller reg on c3 / c4 row samples and compares with the oracle and with bf16x3.  usage: python tools/f16x2_range_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
__import__("sys").path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__))); import switches; switches.set("PILOT_OT_H_MAX_COST_OVER_REG", "40")
from oracle import oracle as O
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem

for cfg, step in (("c3", 30), ("c2", 2), ("c4", 250)):
    P, M = make_problem(**CONFIGS[cfg])
    for reg in (0.09, 0.08, 0.07, 0.0625, 0.055, 0.05, 0.045, 0.04):
        Eo, io = O.sinkhorn_grid(P, M, reg, row_step=step, n_threads=64, return_info=True)
        row = "%s max(M)/reg %5.1f:" % (cfg, M.max() / reg)
        for prec in ("bf16x3", "f16x2"):
            E, inf = engine.sinkhorn_grid(P, M, reg, precision=prec, row_step=step, return_info=True)
            same = inf["iters"] == io["iters"]
            row += "  %s max|dE| %.2e (same-count pairs %.2e, %.3f of them; capped %d/%d)" % (
                prec, np.abs(E - Eo).max(), np.abs(E - Eo)[same].max() if same.any() else 0.0, same.mean(),
                int((inf["iters"] >= 1000).sum()), int((io["iters"] >= 1000).sum()))
        print(row, flush=True)
P, M = make_problem(**CONFIGS["c3"])
for reg in (0.07, 0.0625):
    for prec in ("bf16x3", "f16x2"):
        plan = engine.DevicePlan(P, M)
        for _ in range(10): plan.run(reg, precision=prec)
        plan.sync(); t = time.perf_counter()
        for _ in range(10): plan.run(reg, precision=prec)
        plan.sync(); print("c3 reg %g %s: %.3f ms per matrix" % (reg, prec, (time.perf_counter() - t) / 10 * 1e3), flush=True)
        plan.close()
