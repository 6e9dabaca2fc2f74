#!/usr/bin/env python3
"""hipGraph replay vs ordinary launches for a repeated Sinkhorn call (pilot_ot_plan_enable_graph): step time and bit-identity."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
for cfg, reg, reps in (("c2", 0.1, 400), ("c3", 0.1, 200), ("c3", 0.01, 10), ("c4", 0.1, 5)):
    P, M = make_problem(**CONFIGS[cfg])
    pl = engine.DevicePlan(P, M)
    res = {}
    for graph in (0, 1, 0, 1):
        pl.enable_graph(bool(graph))
        for _ in range(max(3, reps // 2)): pl.run(reg)
        pl.sync()
        t = time.perf_counter()
        for _ in range(reps): pl.run(reg)
        pl.sync(); dt = (time.perf_counter() - t) / reps
        E, info = pl.fetch()
        if graph in res:
            assert np.array_equal(res[graph][1], E, equal_nan=True)
        res[graph] = (dt, E, info["iters"])
        print("%s reg %g graph=%d: %.4f ms per call" % (cfg, reg, graph, dt * 1e3), flush=True)
    print("   identical results with and without graph:", np.array_equal(res[0][1], res[1][1], equal_nan=True) and np.array_equal(res[0][2], res[1][2]))
    pl.close()
