#!/usr/bin/env python3
"""Per-rank time of a 1/G row shard (what one GPU of G does), on one GPU: the strong-scaling floor."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
P, M = make_problem(**CONFIGS["c3"])
pl = engine.DevicePlan(P, M); pl.enable_timing(True)
for step in (1, 2, 4, 8):
    for _ in range(3): pl.run(0.1, precision="fp32", row_begin=0, row_step=step)
    pl.sync()
    t = time.perf_counter()
    for _ in range(10): pl.run(0.1, precision="fp32", row_begin=0, row_step=step)
    pl.sync(); dt = (time.perf_counter() - t) / 10
    a, b = pl.kernel_times_ms(10)
    print("PILOT_OT_DEBUG=%s rows 0::%d (%d pairs): main kernel %.3f ms, whole call %.3f ms" % (os.environ.get("PILOT_OT_DEBUG", "0"), step, 360000 // step, a.mean(), dt * 1e3))
