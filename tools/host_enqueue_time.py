#!/usr/bin/env python3
"""Host-side cost of enqueueing one device-resident call (does the CPU keep ahead of a 0.4 ms row shard?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
P, M = make_problem(**CONFIGS["c3"])
pl = engine.DevicePlan(P, M); pl.enable_timing(True)
for step in (1, 8):
    for _ in range(5): pl.run(0.1, precision="fp32", row_step=step)
    pl.sync()
    n = 200
    t = time.perf_counter()
    for _ in range(n): pl.run(0.1, precision="fp32", row_step=step)
    t_enq = (time.perf_counter() - t) / n
    pl.sync()
    t_all = (time.perf_counter() - t) / n
    print("rows 0::%d: enqueue %.1f us per call on the host, %.1f us per call end to end" % (step, t_enq * 1e6, t_all * 1e6))
