#!/usr/bin/env python3
"""Wall time of every device-resident call in a long run (spots stalls that averages hide). usage: call_jitter.py [n=300]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
P, M = make_problem(**CONFIGS["c3"])
pl = engine.DevicePlan(P, M)
for prec in ("fp32", "fp64", "fp32"):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); pl.run(0.1, precision=prec); pl.sync(); ts.append((time.perf_counter() - t) * 1e3)
    ts = np.array(ts)
    print("%s: median %.3f ms  p99 %.3f  max %.3f  first five %s  calls > 2x median: %s" % (
        prec, np.median(ts), np.percentile(ts, 99), ts.max(), np.round(ts[:5], 2), np.nonzero(ts > 2 * np.median(ts))[0][:20]), flush=True)
