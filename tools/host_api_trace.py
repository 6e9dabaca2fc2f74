"""A dozen host-buffer calls in a row with per-call wall time (run it under rocprofv3 --kernel-trace --memory-copy-trace
to see what a slow call spends its time on; see tools/trace_dump.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
P, M = make_problem(**CONFIGS["c3"])
ts = []
for _ in range(12):
    t = time.perf_counter(); engine.sinkhorn_grid(P, M, 0.1, precision="fp32"); ts.append((time.perf_counter() - t) * 1e3)
print("fp32", " ".join("%.2f" % x for x in ts), flush=True)
