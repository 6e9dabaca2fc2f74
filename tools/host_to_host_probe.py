#!/usr/bin/env python3
"""engine.sinkhorn_grid, host arrays in -> host matrix out (c3), stage stamps from inside the library (PILOT_OT_HOST_TRACE) for 1 .. 4
threads copying the result out of the pinned block (PILOT_OT_FETCH_THREADS)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
P, M = make_problem(**CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"])
for thr in (4, 3, 2, 1):
    switches.set("PILOT_OT_FETCH_THREADS", str(thr))
    for _ in range(3): E = engine.sinkhorn_grid(P, M, 0.1)
    t = time.perf_counter()
    for _ in range(20): E = engine.sinkhorn_grid(P, M, 0.1)
    dt = (time.perf_counter() - t) / 20
    print("%d copying thread(s): host to host %.3f ms" % (thr, dt * 1e3), flush=True)
    switches.set("PILOT_OT_HOST_TRACE", "1")
    for _ in range(3): engine.sinkhorn_grid(P, M, 0.1)
    switches.set("PILOT_OT_HOST_TRACE", None)
