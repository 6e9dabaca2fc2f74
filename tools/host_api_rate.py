#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer API (numpy in, numpy out) -- noted in DESIGN.md, never bench `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
P, M = make_problem(**CONFIGS["c3"])
N = P.shape[0]
engine.sinkhorn_grid(P, M, 0.1)
for name, fn in (("sinkhorn reg=0.1 f32 (host API)", lambda: engine.sinkhorn_grid(P, M, 0.1)),
                 ("sinkhorn reg=0.1 f64 (host API)", lambda: engine.sinkhorn_grid(P, M, 0.1, precision="fp64")),
                 ("exact EMD (host API, mirror)", lambda: engine.emd_grid(P, M))):
    t = time.perf_counter()
    while time.perf_counter() - t < 1.0: fn()      # first calls run 10x slower (allocation, clocks ramping up on an idle box)
    t = time.perf_counter(); reps = 10
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t) / reps
    print("%-36s %.3f ms per matrix  %.3e pairs/s" % (name, dt * 1e3, N * N / dt))
