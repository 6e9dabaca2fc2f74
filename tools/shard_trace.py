#!/usr/bin/env python3
"""Run a 1/G row shard of c3 a few times (for rocprofv3 --kernel-trace; see tools/step_timeline.py). usage: shard_trace.py G [cfg]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P, M = make_problem(**CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c3"])
pl = engine.DevicePlan(P, M)
for _ in range(6):
    pl.run(0.1, precision="fp32", row_begin=0, row_step=G)
pl.sync()
