#!/usr/bin/env python3
"""Exact duplicates (the diagonal of a grid: a == b) one wave per pair in the leading workgroups of the fast launch, or in the 16-pair
tiles like every other pair (PILOT_OT_DEBUG bit 512)?  Main-kernel time per K at N = 600, and c3 / c4."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
cases = [("600x%d" % K,) + make_problem(600, K, 8, seed=K, cells_per_patient=200) for K in (8, 14, 20, 30, 40, 48, 50, 64)]
cases += [(c,) + make_problem(**CONFIGS[c]) for c in ("c2", "c3")]
for name, P, M in cases:
    plan = engine.DevicePlan(P, M); plan.enable_timing(True)
    out = []
    for rep in range(2):
        for dbg in (None, "512"):
            switches.set("PILOT_OT_DEBUG", dbg)
            for _ in range(4): plan.run(0.1)
            plan.sync()
            for _ in range(20): plan.run(0.1)
            plan.sync()
            m, tr = plan.kernel_times_ms(20)
            out.append("%s %.3f" % ("tiles" if dbg else "one wave per duplicate", m.mean()))
    switches.set("PILOT_OT_DEBUG", None)
    print(name, "main kernel ms:", " | ".join(out), flush=True)
    plan.close()
