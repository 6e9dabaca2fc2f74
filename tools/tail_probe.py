#!/usr/bin/env python3
"""Is a launch waiting for its longest pairs?  The c3 grid with POT's cap lowered (num_iter_max = 1000, 301, 141, 101, 61): the work
barely changes (p99.9 of the update counts is 141), the longest serial chain does."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches
from pilot_amd.synthetic import CONFIGS, make_problem
for name in (sys.argv[1:] or ["c3"]):
    P, M = make_problem(**CONFIGS[name])
    plan = engine.DevicePlan(P, M); plan.enable_timing(True)
    for dbg in (None, "512"):
      switches.set("PILOT_OT_DEBUG", dbg)
      print("PILOT_OT_DEBUG", dbg, "(512: exact duplicates in the tiles instead of one wave per pair)")
      for cap in (1000, 301, 141, 101, 61, 41):
        for _ in range(3): plan.run(0.1, num_iter_max=cap)
        plan.sync()
        n = 20 if name != "c4" else 4
        for _ in range(n): plan.run(0.1, num_iter_max=cap)
        plan.sync()
        m, tr = plan.kernel_times_ms(n)
        it = plan.fetch()[1]["iters"]
        d = np.diag(it[:, :it.shape[0]]) if it.shape[0] == it.shape[1] else it[:0]
        print("   diagonal: mean %.1f max %d | off-diagonal max %d |" % (d.mean(), d.max(), (it - np.diag(np.diag(it))).max()), end=" ")
        print("%s cap %4d: main kernel %.3f ms, updates per pair mean %.2f max %d, total %.3e" % (name, cap, m.mean(), it.mean(), it.max(), it.sum()), flush=True)
    switches.set("PILOT_OT_DEBUG", None)
    plan.close()
