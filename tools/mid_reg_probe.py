"""GPU probe: the regs between the fp16-split range and the two-band path (16 < max(M)/reg <= 60) on c3: time of the fast pass and of the
tracking pass, share of pairs that tau-absorb.  usage: python tools/mid_reg_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
P, M = make_problem(**CONFIGS["c3"])
plan = engine.DevicePlan(P, M)
plan.enable_timing(True)
for reg in (0.1, 0.08, 0.07, 0.0625, 0.055, 0.05, 0.04, 0.03, 0.025, 0.02, 0.0175):
    for _ in range(3): plan.run(reg)
    plan.sync(); t = time.perf_counter()
    for _ in range(5): plan.run(reg)
    plan.sync(); ms = (time.perf_counter() - t) / 5 * 1e3
    E, info = plan.fetch()
    main, track = plan.kernel_times_ms()
    print("reg %-7g max(M)/reg %5.1f: %7.3f ms per matrix  fast pass %7.3f  tracking pass %7.3f  mean updates %6.1f  absorbed %.3f  capped %.4f  f64 %d" % (
        reg, 1 / reg, ms, main[-5:].mean(), track[-5:].mean(), info["iters"].mean(), ((info["flags"] & 8) != 0).mean(),
        (info["iters"] >= 1000).mean(), int(((info["flags"] & 16) != 0).sum())), flush=True)
