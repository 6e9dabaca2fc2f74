#!/usr/bin/env python3
"""The fast Sinkhorn kernel with ONE resident workgroup per CU (one wave per SIMD) against the default: does a second wave on a
SIMD double the throughput (the update is a latency chain) or not (the waves contend for the pipes)?
usage: k2_occupancy_probe.py [c3|c4 ...]   (PILOT_OT_DEBUG bits 4-6: resident workgroups per CU, an experiment switch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import switches
for name in (sys.argv[1:] or ["c3", "c4"]):
    P, M = make_problem(**CONFIGS[name])
    plan = engine.DevicePlan(P, M)
    plan.enable_timing(True)
    reps = 20 if name != "c4" else 4
    for wg in (0, 1, 2, 3):
        switches.set("PILOT_OT_DEBUG", str(wg << 4) if wg else None)
        for _ in range(3): plan.run(0.1)
        plan.sync()
        t = time.perf_counter()
        for _ in range(reps): plan.run(0.1)
        plan.sync(); dt = (time.perf_counter() - t) / reps
        m, tr = plan.kernel_times_ms(reps)
        print("%s: %s resident workgroup(s) per CU: step %.3f ms, main kernel %.3f ms" % (name, wg or "default", dt * 1e3, m.mean()), flush=True)
    switches.set("PILOT_OT_DEBUG", None)
    plan.close()
