#!/usr/bin/env python3
"""BASELINE config 3: Sinkhorn reg sweep 0.01 / 0.1 / 1.0 on the 600 x 50 synthetic, f32 vs f64 vs auto, device-resident
timing (whole call, wall clock over `reps` launches) + parity of a row sample against the fp64 oracle.
usage: tools/reg_sweep.py [cfg=c3] [reps=5]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P, Mn = make_problem(**CONFIGS[cfg])
N, K = P.shape
plan = engine.DevicePlan(P, Mn)
plan.enable_timing(True)
step = max(1, N // 6)
nthr = os.cpu_count() or 1
print("%s: N=%d K=%d; oracle sample = rows ::%d" % (cfg, N, K, step))
# the oracle passes first: their OpenMP workers spin for a while after a parallel region and starve the HIP runtime's
# helper threads, which shows up as 10-80 ms stalls in the first GPU calls that follow
refs = {reg: O.sinkhorn_grid(P, Mn, reg, row_begin=0, row_end=N, row_step=step, n_threads=nthr, return_info=True)
        for reg in (1.0, 0.1, 0.01)}
time.sleep(2.0)
for reg in (1.0, 0.1, 0.01):
    ref, rinfo = refs[reg]
    for prec in ("f32", "f64", "auto"):
        plan.run(reg, precision=prec); plan.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.run(reg, precision=prec)
        plan.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        emd, info = plan.fetch()
        main, track = plan.kernel_times_ms()
        line = "reg=%-5g %-4s %8.3f ms/matrix  %.3e pairs/s  main %.3f ms  track %.3f ms  mean updates %.1f  capped %d  absorbed %d  nan %d" % (
            reg, prec, ms, N * N / ms * 1e3, main[-reps:].mean(), track[-reps:].mean(), info["iters"].mean(),
            int((info["flags"] & 1 == 0).sum()), int((info["flags"] & 8 != 0).sum()), int((info["flags"] & 2 != 0).sum()))
        if ref is not None:
            d = np.abs(emd[::step] - ref)
            conv = (rinfo["flags"] & 1) != 0
            line += "  max|d| vs oracle %.2e (converged pairs %.2e)" % (d.max(), d[conv].max() if conv.any() else 0.0)
        print(line, flush=True)
