#!/usr/bin/env python3
"""Per-rank time of a 1/G row shard (what one GPU of G does), on one GPU: the strong-scaling floor."""
import sys, time, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
floor = {}
for cfg, reps in (("c3", 20), ("c4", 3)):
    floor[cfg] = {}
    P, M = make_problem(**CONFIGS[cfg])
    N = P.shape[0]
    pl = engine.DevicePlan(P, M); pl.enable_timing(True)
    for _ in range(150 if cfg == "c3" else 4): pl.run(0.1)          # clocks
    pl.sync()
    _, info = pl.fetch()
    it = info["iters"]
    print("%s: updates per pair mean %.1f, p99 %d, p99.9 %d, max %d; max off-diagonal %d" % (
        cfg, it.mean(), np.percentile(it, 99), np.percentile(it, 99.9), it.max(), (it - np.diag(np.diag(it))).max()))
    for step in (1, 2, 4, 8):
        for _ in range(3): pl.run(0.1, row_begin=0, row_step=step)
        pl.sync()
        t = time.perf_counter()
        for _ in range(reps): pl.run(0.1, row_begin=0, row_step=step)
        pl.sync(); dt = (time.perf_counter() - t) / reps
        a, b = pl.kernel_times_ms(reps)
        floor[cfg][str(step)] = {"kernel_ms": round(float(a.mean()), 4), "call_ms": round(dt * 1e3, 4)}
        print("%s rows 0::%d (%d pairs): main kernel %.3f ms, track %.3f ms, whole call %.3f ms" % (cfg, step, N * N // step, a.mean(), b.mean(), dt * 1e3))
    pl.close()
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        json.dump({"what": "one GPU solving rows 0::G of the reg 0.1 grid (what each of G GPUs does): kernel and whole-call ms", "floor": floor}, fh)
