#!/usr/bin/env python3
"""Dev probe (GPU box): parity of the HIP pair grid against the CPU oracle on sampled rows + timing."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem, CONFIGS

print("devices:", _lib.device_count(), _lib.device_name())
cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c1", "c2", "c3"]
regs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1.0, 0.1, 0.01]
precs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fp32", "fp64"]
for cfg in cfgs:
    P, M = make_problem(**CONFIGS[cfg])
    N, K = P.shape
    step = max(1, N // 12)
    for reg in regs:
        Eo, io = O.sinkhorn_grid(P, M, reg, row_step=step, n_threads=os.cpu_count(), return_info=True)
        for prec in precs:
            t = time.time()
            Eg, ig = engine.sinkhorn_grid(P, M, reg, precision=prec, row_step=step, return_info=True)
            dt = time.time() - t
            d = np.abs(Eg - Eo)
            conv = (io["flags"] & 1) > 0
            same_it = np.mean(ig["iters"] == io["iters"])
            print("%s reg=%g %s: max|d|=%.3e (conv %.3e, capped %.3e) iters-equal=%.4f nan=%d absorbed=%d absorb_last gpu/oracle=%d/%d  t=%.3fs" % (
                cfg, reg, prec, d.max(), d[conv].max() if conv.any() else 0, d[~conv].max() if (~conv).any() else 0,
                same_it, int(((ig["flags"] & 2) > 0).sum()), int(((ig["flags"] & 8) > 0).sum()),
                int(((ig["flags"] & 4) > 0).sum()), int(((io["flags"] & 4) > 0).sum()), dt))
            if d.max() > 1e-5:
                w = np.unravel_index(np.argmax(d), d.shape)
                print("   worst at", w, "gpu", Eg[w], "oracle", Eo[w], "iters", ig["iters"][w], io["iters"][w], "flags", ig["flags"][w], io["flags"][w])
    # full-grid timing through the resident plan
    plan = engine.DevicePlan(P, M)
    for reg in regs:
        for prec in precs:
            plan.run(reg, precision=prec); plan.sync()
            t = time.time()
            reps = 3
            for _ in range(reps):
                plan.run(reg, precision=prec)
            plan.sync()
            dt = (time.time() - t) / reps
            E, info = plan.fetch()
            print("%s FULL reg=%g %s: %.3f ms/matrix  %.3e pairs/s  mean iters %.1f  asym %.2e" % (
                cfg, reg, prec, dt * 1e3, N * N / dt, info["iters"].mean(), np.abs(E - E.T).max()))
    plan.close()
