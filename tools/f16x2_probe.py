"""GPU probe: the fp16-split configuration (f16x2) against the oracle and against bf16x3 / fp32, c1..c4 + timing at c3/c4.
usage: python tools/f16x2_probe.py [quick]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem

QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"
for cfg, step in ((("c3", 150),) if QUICK else (("c1", 1), ("c2", 5), ("c3", 40), ("c4", 400))):
    P, M = make_problem(**CONFIGS[cfg])
    for reg in ((0.1,) if QUICK else (1.0, 0.1)):
        Eo, io = O.sinkhorn_grid(P, M, reg, row_step=step, n_threads=64, return_info=True)
        out = {}
        for prec in ("fp32", "bf16x3", "f16x2"):
            E, inf = engine.sinkhorn_grid(P, M, reg, precision=prec, row_step=step, return_info=True)
            out[prec] = (E, inf)
            print("%s reg %-4g %-7s max|E-oracle| %.3e  iters<=oracle %s  mean iters %.2f (oracle %.2f)  flags nan %d  err ratio %.3f"
                  % (cfg, reg, prec, np.abs(E - Eo).max(), bool((inf["iters"] <= io["iters"]).all()), inf["iters"].mean(),
                     io["iters"].mean(), int(((inf["flags"] & 2) > 0).sum()),
                     float(np.median(inf["err"][inf["iters"] == io["iters"]] / np.maximum(io["err"][inf["iters"] == io["iters"]], 1e-300)))), flush=True)
        for other in ("fp32", "bf16x3"):
            print("   %s vs f16x2: max|dE| %.3e, same iters %.4f" % (other, np.abs(out[other][0] - out["f16x2"][0]).max(),
                  (out[other][1]["iters"] == out["f16x2"][1]["iters"]).mean()))
for cfg, regs in (("c3", (0.1, 1.0)), ("c2", (0.1,)), ("c4", (0.1,))):
    if QUICK and cfg != "c3":
        continue
    P, M = make_problem(**CONFIGS[cfg])
    for reg in regs:
        for prec in ("bf16x3", "f16x2"):
            plan = engine.DevicePlan(P, M)
            plan.enable_timing(True)
            n = 30 if cfg != "c4" else 4
            for _ in range(n):
                plan.run(reg, precision=prec)
            plan.sync()
            t = time.perf_counter()
            for _ in range(n):
                plan.run(reg, precision=prec)
            plan.sync()
            dt = (time.perf_counter() - t) / n
            m, tr = plan.kernel_times_ms(n)
            it = plan.fetch()[1]["iters"]
            print("%s reg %g %-7s step %.4f ms  kernel %.4f ms  track %.4f  mean iters %.2f -> %.1f ns per 16-pair update per SIMD"
                  % (cfg, reg, prec, dt * 1e3, m.mean(), tr.mean(), it.mean(), m.mean() * 1e6 / (it.sum() / 16 / 1024)), flush=True)
            plan.close()
