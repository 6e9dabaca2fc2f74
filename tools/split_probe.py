"""GPU probe: the bf16-split configuration against the oracle and against the exact-f32 kernel (c1..c4 shapes)."""
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
        for prec in ("fp32", "bf16x3"):
            E, inf = engine.sinkhorn_grid(P, M, reg, precision=prec, row_step=step, return_info=True)
            out[prec] = (E, inf)
            print("%s reg %-4g %-7s max|E-oracle| %.3e  iters<=oracle %s  mean iters %.2f (oracle %.2f)  flags nan %d"
                  % (cfg, reg, prec, np.abs(E - Eo).max(), bool((inf["iters"] <= io["iters"]).all()), inf["iters"].mean(),
                     io["iters"].mean(), int(((inf["flags"] & 2) > 0).sum())))
        print("   fp32 vs bf16x3: max|dE| %.3e, same iters %.4f" % (np.abs(out["fp32"][0] - out["bf16x3"][0]).max(),
              (out["fp32"][1]["iters"] == out["bf16x3"][1]["iters"]).mean()))
P, M = make_problem(**CONFIGS["c3"])
REG = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
for prec in (("bf16x3", "fp32") if len(sys.argv) > 3 else (("bf16x3",) if QUICK else ("fp32", "bf16x3"))):
    plan = engine.DevicePlan(P, M)
    plan.enable_timing(True)
    for _ in range(30):
        plan.run(REG, precision=prec)
    plan.sync()
    t = time.perf_counter()
    for _ in range(20):
        plan.run(REG, precision=prec)
    plan.sync()
    dt = (time.perf_counter() - t) / 20
    m, tr = plan.kernel_times_ms(20)
    it = plan.fetch()[1]["iters"]
    print("c3 reg %g %-7s step %.4f ms  kernel %.4f ms  track %.4f  mean iters %.2f -> %.1f ns per 16-pair iteration per SIMD"
          % (REG, prec, dt * 1e3, m.mean(), tr.mean(), it.mean(), m.mean() * 1e6 / (it.sum() / 16 / 1024)))
    plan.close()
