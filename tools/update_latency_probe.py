#!/usr/bin/env python3
"""What a small cohort's Sinkhorn call waits for: the launch ends with its longest pair, a serial chain of updates on one wave.
Per-update latency of a lone wave (16 patients: one tile per wave, 1000 updates forced) and the update-count tail of c1 / c2."""
import sys, time, numpy as np
sys.path.insert(0,'.')
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
for name in ("c2","c1"):
    P,M=make_problem(**CONFIGS[name])
    plan=engine.DevicePlan(P,M); plan.enable_timing(True)
    for _ in range(5): plan.run(0.1)
    plan.sync()
    t=time.perf_counter()
    for _ in range(50): plan.run(0.1)
    plan.sync(); dt=(time.perf_counter()-t)/50
    m,tr=plan.kernel_times_ms(50)
    it=plan.fetch()[1]["iters"]
    print("%s: N=%d K=%d step %.4f ms main kernel %.4f ms track %.4f; updates mean %.1f p99 %d max %d -> %.3f us per update of the longest pair"%(name,P.shape[0],P.shape[1],dt*1e3,m.mean(),tr.mean(),it.mean(),np.percentile(it,99),it.max(),m.mean()*1e3/it.max()))
    plan.close()
# a single pair, many updates: latency of one wave alone
for K in (14,30,50,100):
    P,M=make_problem(16,K,8,seed=K,cells_per_patient=300)
    plan=engine.DevicePlan(P,M); plan.enable_timing(True)
    for reg in (0.1,):
        for _ in range(5): plan.run(reg, stop_thr=0.0, num_iter_max=1000, f32_floor_ulps=1e-30)
        plan.sync()
        m,tr=plan.kernel_times_ms(5)
        it=plan.fetch()[1]["iters"]
        print("K=%d 16 patients, 1000 updates forced: main kernel %.3f ms, updates max %d -> %.3f us per update (one tile per wave)"%(K,m.mean(),it.max(),m.mean()*1e3/max(1,it.max())))
    plan.close()
