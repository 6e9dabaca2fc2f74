"""Is the small-K Sinkhorn grid bound by throughput (time ~ N^2) or by a serial tail (time ~ const)?"""
import os, sys, subprocess
__import__("sys").path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))   # tools/switches.py
import switches
sys.path.insert(0, ".")
import numpy as np
def run(K, N, debug):
    switches.set("PILOT_OT_DEBUG", debug)
    from pilot_amd import engine
    from pilot_amd.synthetic import make_problem
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
    plan = engine.DevicePlan(P, M); plan.enable_timing(True)
    for _ in range(10): plan.run(0.1)
    plan.sync()
    a, b = plan.kernel_times_ms(10)
    _, info = plan.fetch()
    it = info["iters"]
    print("K=%d N=%4d debug=%d: main %.4f ms track %.4f | updates mean %.1f, capped %d, sum/16 = %.0f tile-updates" % (K, N, debug, a.mean(), b.mean(), it.mean(), (it >= 1000).sum(), it.sum() / 16), flush=True)
    plan.close()
if len(sys.argv) > 1:
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
else:
    for K in (2, 4):
        for N in (150, 300, 600, 1200):
            subprocess.run([sys.executable, __file__, str(K), str(N), "0"])
        for dbg in (16, 48, 64, 512):     # 1, 3, 4 resident workgroups per CU; no solo path
            subprocess.run([sys.executable, __file__, str(K), "600", str(dbg)])
