"""Serial update latency of the two paths at small K: a grid of 16 identical patients whose pairs all run to the cap
(1000 updates) is ONE tile (PILOT_OT_DEBUG=512: no one-wave path) or, with N = 1, one wave on the diagonal path; the kernel time
divided by 1000 is the latency of one update.  Then the 600-patient grid with the one-wave path on / off and in natural order."""
import os, sys, subprocess
__import__("sys").path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))   # tools/switches.py
import switches
sys.path.insert(0, ".")
import numpy as np


def run(K, N, debug, full=False):
    switches.set("PILOT_OT_DEBUG", debug)
    from pilot_amd import engine
    from pilot_amd.synthetic import make_problem
    P, M = make_problem(600, K, 8, seed=K, cells_per_patient=200)
    if not full:
        from oracle import oracle as O
        _, io = O.sinkhorn_grid(P, M, 0.1, n_threads=8, return_info=True, row_end=60)
        d = np.argwhere(io["iters"][np.arange(60), np.arange(60)] >= 1000).ravel()
        P = np.repeat(P[d[:1]], N, axis=0)
    plan = engine.DevicePlan(P, M); plan.enable_timing(True)
    for _ in range(10): plan.run(0.1)
    plan.sync()
    a, b = plan.kernel_times_ms(10)
    _, info = plan.fetch()
    print("K=%d N=%d debug=%d: main %.4f ms track %.4f ms | updates mean %.1f max %d" % (K, P.shape[0], debug, a.mean(), b.mean(), info["iters"].mean(), info["iters"].max()), flush=True)
    plan.close()


if len(sys.argv) > 1:
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), len(sys.argv) > 4)
else:
    for K in (2, 4, 8):
        for args in ((K, 16, 512), (K, 1, 0), (K, 600, 0, 1), (K, 600, 512, 1), (K, 600, 2, 1)):
            subprocess.run([sys.executable, __file__] + [str(a) for a in args])
