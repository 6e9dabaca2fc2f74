#!/usr/bin/env python3
"""Device time of the matrix consumers on a resident N x N matrix, N = 600 (c3) and 2000 (c4): the fused
chains with the matrix already in HBM (engine.DeviceMatrix), wall time per call incl. the small D2H of their results."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
for cfg in ("c3", "c4"):
    P, M = make_problem(**CONFIGS[cfg])
    N = P.shape[0]
    plan = engine.DevicePlan(P, M)
    plan.run(0.1); plan.sync()
    dm = plan.device_matrix()
    labels = np.arange(N) % 2
    for name, fn in (("silhouette_of_rows (cosine)", lambda: engine.silhouette_of_rows(dm, labels, metric="cosine", normalize_by_max=True)),
                     ("silhouette_of_rows (euclidean)", lambda: engine.silhouette_of_rows(dm, labels, metric="euclidean", normalize_by_max=True)),
                     ("diffusion_kernel_of_rows (k=64), kernel matrix to the host", lambda: engine.diffusion_kernel_of_rows(dm, k=64, epsilon=1.0, return_distances=False))):
        fn()
        t = time.perf_counter()
        for _ in range(5): fn()
        print("%s N=%4d  %-62s %.3f ms per call" % (cfg, N, name, (time.perf_counter() - t) / 5 * 1e3), flush=True)
    plan.close()
