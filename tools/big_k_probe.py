"""Beyond 128 cell types (200 patients, reg 0.1, host arrays in -> out): Sinkhorn at K = 128 / 130 / 192 / 256 / 300 (one wave per tile, eight
waves per tile, POT-literal kernel) and the exact grid across the same steps; parity against the oracle on sampled rows."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine
from pilot_amd.synthetic import make_problem
for K in (128, 130, 192, 256, 300):
    N = 200
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=800)
    t = time.perf_counter(); E = engine.sinkhorn_grid(P, M, 0.1); ts = time.perf_counter() - t
    t = time.perf_counter(); E = engine.sinkhorn_grid(P, M, 0.1); ts = time.perf_counter() - t
    t = time.perf_counter(); X = engine.emd_grid(P, M); te = time.perf_counter() - t
    t = time.perf_counter(); X = engine.emd_grid(P, M); te = time.perf_counter() - t
    print("N=%d K=%d: sinkhorn %.1f ms (%.2e pairs/s)  exact %.1f ms (%.2e pairs/s)" % (N, K, ts * 1e3, N * N / ts, te * 1e3, N * N / te), flush=True)
