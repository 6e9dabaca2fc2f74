#!/usr/bin/env python3
"""Tile rate of the cell-level W2 kernel on uniform work (every pair exactly `cap` updates): dot-product TFLOP/s."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
cap = 11
for (N, cells, D) in [(64, 2000, 30), (48, 5000, 30), (48, 5000, 50), (64, 2000, 16)]:
    rng = np.random.default_rng(0)
    offs = np.arange(N + 1, dtype=np.int64) * cells
    X = (rng.standard_normal((N, 1, D)) * 0.5 + rng.standard_normal((N, cells, D))).reshape(-1, D).astype(np.float32)
    scale = 2.0 * float(((X - X.mean(0)) ** 2).sum(1).mean())
    engine.cell_w2_grid(X[:offs[2]], offs[:3], scale, 0.1, num_iter_max=cap)
    t = time.perf_counter(); W, info = engine.cell_w2_grid(X, offs, scale, 0.1, num_iter_max=cap, return_info=True); dt = time.perf_counter() - t
    its = int(info["iters"].sum())
    passes = 2 * its + N * N                      # + the value pass
    Dp = 16 if D <= 16 else (32 if D <= 32 else 64)
    print("N=%d cells=%d D=%d: %.3f s  %d updates  %.1f TF/s of dot products (padded D=%d: %.1f)" % (
        N, cells, D, dt, its, passes * 2.0 * cells * cells * D / dt / 1e12, Dp, passes * 2.0 * cells * cells * Dp / dt / 1e12), flush=True)
