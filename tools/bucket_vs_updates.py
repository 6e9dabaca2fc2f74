"""How well does the order key (L1 distance bucket, 4 per octave) predict the update count of a pair?  c3, reg 0.1."""
import sys
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
for cfg in ("c3", "c2"):
    P, M = make_problem(**CONFIGS[cfg])
    E, info = engine.sinkhorn_grid(P, M, 0.1, return_info=True)
    it = info["iters"]
    l1 = np.abs(P[:, None, :] - P[None, :, :]).sum(-1)
    b = np.where(l1 > 0, np.clip(np.floor(4 * (1 - np.log2(np.maximum(l1, 1e-300)))), 0, 46), 47).astype(int)
    print(cfg, "pairs", it.size, "mean updates %.1f" % it.mean())
    for k in range(48):
        m = b == k
        if m.any():
            v = it[m]
            print("  bucket %2d: %7d pairs  updates min %3d  median %3d  p90 %3d  max %3d   share >=61: %.3f" % (k, m.sum(), v.min(), np.median(v), np.percentile(v, 90), v.max(), (v >= 61).mean()))
