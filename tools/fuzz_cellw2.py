"""Random cell-level W2 cohorts against the fp64 oracle.  Usage: python tools/fuzz_cellw2.py [n_cases] [seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from pilot_amd import engine
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    Np = int(rng.integers(2, 6)); D = int(rng.choice([1, 2, 5, 16, 17, 30, 31, 32, 33, 50, 62, 63, 64]))
    sizes = rng.choice([1, 2, 15, 16, 17, 31, 33, 64, 65, 100, 257], size=Np)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    centres = rng.standard_normal((Np, D)) * float(rng.choice([0.0, 0.5, 3.0]))
    X = np.concatenate([centres[p] + rng.standard_normal((sizes[p], D)) for p in range(Np)]).astype(np.float32)
    if rng.random() < 0.2: X += 10.0                              # far from the origin: cancellation in |x|^2 + |y|^2 - 2 x.y
    mu = X.mean(0, dtype=np.float64)
    scale = 2.0 * max(float(((X - mu) ** 2).sum(1).mean()), 1e-3)
    reg = float(rng.choice([0.5, 0.1, 0.05]))
    kw = {}
    if rng.random() < 0.3: kw = dict(numItermax=int(rng.choice([1, 2, 11, 25])))
    Wo = O.cell_w2_grid(X, offs, scale, reg, **kw)
    Wg, info = engine.cell_w2_grid(X, offs, scale, reg, return_info=True, **({"num_iter_max": kw["numItermax"]} if kw else {}))
    d = np.abs(Wg - Wo).max()
    ok = np.isfinite(Wg).all() and d <= 1e-5 * max(1.0, np.abs(Wo).max())
    if not ok: bad += 1
    print("%s Np=%d D=%d sizes=%s reg=%g %s max|d| %.2e (max W %.3g)" % ("ok  " if ok else "FAIL", Np, D, list(sizes), reg, kw, d, np.abs(Wo).max()), flush=True)
print("%d of %d cases failed" % (bad, n_cases))
