import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from oracle import oracle as O
from pilot_amd import engine
rng = np.random.default_rng(3)
for K in (12, 50):
    P = rng.dirichlet(0.3 * np.ones(K), size=24)
    P[P < 2e-2] = 0.0
    P[0] = 0.0; P[0, 3] = 1.0
    P /= P.sum(1, keepdims=True)
    X = rng.standard_normal((K, 8))
    from scipy.spatial.distance import pdist, squareform
    M = squareform(pdist(X, "cosine")); M /= M.max()
    for reg in (1.0, 0.1, 0.02):
        Eo, io = O.sinkhorn_grid(P, M, reg, return_info=True, n_threads=8)
        for prec in ("auto", "fp32", "fp64"):
            Eg, ig = engine.sinkhorn_grid(P, M, reg, precision=prec, return_info=True)
            nan_o, nan_g = np.isnan(Eo), np.isnan(Eg)
            ok = ~nan_o & ~nan_g
            print("K=%d reg=%g %-5s zeros %.0f%%  nan oracle %d gpu %d  max|d| %.3e  iters equal %d/%d  flags(o) %s flags(g) %s" % (
                K, reg, prec, 100 * (P == 0).mean(), nan_o.sum(), nan_g.sum(), np.abs(Eg - Eo)[ok].max(), (ig["iters"] == io["iters"]).sum(), Eo.size,
                np.unique(io["flags"]), np.unique(ig["flags"])))
