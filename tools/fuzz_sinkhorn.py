"""Random (N, K, reg, tau, cap, masses) against the oracle: fp64 must agree in value and update count; f32 in value except where
the ABSORB_LAST knife edge differs.  Usage: python tools/fuzz_sinkhorn.py [n_cases] [seed]"""
import sys
sys.path.insert(0, ".")
import os
import numpy as np
from scipy.spatial.distance import pdist, squareform
from oracle import oracle as O
from pilot_amd import _lib, engine
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    N = int(rng.integers(1, 400 if os.environ.get("FUZZ_BIG") else 70)); K = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 13, 16, 17, 20, 31, 32, 33, 48, 50, 64, 65, 80, 100, 128, 129, 130, 150, 192, 200, 255, 256]))
    reg = float(rng.choice([1.0, 0.3, 0.1, 0.05, 0.02, 0.01]))
    alpha = float(rng.choice([0.2, 1.0, 5.0]))
    P = rng.dirichlet(alpha * np.ones(K), size=N)
    if rng.random() < 0.3: P[P < 0.02] = 0.0; P[P.sum(1) == 0, 0] = 1.0; P /= P.sum(1, keepdims=True)
    if rng.random() < 0.2: P *= rng.uniform(0.5, 2.0, size=(N, 1))
    if K > 1:
        M = squareform(pdist(rng.standard_normal((K, 6)), rng.choice(["cosine", "euclidean", "cityblock"]))); M /= M.max()
    else:
        M = np.zeros((1, 1))
    sym = rng.random() < 0.85
    if not sym and K > 1: M = M * rng.uniform(0.7, 1.0, size=M.shape); M /= M.max()
    kw_o, kw_g = {}, {}
    if rng.random() < 0.2: kw_o["tau"] = kw_g["tau"] = float(rng.choice([6.5, 47.3]))     # (tau == K is a knife edge: the residual after a reset is K)
    if rng.random() < 0.2: kw_o["numItermax"] = kw_g["num_iter_max"] = int(rng.choice([1, 7, 40, 200]))
    Eo, io = O.sinkhorn_grid(P, M, reg, return_info=True, n_threads=8, **kw_o)
    msgs = []
    E64, i64 = engine.sinkhorn_grid(P, M, reg, precision="fp64", return_info=True, **kw_g)
    fin = np.isfinite(Eo)
    if not (np.isfinite(E64) == fin).all() or np.abs(E64 - Eo)[fin].max(initial=0) > 1e-11 * max(1.0, np.abs(Eo[fin]).max(initial=0)) or not (i64["iters"] == io["iters"]).all():
        msgs.append("fp64: max|d| %.2e iters equal %d/%d" % (np.abs(E64 - Eo)[fin].max(initial=0), (i64["iters"] == io["iters"]).sum(), Eo.size))
    for prec in (("auto", "fp32") if 1.0 / reg <= 60.0 else ("auto",)):      # explicit fp32 only inside its range (header: PREC_AUTO)
        Eg, ig = engine.sinkhorn_grid(P, M, reg, precision=prec, return_info=True, **kw_g)
        edge = ((io["flags"] & O.FLAG_ABSORB_ON_LAST) > 0) != ((ig["flags"] & _lib.FLAG_ABSORB_LAST) > 0)
        ok = fin & ~edge
        d = np.abs(Eg - Eo)[ok].max(initial=0)
        if not np.isfinite(Eg[fin]).all() or d > 1e-5 * max(1.0, np.abs(Eo[ok]).max(initial=0)) or edge.mean() > 0.1:
            msgs.append("%s: max|d| %.2e edge %.3f nan %d" % (prec, d, edge.mean(), np.isnan(Eg).sum()))
    tag = "N=%d K=%d reg=%g alpha=%g sym=%s %s" % (N, K, reg, alpha, sym, kw_g)
    if msgs:
        bad += 1; print("FAIL", tag, "|", "; ".join(msgs), flush=True)
        d64 = np.abs(E64 - Eo); d64[~np.isfinite(d64)] = np.inf
        for (i, j) in np.argwhere(d64 > 1e-9)[:3]:
            print("     fp64 (%d,%d): oracle %.6g it %d fl %d | gpu %.6g it %d fl %d | zeros a %d b %d  sum a %.4g b %.4g" % (
                i, j, Eo[i, j], io["iters"][i, j], io["flags"][i, j], E64[i, j], i64["iters"][i, j], i64["flags"][i, j],
                (P[i] == 0).sum(), (P[j] == 0).sum(), P[i].sum(), P[j].sum()), flush=True)
    else: print("ok  ", tag, flush=True)
print("%d of %d cases failed" % (bad, n_cases))
