"""Random exact-OT grids against the oracle (and LP bounds): usage python tools/fuzz_emd.py [n_cases] [seed]"""
import sys
sys.path.insert(0, ".")
import os
import numpy as np
from scipy.spatial.distance import pdist, squareform
from oracle import oracle as O
from pilot_amd import engine
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    N = int(rng.integers(1, 300 if os.environ.get("FUZZ_BIG") else 60)); K = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 50, 63, 64, 65, 90, 100, 128, 129, 160, 192, 193, 256]))
    if K > 128: N = min(N, 24)
    kind = rng.choice(["dirichlet", "lattice", "sparse"])
    if kind == "dirichlet": P = rng.dirichlet(float(rng.choice([0.1, 1.0, 10.0])) * np.ones(K), size=N)
    elif kind == "lattice": P = rng.multinomial(16, np.ones(K) / K, size=N) / 16.0
    else:
        P = rng.random((N, K)) ** 4; P[rng.random((N, K)) < 0.85] = 0.0; P[P.sum(1) == 0, rng.integers(0, K)] = 1.0; P /= P.sum(1, keepdims=True)
    if rng.random() < 0.2: P *= rng.uniform(0.3, 3.0, size=(N, 1))
    ckind = rng.choice(["cosine", "line", "random", "quantised", "nonsym"])
    if K == 1: M = np.zeros((1, 1))
    elif ckind == "cosine": M = squareform(pdist(rng.standard_normal((K, 5)), "cosine"))
    elif ckind == "line": M = np.minimum(np.abs(np.arange(K)[:, None] - np.arange(K)[None, :]).astype(float), 5.0)
    elif ckind == "random": M = rng.random((K, K)); M = M + M.T; np.fill_diagonal(M, 0)
    elif ckind == "quantised": M = np.round(rng.random((K, K)) * 4) / 4; M = np.maximum(M, M.T); np.fill_diagonal(M, 0 if rng.random() < 0.7 else 0.25)
    else: M = rng.random((K, K))
    rb = int(rng.integers(0, N)); re_ = int(rng.integers(rb + 1, N + 1)); rs = int(rng.integers(1, 6))
    if rng.random() < 0.5: rb, re_, rs = 0, N, 1
    mode = "auto" if (rb, re_, rs) == (0, N, 1) else str(rng.choice(["all", "upper"]))
    Eo = O.emd_grid(P, M, row_begin=rb, row_end=re_, row_step=rs, n_threads=8)
    try:
        Eg, info = engine.emd_grid(P, M, row_begin=rb, row_end=re_, row_step=rs, mode=mode, return_info=True)
    except Exception as e:
        bad += 1; print("FAIL N=%d K=%d %s/%s rows %d:%d:%d mode %s: %s" % (N, K, kind, ckind, rb, re_, rs, mode, e), flush=True); continue
    rows = np.arange(rb, re_, rs)
    want = Eo if mode != "upper" else np.where(np.arange(N)[None, :] >= rows[:, None], Eo, 0.0)
    scale = max(1.0, np.abs(Eo).max())
    d = np.abs(Eg - want).max()
    ok = d <= 1e-11 * scale and np.isfinite(Eg).all()
    if not ok: bad += 1
    print("%s N=%d K=%d %s/%s rows %d:%d:%d mode %s max|d| %.2e" % ("ok  " if ok else "FAIL", N, K, kind, ckind, rb, re_, rs, mode, d), flush=True)
print("%d of %d cases failed" % (bad, n_cases))
