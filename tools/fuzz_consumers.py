"""Random matrices for the f-4 consumer kernels against scipy / scikit-learn.  Usage: python tools/fuzz_consumers.py [n] [seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from scipy.spatial.distance import cdist
from sklearn.metrics import silhouette_samples, silhouette_score
from sklearn.metrics.pairwise import cosine_distances
from sklearn.neighbors import NearestNeighbors
from pilot_amd import engine
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    N = int(rng.choice([2, 3, 5, 16, 17, 63, 64, 65, 100, 257, 600]))
    X = rng.random((N, int(rng.integers(1, 6))))
    E = cdist(X, X) * float(rng.choice([1e-3, 1.0, 50.0]))
    if rng.random() < 0.3: E = np.round(E, 2)                       # ties
    if not (E.max() > 0) or (np.abs(E).sum(1) == 0).any():           # the all-zero matrix / zero rows: 0/0 in the reference too
        print("ok   (degenerate, skipped)"); continue
    msgs = []
    for metric in ("euclidean", "cosine"):
        for norm in (False, True):
            En = E / E.max() if (norm and E.max() > 0) else E
            want = cdist(En, En) if metric == "euclidean" else cosine_distances(En)
            got = engine.row_distances(E, metric=metric, normalize_by_max=norm)
            if not np.allclose(got, want, rtol=0, atol=1e-10 * max(1.0, np.abs(want).max())) and not (metric == "cosine" and not np.isfinite(want).all()):
                msgs.append("row_distances %s norm=%s: %.3e" % (metric, norm, np.abs(got - want).max()))
    D = engine.row_distances(E, metric="euclidean", normalize_by_max=True) if E.max() > 0 else E
    n_lab = int(rng.integers(2, max(3, min(N, 8))))
    labels = rng.integers(0, n_lab, N)
    if len(np.unique(labels)) >= 2 and len(np.unique(labels)) < N:
        want_s = silhouette_samples(D, labels, metric="precomputed")
        got, got_s = engine.silhouette_precomputed(D, labels, return_samples=True)
        if abs(got - silhouette_score(D, labels, metric="precomputed")) > 1e-10 or np.abs(got_s - want_s).max() > 1e-10:
            msgs.append("silhouette: %.3e" % np.abs(got_s - want_s).max())
    k = int(rng.choice([1, 2, 5, 64, N, N + 3])); eps = float(rng.choice([0.1, 1.0, 7.0]))
    Kg = engine.knn_gaussian_kernel(D, k=k, epsilon=eps)
    kk = min(k, N)
    # sklearn's neighbour order under exact ties is implementation defined: compare the kernel VALUES row by row
    want_rows = np.sort(np.exp(-np.sort(D, axis=1)[:, :kk] ** 2 / (4.0 * eps)), axis=1)
    got_rows = np.sort(np.where(Kg > 0, Kg, np.nan), axis=1)[:, :kk]
    nz = (Kg > 0).sum(1)
    if not (nz == kk).all() or np.nanmax(np.abs(got_rows - want_rows)) > 1e-12:
        msgs.append("knn kernel: nonzeros %s..%s (want %d), max|d| %.3e" % (nz.min(), nz.max(), kk, np.nanmax(np.abs(got_rows - want_rows))))
    tag = "N=%d k=%d eps=%g labels=%d" % (N, k, eps, n_lab)
    if msgs: bad += 1; print("FAIL", tag, "|", "; ".join(msgs), flush=True)
    else: print("ok  ", tag, flush=True)
print("%d of %d cases failed" % (bad, n_cases))
