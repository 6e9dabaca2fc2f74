"""Random inputs for the host pre-pass kernels (proportions, per-type medians, pdist) against pandas / scipy / the oracle.
Usage: python tools/fuzz_prepass.py [n_cases] [seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import pandas as pd
from scipy.spatial.distance import pdist, squareform
from oracle import oracle as O
from pilot_amd import engine, _lib
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
METRICS = list(_lib.METRICS)
for case in range(n_cases):
    C = int(rng.choice([2, 3, 5, 63, 64, 65, 1000, 4096, 4097, 30000, 250000])); D = int(rng.choice([1, 2, 3, 30, 64, 65, 80, 150, 400])); K = int(rng.choice([1, 2, 3, 7, 50, 130]))
    if C * D > 3e7: D = 30
    N = int(rng.choice([1, 2, 7, 39, 700, 3000]))
    dtype = rng.choice([np.float32, np.float64])
    X = (rng.standard_normal((C, D)) * rng.choice([1e-3, 1.0, 1e4])).astype(dtype)
    if rng.random() < 0.3: X = np.round(X, 1)                      # many ties
    if rng.random() < 0.2: X[rng.integers(0, C, max(1, C // 50)), rng.integers(0, D)] = -0.0
    cc = rng.integers(0, K, C).astype(np.int32)
    if rng.random() < 0.3 and K > 1: cc[cc == K - 1] = 0              # an empty type
    sc = rng.integers(0, N, C).astype(np.int32)
    if rng.random() < 0.5: sc = np.sort(sc)                          # sample-major, like a stored cohort (the LDS window of the count kernel)
    if rng.random() < 0.3: cc[rng.random(C) < 0.02] = -1             # missing labels
    if rng.random() < 0.3: sc[rng.random(C) < 0.02] = -1
    ok_c, ok_s = cc >= 0, sc >= 0
    msgs = []
    # medians vs pandas (in the data's dtype, like the reference)
    got = engine.centroid_medians(X, cc, K)
    df = pd.DataFrame(X)
    want = np.stack([df[cc == k].median(axis=0).to_numpy(dtype=np.float64) if (cc == k).any() else np.full(D, np.nan) for k in range(K)])
    if not np.array_equal(got, want, equal_nan=True): msgs.append("medians differ: max|d| %.3e" % np.nanmax(np.abs(got - want)))
    # the whole pre-pass from one upload of the code columns (pilot_ot_prepass_dev): the bits of the separate calls
    up = engine.EmbeddingUpload(X.copy())
    if up.thread is None:                                            # (small arrays take the host-array path: force the resident one)
        up.SMALL_BYTES = 0; up.__init__(X.copy())
    try:
        P1, f1, c1 = up.prepass(cc, sc, N, K, regulizer=0.2, n_total=C)
    finally:
        up.close()
    P2, f2 = engine.proportions_and_first_rows(cc, sc, N, K, regulizer=0.2, n_total=C)
    if not (np.array_equal(P1, P2, equal_nan=True) and np.array_equal(f1, f2) and np.array_equal(c1, got, equal_nan=True)): msgs.append("prepass_dev differs from the separate calls")
    wf = np.full(N, -1, dtype=np.int64); idx = np.flatnonzero(ok_s); wf[sc[idx][::-1]] = idx[::-1]
    if not np.array_equal(f2, wf): msgs.append("first rows differ")
    # proportions vs the formula of Trajectory.py:405-430
    reg = float(rng.choice([0.0, 0.2, 1.0]))
    for norm in (True, False):
        P = engine.proportions(cc, sc, N, K, regulizer=reg, normalization=norm, n_total=C)
        cnt = np.zeros((N, K)); np.add.at(cnt, (sc[ok_c & ok_s], cc[ok_c & ok_s]), 1.0)
        if norm:
            if C > 1:
                prior = np.array([reg * float(cnt[:, k].sum()) / (C - 1) for k in range(K)])
                ref = np.stack([np.array([(cnt[s, k] + prior[k]) for k in range(K)]) / (sum(cnt[s].tolist()) + sum(prior.tolist())) for s in range(N)]) if True else None
                ok = np.allclose(P, ref, rtol=1e-15, atol=0) if np.isfinite(ref).all() else True
                if not ok: msgs.append("proportions(norm) differ: %.3e" % np.abs(P - ref).max())
        else:
            if not np.array_equal(P, cnt): msgs.append("raw counts differ")
    # pdist
    cent = got[~np.isnan(got).any(1)]
    if len(cent) >= 2:
        metric = str(rng.choice(METRICS))
        if metric in ("dice", "jensenshannon", "jaccard", "yule", "russellrao", "sokalsneath", "rogerstanimoto", "sokalmichener", "kulczynski1"):
            # scipy's set-style dissimilarities are meant for non-negative rows (dice on signed values divides by a sum that
            # cancels: any summation order gives a different number): non-negative rows with actual zeros
            cent = np.abs(cent)
            cent[rng.random(cent.shape) < 0.3] = 0.0
        if metric == "jensenshannon" and rng.random() < 0.5:
            # proportional rows: the distance is 0 up to rounding, and scipy returns the root of whatever the rounding left (NaN below zero)
            cent[1::2] = cent[0::2][:len(cent[1::2])] * float(np.float32(rng.uniform(0.1, 3.0)))
        want = err_ref = None
        try:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = squareform(pdist(cent, metric))
        except Exception as e:                          # scipy refuses (mahalanobis with K <= D, a singular covariance)
            err_ref = e
        try:
            g = engine.pdist_square(cent, metric)
            if err_ref is not None:
                msgs.append("pdist %s: scipy raised %r, the engine did not" % (metric, err_ref))
            else:
                fin = np.isfinite(want)
                same_nan = (np.isnan(g) == np.isnan(want)).all() and (np.isinf(g) == np.isinf(want)).all()
                if not same_nan or np.abs(g - want)[fin].max(initial=0) > 1e-11 * max(1.0, np.abs(want[fin]).max(initial=0)):
                    msgs.append("pdist %s differs: %.3e" % (metric, np.abs(g - want)[fin].max(initial=0)))
                    bad_at = np.argwhere((np.isnan(g) != np.isnan(want)) | (np.isinf(g) != np.isinf(want)))
                    for (i, j) in bad_at[:3]:
                        msgs.append("[%d, %d]: engine %r scipy %r rows %r %r" % (i, j, g[i, j], want[i, j], cent[i].tolist()[:4], cent[j].tolist()[:4]))
        except Exception as e:
            if err_ref is None:
                msgs.append("pdist %s: the engine raised %r, scipy did not" % (metric, e))
    tag = "C=%d D=%d K=%d N=%d %s reg=%g" % (C, D, K, N, np.dtype(dtype).name, reg)
    if msgs: bad += 1; print("FAIL", tag, "|", "; ".join(msgs), flush=True)
    else: print("ok  ", tag, flush=True)
print("%d of %d cases failed" % (bad, n_cases))
