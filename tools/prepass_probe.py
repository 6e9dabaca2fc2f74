#!/usr/bin/env python3
"""Device pre-pass at cohort scale (GPU): per-type medians of a C x D embedding resident in HBM, and the (sample, type)
histogram, timed per call (wall clock around the C-ABI call, codes uploaded inside it) and checked against numpy.
  python tools/prepass_probe.py [C] [D] [K] [dtype] [reps]        default: 1 800 000 x 30, 50 types, float32 (BASELINE c3)
Under `rocprofv3 --kernel-trace --stats` the same command gives the per-kernel times quoted in DESIGN.md / bench.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1_800_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 30
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dtype = np.dtype(sys.argv[4] if len(sys.argv) > 4 else "float32")
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
N = max(1, C // 3000)
rng = np.random.default_rng(2)
w = rng.dirichlet(0.5 * np.ones(K), size=N)
per = -(-C // N)
sc = (np.arange(C) // per).astype(np.int32)
cc = np.concatenate([rng.choice(K, size=min(per, C - n * per), p=w[n]) for n in range(N)]).astype(np.int32)
mu = rng.standard_normal((K, D))
X = (mu[cc] + 0.3 * rng.standard_normal((C, D))).astype(dtype)
print("C=%d D=%d K=%d N=%d dtype=%s  (%.1f MB embedding)" % (C, D, K, N, dtype, X.nbytes / 1e6))

up = engine.EmbeddingUpload(X)
got = up.medians(cc, K)
t = []
for _ in range(reps):
    t0 = time.perf_counter(); got = up.medians(cc, K); t.append(time.perf_counter() - t0)
print("medians (resident embedding, codes H2D + result D2H inside): best %.3f ms, median %.3f ms" % (min(t) * 1e3, sorted(t)[len(t) // 2] * 1e3))
t = []
for _ in range(reps):
    t0 = time.perf_counter(); P, first = engine.proportions_and_first_rows(cc, sc, N, K); t.append(time.perf_counter() - t0)
print("proportions + first rows (codes H2D inside): best %.3f ms, median %.3f ms" % (min(t) * 1e3, sorted(t)[len(t) // 2] * 1e3))
up.close()
t0 = time.perf_counter()
order = np.argsort(cc, kind="stable")
bounds = np.searchsorted(cc[order], np.arange(K + 1))
want = np.stack([np.median(X[order[bounds[k]:bounds[k + 1]]], axis=0).astype(np.float64) if bounds[k + 1] > bounds[k] else np.full(D, np.nan)
                 for k in range(K)])
print("numpy medians on the host: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
print("medians bit-exact vs numpy:", bool(np.array_equal(got, want, equal_nan=True)))
counts = np.bincount(sc.astype(np.int64) * K + cc, minlength=N * K).reshape(N, K).astype(np.float64)
prior = counts.sum(0) / (C - 1) * 0.2
wantP = np.stack([(counts[n] + prior) / (sum(counts[n]) + sum(prior)) for n in range(N)])
print("proportions bit-exact vs numpy:", bool(np.array_equal(P, wantP)))
