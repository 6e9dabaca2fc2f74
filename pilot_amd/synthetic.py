"""Seeded synthetic inputs of the shapes BASELINE.json names.

Two generators:

* :func:`make_cells` -- a full cell-level cohort (embedding C x D float32 + the three
  ``obs`` columns), the input of ``tl.wasserstein_distance``.
* :func:`make_problem` -- the (N x K proportions, K x K cost) pair grid problem drawn from
  the same distribution without materialising cells; what ``bench.py`` and the
  kernel-level parity tests feed to the engine.

Configs (N patients x K cell types x D PCA dims, seed): c1 20x10x10 (0), c2 100x30x30 (1),
c3 600x50x30 (2), c4 2000x100x50 (3).
"""
from __future__ import annotations

import numpy as np

CONFIGS = {
    "c1": dict(n_patients=20, n_types=10, n_dims=10, seed=0, cells_per_patient=200),
    "c2": dict(n_patients=100, n_types=30, n_dims=30, seed=1, cells_per_patient=3000),
    "c3": dict(n_patients=600, n_types=50, n_dims=30, seed=2, cells_per_patient=3000),
    "c4": dict(n_patients=2000, n_types=100, n_dims=50, seed=3, cells_per_patient=3000),
}


class Cohort:
    """Duck-typed stand-in for an AnnData: ``.obsm``, ``.obs``, ``.uns``, ``.X``, ``.var_names``
    (anndata itself is not installed on the build or GPU boxes)."""

    def __init__(self, emb, obs, emb_key="X_pca"):
        self.obsm = {emb_key: emb}
        self.obs = obs
        self.uns = {}
        self.X = emb
        self.var_names = ["feat_%d" % i for i in range(emb.shape[1])]

    def __getitem__(self, key):
        # adata[:, var_names] as used by the pathomics branch (Trajectory.py:292)
        _, names = key
        idx = [self.var_names.index(v) for v in names]
        sub = Cohort(self.X[:, idx], self.obs)
        sub.var_names = list(names)
        return sub


def make_cells(n_patients, n_types, n_dims, seed, cells_per_patient=3000, emb_key="X_pca"):
    """Cell-level cohort: centres mu_k ~ N(0, I); patient mixing weights ~ Dirichlet(0.5);
    cells_per_patient cells ~ Multinomial; cell = mu_k + 0.3 N(0, I) (float32); status by parity;
    obs order patient-major."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((n_types, n_dims))
    types, samples, status, chunks = [], [], [], []
    for p in range(n_patients):
        w = rng.dirichlet(0.5 * np.ones(n_types))
        counts = rng.multinomial(cells_per_patient, w)
        k = np.repeat(np.arange(n_types), counts)
        chunks.append((mu[k] + 0.3 * rng.standard_normal((k.size, n_dims))).astype(np.float32))
        types.append(k)
        samples.append(np.full(k.size, p))
        status.append(np.full(k.size, p % 2))
    k = np.concatenate(types)
    s = np.concatenate(samples)
    st = np.concatenate(status)
    obs = pd.DataFrame({
        "cell_types": np.array(["ct%03d" % i for i in range(n_types)], dtype=object)[k],
        "sampleID": np.array(["P%04d" % i for i in range(n_patients)], dtype=object)[s],
        "status": np.array(["ctrl", "case"], dtype=object)[st],
    })
    return Cohort(np.concatenate(chunks), obs, emb_key=emb_key)


def make_problem(n_patients, n_types, n_dims, seed, cells_per_patient=3000, regulizer=0.2,
                 **_unused):
    """Pair-grid problem without cells: returns (P float64 N x K, M float64 K x K in [0,1]).

    Counts ~ Multinomial(cells_per_patient, Dirichlet(0.5)); proportions with PILOT's smoothing
    (Trajectory.py:405-430: prior_k = regulizer * n_k / (C - 1)); centroids = centres plus the
    sampling noise of a median of ~n_k cells; M = cosine pdist / max (Trajectory.py:468, :101).
    """
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((n_types, n_dims))
    counts = np.stack([rng.multinomial(cells_per_patient, rng.dirichlet(0.5 * np.ones(n_types)))
                       for _ in range(n_patients)]).astype(np.float64)
    n_k = counts.sum(axis=0)
    prior = regulizer * n_k / (counts.sum() - 1.0)
    P = (counts + prior) / (counts.sum(axis=1, keepdims=True) + prior.sum())
    cent = mu + 0.3 * 1.2533 / np.sqrt(np.maximum(n_k, 1.0))[:, None] * rng.standard_normal(mu.shape)
    nrm = np.linalg.norm(cent, axis=1)
    M = 1.0 - (cent @ cent.T) / np.outer(nrm, nrm)
    M = 0.5 * (M + M.T)
    np.fill_diagonal(M, 0.0)
    M = np.clip(M, 0.0, 2.0)
    mx = M.max()
    return np.ascontiguousarray(P), np.ascontiguousarray(M / mx if mx > 0 else M)       # (K = 1: the single zero stays)


def make_cell_clouds(n_patients, cells_per_patient, n_dims, seed, n_types=20):
    """Cell-level cohort for the cell-level W2 extension (BASELINE config 5: 200 patients x 5000 cells x 30 dims):
    the same mixture model as :func:`make_cells`, returned as what ``engine.cell_w2_grid`` takes --
    (X float32 C x D with every patient's cells contiguous, offsets int64 N + 1, scale) where scale is twice the mean
    squared distance of the cells to the global centroid (the default of ``tl.cell_level_wasserstein``)."""
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((n_types, n_dims))
    X = np.empty((n_patients * cells_per_patient, n_dims), dtype=np.float32)
    for p in range(n_patients):
        w = rng.dirichlet(0.5 * np.ones(n_types))
        k = rng.choice(n_types, size=cells_per_patient, p=w)
        X[p * cells_per_patient:(p + 1) * cells_per_patient] = mu[k] + 0.3 * rng.standard_normal((cells_per_patient, n_dims))
    offsets = np.arange(n_patients + 1, dtype=np.int64) * cells_per_patient
    m = X.mean(axis=0, dtype=np.float64)
    scale = 2.0 * float(((X.astype(np.float64) - m) ** 2).sum(axis=1).mean())
    return X, offsets, scale
