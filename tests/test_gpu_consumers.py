"""Downstream consumers of the distance matrix on the device against scipy / scikit-learn, and the
Precomputed_distance injection contract (pilotpy/tools/Trajectory.py:1687-1727)."""
import numpy as np
import pytest
from scipy.spatial.distance import cdist
from sklearn.metrics import silhouette_samples, silhouette_score
from sklearn.metrics.pairwise import cosine_distances
from sklearn.neighbors import NearestNeighbors

from conftest import load_golden, golden_adata
from pilot_amd import engine, tl
from pilot_amd.synthetic import CONFIGS, make_cells, make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def emd_c2():
    P, M = make_problem(**CONFIGS["c2"])
    return engine.emd_grid(P, M)


@pytest.mark.parametrize("N", [1, 7, 100, 333])
def test_row_distances_match_scipy_and_sklearn(N):
    rng = np.random.default_rng(N)
    E = rng.random((N, N)); E = E + E.T; np.fill_diagonal(E, 0.0)
    D = engine.row_distances(E, metric="euclidean")
    np.testing.assert_allclose(D, cdist(E, E), rtol=0, atol=1e-12)
    Dn = engine.row_distances(E, metric="euclidean", normalize_by_max=True)
    if N > 1:
        np.testing.assert_allclose(Dn, cdist(E / E.max(), E / E.max()), rtol=0, atol=1e-12)
        Dc = engine.row_distances(E, metric="cosine")
        np.testing.assert_allclose(Dc, cosine_distances(E), rtol=0, atol=1e-12)
        assert (np.diag(Dc) == 0).all()
    with pytest.raises(NotImplementedError):
        engine.row_distances(E, metric="mahalanobis")


@pytest.mark.parametrize("metric", ["cosine", "euclidean"])
def test_sil_computing_matches_sklearn(emd_c2, metric):
    E = emd_c2 / emd_c2.max()
    rng = np.random.default_rng(0)
    for labels in (np.arange(100) % 2, rng.integers(0, 5, 100), np.array(["a", "b", "c", "d"])[rng.integers(0, 4, 100)],
                   np.r_[np.zeros(99, dtype=int), 1]):                      # a singleton cluster: s = 0 for it
        want = silhouette_score(E, labels, metric=metric)
        assert abs(tl.Sil_computing(E, labels, metric=metric) - want) <= 1e-12
        D = engine.row_distances(E, metric=metric)
        got, samples = engine.silhouette_precomputed(D, labels, return_samples=True)
        np.testing.assert_allclose(samples, silhouette_samples(E, labels, metric=metric), rtol=0, atol=1e-11)
    with pytest.raises(ValueError):
        engine.silhouette_precomputed(D, np.zeros(100))                     # a single label, as sklearn refuses


def test_silhouette_without_the_lds_row_gives_the_same_bits(emd_c2, switches):
    """Beyond ~12 700 samples a row no longer fits LDS and the kernel reads it from global memory in the same order (round 3
    refused such N): forced here on a small matrix, the per-sample scores must not move by a bit."""
    E = emd_c2 / emd_c2.max()
    labels = np.random.default_rng(3).integers(0, 4, 100)
    D = engine.row_distances(E, metric="cosine")
    _, staged = engine.silhouette_precomputed(D, labels, return_samples=True)
    switches.setenv("PILOT_OT_SIL_UNSTAGED", "1")
    _, unstaged = engine.silhouette_precomputed(D, labels, return_samples=True)
    np.testing.assert_array_equal(staged, unstaged)


def test_diffusion_kernel_matches_sklearn_neighbours(emd_c2):
    ad = type("A", (), {})()
    ad.uns = {"EMD": emd_c2}
    k, eps = 16, 0.5
    D, Kmat = tl.diffusion_kernel(ad, epsilon=eps, knn=k)
    X = emd_c2 / emd_c2.max()
    np.testing.assert_allclose(D, cdist(X, X), rtol=0, atol=1e-12)
    G = NearestNeighbors(n_neighbors=k, metric="euclidean").fit(X).kneighbors_graph(X, mode="distance").toarray()
    mask = G > 0
    mask[np.arange(len(X)), np.arange(len(X))] = True                       # the point itself (distance 0) is neighbour 1
    assert (mask.sum(1) == k).all()
    want = np.where(mask, np.exp(-cdist(X, X) ** 2 / (4 * eps)), 0.0)
    np.testing.assert_allclose(Kmat, want, rtol=0, atol=1e-12)
    assert ((Kmat > 0).sum(1) == k).all() and (np.diag(Kmat) == 1.0).all()
    # k >= N keeps the full Gaussian kernel
    Kfull = engine.knn_gaussian_kernel(D, k=1000, epsilon=eps)
    np.testing.assert_allclose(Kfull, np.exp(-D ** 2 / (4 * eps)), rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", ["c1_20x10x10", "pathomics_15x6x8"])
def test_precomputed_distance_fills_the_uns_contract(name, tmp_path, monkeypatch):
    """Precomputed_distance (Trajectory.py:1687-1727) is the documented hook for handing an externally computed matrix to
    stock pilotpy: after it, adata.uns must hold exactly what wasserstein_distance leaves there for the downstream readers
    (ploting.py:95, 175, 190, 310-313) -- data, annot, proportions, cost, EMD, real_labels -- with the same types."""
    monkeypatch.chdir(tmp_path)
    g = load_golden(name)
    pathomics = str(g["data_type"]) != "scRNA"
    ad, cell_col = golden_adata(g)
    kw = dict(clusters_col=cell_col, sample_col="sampleID", status="status")
    dt = "Pathomics" if pathomics else "scRNA"
    tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized="unreg", data_type=dt, **kw)
    ref = dict(ad.uns)
    ad2, _ = golden_adata(g)
    tl.Precomputed_distance(ad2, ref["EMD"], ref["cost"], ref["proportions"], emb_matrix="X_pca", data_type=dt, **kw)
    assert set(ad2.uns) == {"data", "annot", "proportions", "cost", "EMD", "real_labels"}
    assert np.abs(ad2.uns["EMD"] - g["emd_unreg"]).max() <= 1e-12
    assert [str(x) for x in ad2.uns["real_labels"]] == [str(x) for x in ref["real_labels"]] == list(g["real_labels"])
    assert ad2.uns["annot"].equals(ref["annot"]) and ad2.uns["data"].equals(ref["data"])
    assert list(ad2.uns["annot"].columns) == ["cell_type", "sampleID", "status"]
    assert ad2.uns["cost"] is ref["cost"] and ad2.uns["proportions"] is ref["proportions"]
    # a stock consumer's first lines run on it: EMD / EMD.max() (ploting.py:95) and the frame of ploting.py:310-313
    import pandas as pd
    E = ad2.uns["EMD"] / ad2.uns["EMD"].max()
    df = pd.DataFrame(ad2.uns["EMD"], columns=ad2.uns["proportions"].keys())
    df["sampleID"] = ad2.uns["proportions"].keys()
    df["status"] = list(ad2.uns["real_labels"])
    assert E.max() == 1.0 and df.shape == (len(ad2.uns["real_labels"]), len(ad2.uns["real_labels"]) + 2)


def test_consumers_take_the_matrix_where_the_pair_grid_left_it():
    """pair grid -> row distances -> silhouette / diffusion kernel without the N x N matrix ever visiting host memory: the
    plan's (and the multi-device plan's) result in HBM is handed to the fused chains as an engine.DeviceMatrix.  Same bits as
    the host-array route."""
    from pilot_amd import multi
    P, M = make_problem(**CONFIGS["c2"])
    labels = np.arange(100) % 3
    plan = engine.DevicePlan(P, M)
    plan.run(0.1)
    plan.sync()
    E = plan.fetch()[0]
    dm = plan.device_matrix()
    for metric in ("cosine", "euclidean"):
        host = engine.silhouette_of_rows(E, labels, metric=metric, normalize_by_max=True, return_samples=True)
        dev = engine.silhouette_of_rows(dm, labels, metric=metric, normalize_by_max=True, return_samples=True)
        assert host[0] == dev[0]
        np.testing.assert_array_equal(host[1], dev[1])
        assert abs(host[0] - silhouette_score(E / E.max(), labels, metric=metric)) <= 1e-12
    Dh, Kh = engine.diffusion_kernel_of_rows(E, k=9, epsilon=0.7)
    Dd, Kd = engine.diffusion_kernel_of_rows(dm, k=9, epsilon=0.7)
    np.testing.assert_array_equal(Dh, Dd)
    np.testing.assert_array_equal(Kh, Kd)
    plan.close()
    mp = multi.MultiPlan(P, M, devices=[0, 0, 0])
    mp.sinkhorn(0.1)
    mp.sync()
    assert engine.silhouette_of_rows(mp.device_matrix(), labels, metric="cosine", normalize_by_max=True) == \
        engine.silhouette_of_rows(E, labels, metric="cosine", normalize_by_max=True)
    mp.close()


def test_knn_kernel_keeps_exactly_k_entries_when_distances_tie():
    """sklearn's kneighbors returns exactly k neighbours; among rows tied at the k-th distance this kernel takes the smallest
    indices (a fixed choice where sklearn's partial sort leaves the choice open)."""
    N, k = 12, 5
    D = np.ones((N, N))
    np.fill_diagonal(D, 0.0)                  # every other point at distance 1: an 11-way tie for 4 places
    D[3, 7] = D[7, 3] = 0.5
    Kmat = engine.knn_gaussian_kernel(D, k=k, epsilon=1.0)
    assert ((Kmat > 0).sum(1) == k).all()
    want0 = np.zeros(N); want0[[0, 1, 2, 3, 4]] = np.exp(-D[0, [0, 1, 2, 3, 4]] ** 2 / 4.0)
    np.testing.assert_allclose(Kmat[0], want0, atol=1e-15)
    assert set(np.flatnonzero(Kmat[3])) == {3, 7, 0, 1, 2}            # itself, the closer one, then ties in index order
    assert set(np.flatnonzero(Kmat[11])) == {11, 0, 1, 2, 3}
