"""Host-side steps of tl.wasserstein_distance against the reference-executed golden fixtures and
against the oracle's literal restatements.  CPU only (nothing here touches the device)."""
import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN_CASES, golden_adata, load_golden
from oracle import oracle as O
from pilot_amd import tl


def _annot(g):
    ad, cell_col = golden_adata(g)
    if str(g["data_type"]) == "scRNA":
        return tl.extract_data_anno_scRNA_from_h5ad(ad, emb_matrix="X_pca", clusters_col=cell_col,
                                                    sample_col="sampleID", status="status")
    return tl.extract_data_anno_pathomics_from_h5ad(ad, var_names=list(ad.var_names), clusters_col=cell_col,
                                                    sample_col="sampleID", status="status")


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_extract_frames_have_reference_columns(name):
    g = load_golden(name)
    data, annot = _annot(g)
    assert list(annot.columns) == ["cell_type", "sampleID", "status"]
    assert list(annot.index) == list(range(len(annot)))
    if str(g["data_type"]) == "scRNA":
        assert list(data.columns) == ["PCA_%d" % i for i in range(1, g["emb"].shape[1] + 1)]
    assert tl.path_to_results == "Results_PILOT/plots"


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("categorical", [False, True])
def test_cluster_representations_bit_exact_vs_reference(name, categorical):
    g = load_golden(name)
    data, annot = _annot(g)
    if categorical:
        for c in annot.columns:
            annot[c] = annot[c].astype("category")
    rep = tl.Cluster_Representations(annot, regulizer=0.2, normalization=True)
    assert isinstance(rep, dict)
    assert [str(k) for k in rep.keys()] == list(g["samples"])      # first-appearance order
    got = np.stack(list(rep.values()))
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, g["proportions"])           # bit-exact with the reference's output
    np.testing.assert_allclose(got.sum(1), 1.0, atol=1e-15)
    assert (got > 0).all()


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_cluster_representations_matches_oracle_restatement(name):
    g = load_golden(name)
    _, annot = _annot(g)
    for regulizer in (0.2, 1.0):
        rep = tl.Cluster_Representations(annot, regulizer=regulizer)
        ora, cells = O.cluster_representations(annot["cell_type"], annot["sampleID"], regulizer=regulizer)
        assert list(rep.keys()) == list(ora.keys())
        for k in rep:
            np.testing.assert_array_equal(rep[k], ora[k])
    raw = tl.Cluster_Representations(annot, normalization=False)
    ora, _ = O.cluster_representations(annot["cell_type"], annot["sampleID"], normalization=False)
    for k in raw:
        np.testing.assert_array_equal(raw[k], ora[k])


def test_cluster_representations_prior_uses_c_minus_one():
    annot = pd.DataFrame({"cell_type": list("aabbbc"), "sampleID": list("xxyyyz"), "status": list("ppqqqr")})
    rep = tl.Cluster_Representations(annot, regulizer=0.5)
    prior = 0.5 * np.array([2, 3, 1]) / 5.0                        # n_k / (C - 1), C = 6
    np.testing.assert_array_equal(rep["x"], (np.array([2., 0, 0]) + prior) / (2 + sum(prior)))
    np.testing.assert_array_equal(rep["z"], (np.array([0., 0, 1]) + prior) / (1 + sum(prior)))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_centroid_medians_reproduce_reference_cost_through_scipy(name):
    """Host median step + scipy pdist (the call the reference makes) == the reference's cost matrix.
    (The device pdist kernel is checked against the same numbers in the gpu tests.)"""
    import scipy.spatial.distance as ssd
    g = load_golden(name)
    data, annot = _annot(g)
    codes, cells = tl._first_appearance_codes(annot["cell_type"])
    assert [str(c) for c in cells] == list(g["cells"])
    cent = tl._centroid_medians(data, codes, len(cells))
    dis = ssd.squareform(ssd.pdist(cent, metric="cosine"))
    np.testing.assert_allclose(dis, g["cost"], rtol=0, atol=1e-15)
    ora_dis, ora_cent, _ = O.cost_matrix(data, annot["cell_type"])
    np.testing.assert_array_equal(cent, ora_cent)                  # medians in the frame's dtype, like pandas
    np.testing.assert_array_equal(ora_dis, g["cost"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_return_real_labels(name):
    g = load_golden(name)
    _, annot = _annot(g)
    labels = tl.return_real_labels(annot)
    assert isinstance(labels, list) and [str(x) for x in labels] == list(g["real_labels"])
    assert labels == O.return_real_labels(annot["sampleID"], annot["status"])


def test_set_path_for_results_creates_dir(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("PILOT_AMD_NO_RESULTS_DIR", "0")
    assert tl.set_path_for_results() == "Results_PILOT/plots"
    assert (tmp_path / "Results_PILOT" / "plots").is_dir()


def test_return_sil_ari_is_rejected_loudly_before_any_device_work():
    # argument handling only: the engine is never reached because extraction fails first on a bad key
    g = load_golden("c1_20x10x10")
    ad, _ = golden_adata(g)
    with pytest.raises(KeyError):
        tl.wasserstein_distance(ad, emb_matrix="X_PCA")            # reference default key, absent here
