"""Host-side steps of tl.wasserstein_distance against the reference-executed golden fixtures and
against the oracle's literal restatements.  CPU only (nothing here touches the device)."""
import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN_CASES, GOLDEN_OPTION_CASES, GOLDEN_REAL, golden_adata, load_golden, load_golden_pack
from oracle import oracle as O
from pilot_amd import engine, tl


def _annot(g):
    ad, cell_col = golden_adata(g)
    if str(g["data_type"]) == "scRNA":
        return tl.extract_data_anno_scRNA_from_h5ad(ad, emb_matrix="X_pca", clusters_col=cell_col,
                                                    sample_col="sampleID", status="status")
    return tl.extract_data_anno_pathomics_from_h5ad(ad, var_names=list(ad.var_names), clusters_col=cell_col,
                                                    sample_col="sampleID", status="status")


@pytest.mark.parametrize("name", GOLDEN_CASES + [GOLDEN_REAL])
def test_extract_frames_have_reference_columns(name):
    g = load_golden(name)
    data, annot = _annot(g)
    assert list(annot.columns) == ["cell_type", "sampleID", "status"]
    assert list(annot.index) == list(range(len(annot)))
    if str(g["data_type"]) == "scRNA":
        assert list(data.columns) == ["PCA_%d" % i for i in range(1, g["emb"].shape[1] + 1)]
    assert tl.path_to_results == "Results_PILOT/plots"


@pytest.mark.parametrize("name", GOLDEN_CASES + [GOLDEN_REAL])
@pytest.mark.parametrize("categorical", [False, True])
def test_label_codes_follow_first_appearance_order(name, categorical):
    """The host's only job for Cluster_Representations / cost_matrix is factorising the label columns; the
    codes must enumerate samples and cell types in ``Series.unique()`` order (Trajectory.py:402,412)."""
    g = load_golden(name)
    _, annot = _annot(g)
    if categorical:
        for c in annot.columns:
            annot[c] = annot[c].astype("category")
    ccodes, cells = tl._first_appearance_codes(annot["cell_type"])
    scodes, samples = tl._first_appearance_codes(annot["sampleID"])
    assert [str(c) for c in cells] == list(g["cells"]) == [str(c) for c in annot["cell_type"].unique()]
    assert [str(x) for x in samples] == list(g["samples"])
    assert ccodes.min() >= 0 and ccodes.max() == len(cells) - 1
    np.testing.assert_array_equal(np.asarray(cells, dtype=object)[ccodes], annot["cell_type"].astype(object).to_numpy())


@pytest.mark.parametrize("name", GOLDEN_CASES + [GOLDEN_REAL])
def test_oracle_restatements_match_the_reference_fixture(name):
    """oracle.cluster_representations / oracle.cost_matrix (what the GPU kernels are checked against) reproduce
    the numbers the reference's own code produced, bit for bit."""
    g = load_golden(name)
    data, annot = _annot(g)
    ora, cells = O.cluster_representations(annot["cell_type"], annot["sampleID"], regulizer=0.2)
    assert [str(k) for k in ora.keys()] == list(g["samples"])
    np.testing.assert_array_equal(np.stack(list(ora.values())), g["proportions"])
    ora_dis, ora_cent, _ = O.cost_matrix(data, annot["cell_type"])
    np.testing.assert_array_equal(ora_dis, g["cost"])
    assert ora_cent.shape == (len(g["cells"]), data.shape[1])


def test_real_dataset_fixture_is_what_the_reference_test_reads():
    """tests/golden/kidney_igan_g_634x14x14.npz: the reference test's own input (test/test_pilot.py:6-15) run through the
    reference's wasserstein_distance(data_type='Pathomics').  Shapes as test_pilot.py:26-28 asserts them, and the oracle's OT
    numbers on a few of the stored rows (the fixture's were produced by the same oracle behind the reference's loop: this
    guards the fixture against an oracle change, it does not pin POT -- ot_source says which it was)."""
    g = load_golden(GOLDEN_REAL)
    N, K = g["proportions"].shape
    assert (N, K) == (634, 14) and g["emb"].shape == (24227, 14) and len(g["real_labels"]) == N
    rs = int(g["row_step"])
    assert g["emd_unreg"].shape == g["emd_reg"].shape == (len(range(0, N, rs)), N)
    assert sorted(set(g["real_labels"])) == ["30-60", "<30", ">60"]
    M = g["cost"] / g["cost"].max()                                   # Trajectory.py:101
    rows = [0, 30, 120]
    for r in rows:
        np.testing.assert_allclose(O.emd_grid(g["proportions"], M, row_begin=r, row_end=r + 1)[0], g["emd_unreg"][r // rs], atol=1e-13)
        np.testing.assert_allclose(O.sinkhorn_grid(g["proportions"], M, 0.1, row_begin=r, row_end=r + 1)[0], g["emd_reg"][r // rs], atol=1e-13)


def test_oracle_prior_uses_c_minus_one():
    annot = pd.DataFrame({"cell_type": list("aabbbc"), "sampleID": list("xxyyyz"), "status": list("ppqqqr")})
    rep, _ = O.cluster_representations(annot["cell_type"], annot["sampleID"], regulizer=0.5)
    prior = 0.5 * np.array([2, 3, 1]) / 5.0                        # n_k / (C - 1), C = 6
    np.testing.assert_array_equal(rep["x"], (np.array([2., 0, 0]) + prior) / (2 + sum(prior)))
    np.testing.assert_array_equal(rep["z"], (np.array([0., 0, 1]) + prior) / (1 + sum(prior)))


@pytest.mark.parametrize("name", GOLDEN_CASES + [GOLDEN_REAL])
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


def test_a_wrong_embedding_key_fails_in_the_extraction_step():
    # argument handling only: the engine is never reached because extraction fails first on a bad key
    g = load_golden("c1_20x10x10")
    ad, _ = golden_adata(g)
    with pytest.raises(KeyError):
        tl.wasserstein_distance(ad, emb_matrix="X_PCA")            # reference default key, absent here
    assert ad.uns == {}


def test_return_sil_ari_is_refused_before_any_device_work(monkeypatch):
    """return_sil_ari=True (Trajectory.py:108-113) needs the Leiden clustering of the finished matrix (scanpy), a consumer
    outside this engine's scope.  Nothing of it is restated here: the reference's OWN Clustering is
    used when pilotpy (and its scanpy / leidenalg stack) is importable, and otherwise the call is refused up front -- nothing
    computed, nothing written to adata.uns, no GPU touched (this test runs on the CPU box, where pilotpy is not importable)."""
    import sys, types
    g = load_golden("c1_20x10x10")
    ad, _ = golden_adata(g)
    with pytest.raises(NotImplementedError, match="Clustering"):
        tl.wasserstein_distance(ad, emb_matrix="X_pca", return_sil_ari=True)
    assert ad.uns == {}
    assert not hasattr(tl, "Clustering")
    # with the reference importable the flag resolves to ITS function (looked up before any device work)
    calls = []
    fake = types.ModuleType("pilotpy.tools.Trajectory")
    fake.Clustering = lambda EMD, annot, metric="cosine", res=0.01, steper=0.01: calls.append(1)
    for name, mod in (("pilotpy", types.ModuleType("pilotpy")), ("pilotpy.tools", types.ModuleType("pilotpy.tools")), ("pilotpy.tools.Trajectory", fake)):
        monkeypatch.setitem(sys.modules, name, mod)
    assert tl._reference_clustering() is fake.Clustering and not calls


def test_result_frames_are_the_objects_the_reference_builds():
    """adata.uns['EMD_df'] / ['cost'] (Trajectory.py:518-521, :470-473) are built here with one constructor call instead of
    from_dict(A).T + three assignments (a third of the time on the reference test's 634-sample cohort): the SAME frame -- values
    (transposed: a non-symmetric matrix tells), dtypes, column and index labels, index name, axis types -- and not a view of the
    array it was made from."""
    import pandas as pd
    rng = np.random.default_rng(3)
    for names in (["s%d" % i for i in range(7)], [3, 1, 2, 10, 7, 5, 4], list(pd.Categorical(list("gfedcba")))):
        E = rng.random((7, 7))
        ref = pd.DataFrame.from_dict(E).T
        ref.columns = names
        ref["sampleID"] = names
        ref = ref.set_index("sampleID")
        got = tl._emd_frame(E, names)
        pd.testing.assert_frame_equal(got, ref, check_exact=True)
        assert got.index.name == "sampleID" and type(got.index) is type(ref.index) and type(got.columns) is type(ref.columns)
        assert got.index.dtype == ref.index.dtype and got.columns.dtype == ref.columns.dtype
        assert not np.shares_memory(got.to_numpy(), E)
        got.iloc[0, 1] = -1.0
        assert E[1, 0] != -1.0 and E[0, 1] != -1.0
        dis, cost = tl._cost_frame(E, np.asarray(names, dtype=object))
        refc = pd.DataFrame.from_dict(E).T
        refc.columns = np.asarray(names, dtype=object)
        refc["cell_types"] = np.asarray(names, dtype=object)
        refc = refc.set_index("cell_types")
        pd.testing.assert_frame_equal(cost, refc, check_exact=True)
        assert dis is E and cost.index.name == "cell_types"
    empty = tl._emd_frame(np.zeros((0, 0)), [])
    assert empty.shape == (0, 0) and empty.index.name == "sampleID"


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32, np.uint64])
@pytest.mark.parametrize("n,n_threads", [(0, 1), (1, 1), (1000, 1), (200_000, 3)])
def test_label_codes_native_pass_is_pandas_factorize(dtype, n, n_threads):
    """engine.label_codes (host pass of libpilot_ot.so, no device): codes in first-appearance order, -1 for missing, the row
    of every first appearance -- pd.factorize on the same labels; slices on several threads merge to the same numbering;
    a column with more distinct labels than the caller allows is refused (None)."""
    rng = np.random.default_rng(n + n_threads)
    hi = 100 if dtype == np.int8 else 5000
    ids = rng.integers(0, hi, n).astype(dtype)
    if dtype == np.uint64:
        ids = ids * np.uint64(64) + np.uint64(0x7F0000000000)
    missing = rng.random(n) < 0.02
    ids[missing] = 0 if dtype == np.uint64 else -1
    codes, first = engine.label_codes(ids, n_threads=n_threads)
    ser = pd.Series(ids.astype(np.float64))
    ser[missing] = np.nan
    want, uniques = pd.factorize(ser, sort=False, use_na_sentinel=True)
    assert codes.dtype == np.int32 and first.dtype == np.int64
    np.testing.assert_array_equal(codes, want)
    np.testing.assert_array_equal(ids[first].astype(np.float64), uniques)
    if n >= 1000:
        assert engine.label_codes(ids, max_uniques=10, n_threads=n_threads) is None
