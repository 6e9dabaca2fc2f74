#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE's own code.

Runs only in the build container (needs /root/reference); the GPU box uses the committed .npz files.

What is reference-executed: ``pilotpy/tools/Trajectory.py`` is imported from /root/reference with a
``sys.meta_path`` stub finder standing in for the ~15 third-party packages that are not installed
(scanpy, seaborn, ... -- none of them is touched by the functions called here), and the real
``extract_data_anno_*``, ``Cluster_Representations``, ``cost_matrix``, ``return_real_labels`` and the
``wasserstein_d`` double loop run on pandas/numpy/scipy.

What is NOT reference-executed while POT is absent: the per-pair OT arithmetic.  A real ``ot`` (POT) is
used whenever it can be imported; otherwise the ``ot`` module the reference imports is a shim whose
``emd2`` / ``sinkhorn2`` call the CPU oracle (oracle/pilot_oracle.c).  Every fixture records which one
produced its numbers in ``ot_source`` ("pot==<version>" or "oracle-shim"); the tests print it.  With the
shim the fixtures pin the reference's data handling, ordering, loop and output layout -- and record the
oracle's numbers, so a later oracle change is caught -- but do not pin POT itself ("parity unpinned", see
DESIGN.md).  The day POT is importable here: regenerate, diff the capped and absorb-on-last pairs against
the oracle, and only then call the parity pinned.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

from oracle import oracle as O  # noqa: E402
from pilot_amd.synthetic import make_cells  # noqa: E402

REFERENCE = "/root/reference"
OT_SOURCE = None
MISSING = ["scanpy", "anndata", "seaborn", "pydiffmap", "sknetwork", "elpigraph", "adjustText", "gprofiler",
           "plotnine", "joypy", "shap", "rpy2", "gseapy", "leidenalg", "igraph", "statsmodels", "h5py",
           "matplotlib_venn", "networkx", "upsetplot", "pingouin"]


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        return _Stub(self.__name__ + "()")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        if top in MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


def import_reference():
    for name in list(MISSING):
        try:
            __import__(name)
            MISSING.remove(name)
        except Exception:
            pass
    global OT_SOURCE
    try:
        import ot as real_ot                                   # the real thing, if it ever becomes available
        if not (hasattr(real_ot, "emd2") and hasattr(real_ot, "sinkhorn2")):
            raise ImportError("not POT")
        OT_SOURCE = "pot==%s" % getattr(real_ot, "__version__", "unknown")
    except Exception:
        ot = types.ModuleType("ot")          # POT shim -> CPU oracle (see module docstring)
        ot.emd2 = lambda a, b, M, *args, **kw: O.emd2(a, b, M)

        def sinkhorn2(a, b, M, reg, method="sinkhorn", **kw):
            assert method == "sinkhorn_stabilized", method
            return O.sinkhorn2(a, b, M, reg)
        ot.sinkhorn2 = sinkhorn2
        sys.modules["ot"] = ot
        OT_SOURCE = "oracle-shim"
    print("ot_source:", OT_SOURCE)
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REFERENCE)
    import pilotpy.tools.Trajectory as T
    return T


def frame_digests(uns):
    """SHA-256 of what the reference leaves in adata.uns['data'] / ['annot'] (Trajectory.py:92-93): the values' bytes, dtype and
    column names of `data`; the three label columns of `annot` as strings.  The tests hold tl.wasserstein_distance to these."""
    import hashlib
    data, annot = uns["data"], uns["annot"]
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(data.to_numpy()).tobytes())
    h.update(("|" + str(data.to_numpy().dtype) + "|" + "\x1f".join(str(c) for c in data.columns)).encode())
    g = hashlib.sha256()
    g.update("\x1e".join(str(c) for c in annot.columns).encode())
    for c in annot.columns:
        g.update(("\x1d" + "\x1f".join(str(v) for v in annot[c].tolist())).encode())
    g.update(("|%d" % len(annot)).encode())
    return h.hexdigest(), g.hexdigest()


def run_case(T, name, adata, out_dir, data_type="scRNA", emb_key="X_pca", reg=0.1, clusters_col="cell_types", metric="cosine",
             regulizer=0.2, row_step=1, extra=None):
    """row_step > 1: only rows 0, row_step, ... of the two N x N matrices are stored (a 634-patient cohort would be 13 MB of
    float64); the frames' labels and everything else are stored whole."""
    obs = adata.obs
    results = {}
    for mode in ("unreg", "reg"):
        adata.uns = {}
        T.wasserstein_distance(adata, emb_matrix=emb_key, clusters_col=clusters_col, sample_col="sampleID",
                               status="status", regularized=mode, reg=reg, data_type=data_type, metric=metric, regulizer=regulizer)
        results[mode] = dict(adata.uns)
    u = results["unreg"]
    samples = list(u["proportions"].keys())
    cells = list(u["cost"].columns)
    data_sha, annot_sha = frame_digests(u)
    assert (data_sha, annot_sha) == frame_digests(results["reg"])
    for mode in results:       # the frame is from_dict(EMD).T with the sample ids on both axes (Trajectory.py:518-521)
        df = results[mode]["EMD_df"]
        assert np.array_equal(df.to_numpy(), results[mode]["EMD"].T) and list(df.columns) == samples and list(df.index) == samples
    np.savez_compressed(
        os.path.join(out_dir, name + ".npz"),
        data_sha256=np.asarray(data_sha), annot_sha256=np.asarray(annot_sha), row_step=np.asarray(row_step),
        **(extra or {}),
        emb=np.asarray(adata.obsm[emb_key] if emb_key in adata.obsm else adata.X),
        obs_cell=np.asarray(obs[clusters_col].astype(str), dtype=str), obs_sample=np.asarray(obs["sampleID"].astype(str), dtype=str),
        obs_status=np.asarray(obs["status"].astype(str), dtype=str),
        samples=np.asarray(samples, dtype=str), cells=np.asarray(cells, dtype=str),
        proportions=np.stack([u["proportions"][s] for s in samples]),
        cost=u["cost"].to_numpy(), cost_index_name=np.asarray(str(u["cost"].index.name)),
        real_labels=np.asarray(u["real_labels"], dtype=str),
        emd_unreg=u["EMD"][::row_step], emd_reg=results["reg"]["EMD"][::row_step],
        # (the frames are the transposes, asserted above; stored only with the full matrices)
        **({"emd_unreg_df": u["EMD_df"].to_numpy(), "emd_reg_df": results["reg"]["EMD_df"].to_numpy()} if row_step == 1 else {}),
        emd_df_index_name=np.asarray(str(u["EMD_df"].index.name)),
        reg=np.asarray(reg), data_type=np.asarray(data_type), metric=np.asarray(metric), regulizer=np.asarray(regulizer),
        uns_keys=np.asarray(sorted(u.keys()), dtype=str),
        ot_source=np.asarray(OT_SOURCE),
    )
    print(name, "N=%d K=%d C=%d" % (len(samples), len(cells), len(obs)), "EMD unreg max", u["EMD"].max(),
          "reg max", results["reg"]["EMD"].max())


def real_dataset_case(T, out_dir):
    """The reference's OWN test input (test/test_pilot.py:6-15): Tutorial/Datasets/Kidney_IgAN_G.h5ad through
    wasserstein_distance(clusters_col='Cell_type', sample_col='sampleID', status='status', data_type='Pathomics').
    scanpy / anndata / h5py are absent here, so the file is read by tests/golden/mini_h5.py and wrapped in the same
    duck-typed AnnData as the synthetic cohorts, with the obs dtypes anndata would give (int64 cell types, categorical
    sample ids and status with the stored category order).  634 patients x 14 glomerulus clusters x 14 morphometric
    features, 24 227 glomeruli: real labels, real K, real (heavily duplicated, tiny) cohorts."""
    import mini_h5
    from pilot_amd.synthetic import Cohort
    path = os.path.join(REFERENCE, "Tutorial", "Datasets", "Kidney_IgAN_G.h5ad")
    d = mini_h5.read_h5ad(path)
    obs = pd.DataFrame({
        "Cell_type": d["obs"]["Cell_type"],
        "sampleID": pd.Categorical(d["obs"]["sampleID"][0], categories=list(d["obs"]["sampleID"][1])),
        "status": pd.Categorical(d["obs"]["status"][0], categories=list(d["obs"]["status"][1])),
    }, index=pd.Index(d["obs_names"]))
    ad = Cohort(d["X"], obs)
    ad.var_names = [str(v) for v in d["var_names"]]
    extra = dict(obs_cell_dtype=np.asarray(str(obs["Cell_type"].dtype)), sample_categories=np.asarray(list(d["obs"]["sampleID"][1]), dtype=str),
                 status_categories=np.asarray(list(d["obs"]["status"][1]), dtype=str), var_names=np.asarray(ad.var_names, dtype=str),
                 source=np.asarray("Tutorial/Datasets/Kidney_IgAN_G.h5ad (reference test/test_pilot.py:6-15), read by tests/golden/mini_h5.py"))
    run_case(T, "kidney_igan_g_634x14x14", ad, out_dir, data_type="Pathomics", clusters_col="Cell_type", row_step=3, extra=extra)


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    T = import_reference()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as scratch:   # the reference mkdirs Results_PILOT/plots in cwd
        os.chdir(scratch)
        try:
            # c1: BASELINE config 1 (20 x 10 x 10)
            run_case(T, "c1_20x10x10", make_cells(20, 10, 10, seed=0, cells_per_patient=200), out_dir)
            # c2-shaped (100 x 30 x 30) with 100 cells per patient to keep the fixture small
            run_case(T, "c2s_100x30x30", make_cells(100, 30, 30, seed=1, cells_per_patient=100), out_dir)
            # ragged: shuffled cell order, categorical dtype columns, types missing from some samples
            ad = make_cells(12, 7, 5, seed=7, cells_per_patient=40)
            perm = np.random.default_rng(7).permutation(len(ad.obs))
            ad.obs = ad.obs.iloc[perm].reset_index(drop=True)
            ad.obsm["X_pca"] = ad.obsm["X_pca"][perm]
            ad.X = ad.obsm["X_pca"]
            for c in ("cell_types", "sampleID", "status"):
                ad.obs[c] = ad.obs[c].astype("category")
            run_case(T, "ragged_categorical_12x7x5", ad, out_dir)
            # pathomics branch (adata.X over var_names), float64 features, euclidean-scale data
            ad = make_cells(15, 6, 8, seed=11, cells_per_patient=60)
            ad.X = ad.X.astype(np.float64) * 3.0 + 1.0
            ad.obsm["X_pca"] = ad.X
            ad.obs = ad.obs.rename(columns={"cell_types": "Cell_type"})
            run_case(T, "pathomics_15x6x8", ad, out_dir, data_type="Pathomics", clusters_col="Cell_type")
            # non-default options: metric, regulizer, reg; very unbalanced patients (one has a single cell), a cell type seen
            # in one patient only, numeric-looking labels whose first-appearance order is neither numeric nor lexicographic
            ad = make_cells(18, 9, 6, seed=21, cells_per_patient=50)
            rng = np.random.default_rng(21)
            keep = np.ones(len(ad.obs), dtype=bool)
            sid = ad.obs["sampleID"].to_numpy()
            first = np.flatnonzero(sid == sid[0])
            keep[first[1:]] = False                                        # patient 0: one cell
            second = np.flatnonzero(sid == np.unique(sid)[3])
            keep[second[5:]] = False                                       # another: five cells
            ad.obs = ad.obs[keep].reset_index(drop=True)
            ad.obsm["X_pca"] = ad.obsm["X_pca"][keep]
            ad.X = ad.obsm["X_pca"]
            relabel = {c: str(v) for c, v in zip(pd.unique(ad.obs["cell_types"]), [10, 3, 7, 21, 1, 100, 12, 5, 30])}
            ad.obs["cell_types"] = ad.obs["cell_types"].map(relabel)
            lone = ad.obs.index[ad.obs["sampleID"] == pd.unique(ad.obs["sampleID"])[7]][:3]
            ad.obs.loc[lone, "cell_types"] = "999"                         # a type that only one patient has
            run_case(T, "opts_euclidean_18x10x6", ad, out_dir, metric="euclidean", regulizer=0.5, reg=0.05)
            ad2 = make_cells(14, 8, 4, seed=22, cells_per_patient=35)
            run_case(T, "opts_cityblock_14x8x4", ad2, out_dir, metric="cityblock", regulizer=1.0, reg=0.3)
            # a pack of small random cohorts with random options (one file: tests/golden/random_pack.npz)
            rng = np.random.default_rng(2026)
            pack = {}
            metrics = ["cosine", "euclidean", "sqeuclidean", "cityblock", "chebyshev", "correlation", "braycurtis", "canberra"]
            n_pack = 12
            for i in range(n_pack):
                Np, K, D = int(rng.integers(3, 13)), int(rng.integers(2, 9)), int(rng.integers(2, 9))
                ad = make_cells(Np, K, D, seed=100 + i, cells_per_patient=int(rng.integers(8, 40)))
                if rng.random() < 0.5:                                     # shuffled cell order
                    perm = rng.permutation(len(ad.obs))
                    ad.obs = ad.obs.iloc[perm].reset_index(drop=True)
                    ad.obsm["X_pca"] = ad.obsm["X_pca"][perm]
                if rng.random() < 0.5:
                    ad.obsm["X_pca"] = ad.obsm["X_pca"].astype(np.float64) * float(rng.choice([1.0, 7.5])) + float(rng.choice([0.0, 2.0]))
                ad.X = ad.obsm["X_pca"]
                if rng.random() < 0.4:
                    for c in ("cell_types", "sampleID", "status"):
                        ad.obs[c] = ad.obs[c].astype("category")
                opts = dict(metric=str(rng.choice(metrics)), regulizer=float(rng.choice([0.05, 0.2, 1.0, 3.0])), reg=float(rng.choice([0.05, 0.1, 0.5, 1.0])))
                with tempfile.TemporaryDirectory() as one:
                    run_case(T, "case", ad, one, **opts)
                    z = np.load(os.path.join(one, "case.npz"))
                    for k in z.files:
                        pack["c%d_%s" % (i, k)] = z[k]
            pack["n_cases"] = np.asarray(n_pack)
            np.savez_compressed(os.path.join(out_dir, "random_pack.npz"), **pack)
            real_dataset_case(T, out_dir)
        finally:
            os.chdir(cwd)


if __name__ == "__main__":
    main()
