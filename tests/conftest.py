import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("PILOT_AMD_NO_RESULTS_DIR", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["c1_20x10x10", "c2s_100x30x30", "ragged_categorical_12x7x5", "pathomics_15x6x8"]
# the reference's OWN test input (test/test_pilot.py:6-15): Tutorial/Datasets/Kidney_IgAN_G.h5ad, 634 patients x 14 clusters x 14
# morphometric features, run through the reference's wasserstein_distance(data_type='Pathomics'); rows 0::3 of the matrices stored
GOLDEN_REAL = "kidney_igan_g_634x14x14"
# reference-executed with NON-default options (metric, regulizer, reg are stored in the fixture)
GOLDEN_OPTION_CASES = ["opts_euclidean_18x10x6", "opts_cityblock_14x8x4"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of erroring in them (counting the
    devices does not initialise the runtime).  `-m gpu` on such a box therefore reports skips, never false passes."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    try:
        from pilot_amd import _lib
        if not os.path.exists(_lib.LIB_PATH):       # (a fresh checkout: build before asking the library for the device count)
            import __graft_entry__
            __graft_entry__.build()
        n = _lib.device_count()
    except Exception:
        n = 0
    if n < 1:
        skip = pytest.mark.skip(reason="no HIP device visible: gpu-marked test")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the native pieces once (hipcc cross-compiles on the CPU box; no-op when up to date)."""
    import __graft_entry__
    __graft_entry__.build()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_adata(g, categorical=False):
    """Rebuild the duck-typed AnnData the fixture was generated from."""
    import pandas as pd
    from pilot_amd.synthetic import Cohort
    pathomics = str(g["data_type"]) != "scRNA"
    cell_col = "Cell_type" if pathomics else "cell_types"
    obs = pd.DataFrame({cell_col: g["obs_cell"].astype(object), "sampleID": g["obs_sample"].astype(object),
                        "status": g["obs_status"].astype(object)})
    if "obs_cell_dtype" in g:            # the real dataset: the obs dtypes anndata gives (int64 clusters, stored category orders)
        obs[cell_col] = g["obs_cell"].astype(str(g["obs_cell_dtype"]))
        obs["sampleID"] = pd.Categorical(g["obs_sample"].astype(object), categories=list(g["sample_categories"]))
        obs["status"] = pd.Categorical(g["obs_status"].astype(object), categories=list(g["status_categories"]))
    if categorical:
        for c in obs.columns:
            obs[c] = obs[c].astype("category")
    ad = Cohort(g["emb"], obs, emb_key="X_pca")
    if "var_names" in g:
        ad.var_names = [str(v) for v in g["var_names"]]
    return ad, cell_col


def frame_digests(uns):
    """SHA-256 of adata.uns['data'] / ['annot'] exactly as tests/golden/gen_golden.py::frame_digests takes them of the
    reference's frames (values + dtype + column names of `data`; the three label columns of `annot` as strings)."""
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


def load_golden_pack(name="random_pack"):
    """Fixtures packed into one file (tests/golden/gen_golden.py): a list of per-case dicts like load_golden's."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    n = int(z["n_cases"])
    return [{k[len("c%d_" % i):]: z[k] for k in z.files if k.startswith("c%d_" % i)} for i in range(n)]


class _Switches:
    """Test switches of libpilot_ot.so (pilot_ot_test_switch): what the GPU tests use to force a kernel variant."""

    def setenv(self, name, value):
        from pilot_amd import _lib
        _lib.test_switch(name, value)

    def delenv(self, name, raising=True):
        from pilot_amd import _lib
        _lib.test_switch(name, None)


@pytest.fixture
def switches():
    s = _Switches()
    yield s
    from pilot_amd import _lib
    _lib.test_switch(None)                 # every switch cleared, whatever the test left set


def account_for_absorb_on_last(Eg, Eo, last_g, last_o, K, tol, max_one_sided_frac=1e-4):
    """POT returns the transport cost scaled by 1/K^2 when a tau-absorption falls on a pair's FINAL update (u, v are reset to
    1/K and the plan is rebuilt from them).  Every pair of a grid is held to `tol` by a stated rule -- none is excluded:

    * flagged by neither side, or by both: |E_gpu - E_oracle| <= tol (both unscaled, or both scaled);
    * flagged by the oracle alone (the f32 kernel stopped at an earlier check, before that absorption): the GPU returned the
      unscaled cost, |E_gpu - K^2 E_oracle| <= tol;
    * flagged by the GPU alone (its earlier last update was an absorbing one): |K^2 E_gpu - E_oracle| <= tol;
    * the one-sided pairs are at most `max_one_sided_frac` of the grid (at least one pair is always allowed).

    Returns (n_both, n_oracle_only, n_gpu_only)."""
    Eg, Eo = np.asarray(Eg), np.asarray(Eo)
    both, only_o, only_g = last_g & last_o, last_o & ~last_g, last_g & ~last_o
    plain = ~(only_o | only_g)
    d = np.abs(Eg - Eo)
    assert d[plain].max() <= tol, "pairs flagged alike: max|gpu - oracle| = %.3e" % d[plain].max()
    if only_o.any():
        assert np.abs(Eg - K * K * Eo)[only_o].max() <= tol, "oracle-only absorb-on-last pairs: %.3e" % np.abs(Eg - K * K * Eo)[only_o].max()
    if only_g.any():
        assert np.abs(K * K * Eg - Eo)[only_g].max() <= tol, "GPU-only absorb-on-last pairs: %.3e" % np.abs(K * K * Eg - Eo)[only_g].max()
    n_one = int(only_o.sum() + only_g.sum())
    allowed = max(1, int(np.ceil(max_one_sided_frac * Eg.size)))
    assert n_one <= allowed, "%d one-sided absorb-on-last pairs of %d (allowed %d)" % (n_one, Eg.size, allowed)
    return int(both.sum()), int(only_o.sum()), int(only_g.sum())
