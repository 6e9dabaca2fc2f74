"""``pilot_amd.tl`` -- host-side mirror of the ``pilotpy.tl`` functions on the Wasserstein path.

Same names, arguments, defaults, ``adata.uns`` keys and Python types as the reference
(``pilotpy/tools/Trajectory.py``), so code written against ``pilotpy.tl.wasserstein_distance``
runs unchanged; the pair loop and the centroid distance matrix execute on the MI355X through
``libpilot_ot.so`` (``pilot_amd.engine``).  Nothing here falls back to a CPU solver.

=============================================  ==========================================
this module                                    reference (``/root/reference/pilotpy/tools/``)
=============================================  ==========================================
wasserstein_distance                           Trajectory.py:36-115
extract_data_anno_scRNA_from_h5ad              Trajectory.py:234-266
extract_data_anno_pathomics_from_h5ad          Trajectory.py:270-299
set_path_for_results                           Trajectory.py:146-164
Cluster_Representations                        Trajectory.py:377-436
cost_matrix                                    Trajectory.py:441-475
wasserstein_d                                  Trajectory.py:479-523
return_real_labels                             Trajectory.py:617-642
Sil_computing                                  Trajectory.py:592-612
Precomputed_distance                           Trajectory.py:1687-1727
diffusion_kernel                               plot/ploting.py:95-110 (the dense part of pl.trajectory)
=============================================  ==========================================
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np
import pandas as pd

from . import engine

path_to_results = None  # module global, as in the reference (Trajectory.py:254)

# engine options that have no counterpart in the reference signature; set via `engine_options`
_DEFAULT_ENGINE_OPTIONS = dict(precision="auto")


def set_path_for_results():
    """Create ``Results_PILOT/plots`` under the cwd and return it (Trajectory.py:146-164).

    The reference does this as a side effect of every ``wasserstein_distance`` call; set
    ``PILOT_AMD_NO_RESULTS_DIR=1`` to skip the mkdir (the path is still returned).
    """
    if os.environ.get("PILOT_AMD_NO_RESULTS_DIR", "") not in ("", "0"):
        return "Results_PILOT/plots"
    if not os.path.exists("Results_PILOT/plots"):
        os.makedirs("Results_PILOT/plots")
    return "Results_PILOT/plots"


def extract_data_anno_scRNA_from_h5ad(adata, emb_matrix="PCA", clusters_col="cell_type",
                                      sample_col="sampleID", status="status"):
    """(data, annot) for scRNA input: ``adata.obsm[emb_matrix]`` as a frame with columns
    ``PCA_1..D``; ``adata.obs`` columns renamed ``cell_type, sampleID, status`` (Trajectory.py:234-266)."""
    global path_to_results
    data = adata.obsm[emb_matrix]
    col_add = ["PCA_" + str(i) for i in range(1, data.shape[1] + 1)]
    # The reference's frame is an independent copy of the embedding (DataFrame(...).reset_index(drop=True), :249-252).  The
    # frame built here VIEWS adata.obsm[emb_matrix]; wasserstein_distance replaces its array by a private copy (made on a
    # helper thread beside the device work) before it stores the frame in adata.uns['data'].  Direct callers of this helper
    # get the view.
    data = pd.DataFrame(data, columns=col_add)
    annot = _annot_frame(adata.obs, clusters_col, sample_col, status)
    path_to_results = set_path_for_results()
    return data, annot


def extract_data_anno_pathomics_from_h5ad(adata, var_names=[], clusters_col="Cell_type",
                                          sample_col="sampleID", status="status"):
    """(data, annot) for pathomics input: ``adata[:, var_names].X`` (Trajectory.py:270-299)."""
    global path_to_results
    data = adata[:, var_names].X
    data = pd.DataFrame(data, columns=var_names)
    annot = _annot_frame(adata.obs, clusters_col, sample_col, status)
    path_to_results = set_path_for_results()
    return data, annot


def _annot_frame(obs, clusters_col, sample_col, status):
    """``obs[[clusters_col, sample_col, status]]`` renamed to ``cell_type, sampleID, status`` with a fresh RangeIndex
    (Trajectory.py:257-262): one column selection (new arrays, dtypes kept); the shallow copy on top only drops pandas'
    "this is a slice of another frame" marker, so that later column assignments on ``adata.uns['annot']`` neither warn
    (SettingWithCopyWarning under the reference's pandas 2.0) nor touch ``adata.obs``."""
    annot = obs[[clusters_col, sample_col, status]].copy(deep=False)
    annot.columns = ["cell_type", "sampleID", "status"]
    annot.index = pd.RangeIndex(len(annot))
    return annot


def _first_appearance_codes(series):
    """(codes, uniques) with uniques in first-appearance order, as ``Series.unique()`` gives; missing values get -1.

    ``pd.factorize`` on an object column hashes millions of Python objects (~25 ms per 1.8 M cells and column) and holds the
    interpreter lock.  Two shortcuts, both one native pass of ``libpilot_ot.so`` (``engine.label_codes``, int32 codes out,
    no lock held -- ``wasserstein_distance`` runs its two columns side by side):

    * categorical columns (what AnnData gives for ``obs`` labels): the integer codes are there already, only their
      numbering is changed to first appearance;
    * object columns: the labels of 1.8 M cells are a few hundred distinct Python objects repeated (AnnData / pandas /
      numpy fancy indexing copy POINTERS), so the column is numbered by object identity and only the distinct objects
      are then factorised by value (equal strings at different addresses merge there).
    """
    if isinstance(series.dtype, pd.CategoricalDtype):
        values = series.array
        raw = values.codes                                            # int8 / int16 / int32 (int64 beyond 2^31 categories)
        if raw.dtype.itemsize <= 4:
            codes, first = engine.label_codes(raw, max_uniques=len(values.categories) + 1)
            return codes, np.asarray(values.categories[raw[first]])
        seen = pd.unique(raw[raw >= 0]) if (raw < 0).any() else pd.unique(raw)       # category numbers, first appearance
        remap = np.full(len(values.categories) + 1, -1, dtype=np.int64)              # (slot -1 serves the missing code)
        remap[seen] = np.arange(len(seen))
        return remap[raw], np.asarray(values.categories[seen])
    arr = series.to_numpy()
    if arr.dtype == object and arr.ndim == 1 and arr.strides == (arr.itemsize,) and arr.size:
        ptrs = np.ctypeslib.as_array((ctypes.c_size_t * arr.size).from_address(arr.ctypes.data))      # (arr stays alive here)
        #  the shortcut only pays while the labels are few shared objects: a column built row by row holds one
        # str object per cell (as many pointers as cells) -- such a column goes the plain way
        got = engine.label_codes(ptrs.view(np.uint64), max_uniques=max(4096, arr.size // 8))
        if got is None:
            codes, uniques = pd.factorize(series, sort=False, use_na_sentinel=True)
            return codes, np.asarray(uniques)
        pcodes, first = got
        objs = arr[first]                                              # the distinct objects, in order of first appearance
        vcodes, uniques = pd.factorize(objs, sort=False, use_na_sentinel=True)
        if len(uniques) == len(objs) and np.array_equal(vcodes, np.arange(len(objs))):
            return pcodes, np.asarray(uniques)                         # every object its own value: the identity codes stand
        lut = np.append(vcodes, -1).astype(np.int32)                   # (slot -1 serves the missing code)
        return lut[pcodes], np.asarray(uniques)
    codes, uniques = pd.factorize(series, sort=False, use_na_sentinel=True)
    return codes, np.asarray(uniques)


def Cluster_Representations(df, cell_col=0, sample_col=1, regulizer=0.2, normalization=True):
    """Proportions of clusters per sample (Trajectory.py:377-436).

    Returns an insertion-ordered dict ``{sampleID: float64[K]}``: samples and cell types in
    first-appearance order (``.unique()``, :402/:412); prior_k = regulizer * n_k / (C - 1)
    (:405-409, note C - 1); with ``normalization`` p = (counts + prior) / (sum counts + sum prior)
    (:428-430).  The host only factorises the two label columns into integer codes; the histogram and
    the smoothing run on the device (``engine.proportions``) and are bit-identical to the reference.
    """
    cell_col = df.columns[cell_col]
    sample_col = df.columns[sample_col]
    ccodes, cells = _first_appearance_codes(df[cell_col])
    scodes, samples = _first_appearance_codes(df[sample_col])
    K, N = len(cells), len(samples)
    if str(cell_col) != "cell_type":
        # the reference counts n_k in the column literally named 'cell_type' (:403-407); with a different
        # cell_col the two must at least label the same partition
        other, _ = _first_appearance_codes(df["cell_type"])
        if not np.array_equal(other, ccodes):
            raise NotImplementedError("Cluster_Representations: cell_col differs from the 'cell_type' column")
    return _proportions_from_codes(ccodes, scodes, samples, K, len(df), regulizer, normalization)


def _proportions_from_codes(ccodes, scodes, samples, K, n_total, regulizer, normalization):
    N = len(samples)
    P = engine.proportions(ccodes, scodes, N, K, regulizer=regulizer, normalization=normalization, n_total=n_total)
    return dict(zip(samples, np.array(P, dtype=np.float64, order="C", copy=True)))


def cost_matrix(annot, data, metric="cosine"):
    """Ground-cost matrix between cell types (Trajectory.py:441-475): per-type column-wise MEDIAN
    centroids, pairwise ``metric`` distance (device kernel; scipy ``pdist`` names), returned as
    ``(ndarray K x K, DataFrame indexed 'cell_types')``.  NOT normalised (the caller divides by max)."""
    codes, cells = _first_appearance_codes(annot[annot.columns[0]])
    X = data.to_numpy() if isinstance(data, pd.DataFrame) else np.asarray(data)
    return _cost_from_codes(X, codes, cells, metric)


def _cost_from_codes(X, codes, cells, metric):
    centroids = engine.centroid_medians(X, codes, len(cells))
    return _cost_frame(engine.pdist_square(centroids, metric=metric), cells)


def _labelled_square_frame(A, names, index_name):
    """The frame the reference builds as ``from_dict(A).T`` + ``columns = names`` + a names column + ``set_index`` (Trajectory.py:470-473,
    :518-521), in ONE constructor call: the transposed values in a block of their own (the frame must not alias the array it was made
    from), ``names`` as columns and as the named index.  Same object for every consumer -- values, dtypes, both axes and the index
    name are asserted equal to the reference's construction in tests/test_host_logic.py; a third of the time at 634 samples."""
    # (the reference's columns pass through a frame that also holds the str-labelled names column: their Index is of object dtype
    # whatever the labels are, while the index takes the dtype pandas infers for the labels)
    return pd.DataFrame(np.ascontiguousarray(np.asarray(A).T), index=pd.Index(names, name=index_name),
                        columns=pd.Index(names, dtype=object), copy=False)


def _cost_frame(dis, cells):
    return dis, _labelled_square_frame(dis, cells, "cell_types")


def wasserstein_d(Clu_rep, cost, regularized="unreg", reg=0.1, engine_options=None):
    """Wasserstein distances among all ordered sample pairs (Trajectory.py:479-523).

    ``regularized == "unreg"`` -> exact OT (``ot.emd2``); anything else -> entropic OT with POT's
    ``sinkhorn_stabilized`` semantics (``ot.sinkhorn2``).  Returns ``(EMD float64 N x N, DataFrame)``;
    the frame is ``DataFrame.from_dict(EMD).T`` indexed by sample id, exactly as the reference builds it.
    """
    samples_id = list(Clu_rep.keys())
    if len(samples_id) == 0:
        EMD = np.zeros((0, 0))
    else:
        P = np.stack([np.asarray(Clu_rep[s], dtype=np.float64) for s in samples_id])
        EMD = _pair_grid(P, cost, regularized, reg, engine_options)
    return EMD, _emd_frame(EMD, samples_id)


def _pair_grid(P, cost, regularized, reg, engine_options):
    """The N x N matrix of all ordered pairs on the device(s): the loop of Trajectory.py:505-515."""
    opts = dict(_DEFAULT_ENGINE_OPTIONS)
    opts.update(engine_options or {})
    cost = np.asarray(cost, dtype=np.float64)
    if regularized == "unreg":
        # ot.emd2 (POT 0.9: check_marginals=True) refuses histograms of different mass before it rescales them:
        # np.testing.assert_almost_equal(a.sum(0), b.sum(0), decimal=6).  With normalization=False the reference's
        # proportions are raw counts and this is what stops it (Trajectory.py:428-436, :511).
        sums = P.sum(1)
        if sums.size and float(sums.max() - sums.min()) >= 1.5e-6:
            raise AssertionError("a and b vector must have the same sum (sample masses range from %g to %g)"
                                 % (sums.min(), sums.max()))
    multi = {k: opts.pop(k) for k in ("devices", "n_devices", "gather") if k in opts}
    if multi.get("devices") is not None or (multi.get("n_devices") or 1) > 1:
        # the pair grid row-sharded over several GPUs of this node, one RCCL all-gather (pilot_amd.multi)
        from . import multi as _multi
        if regularized == "unreg":
            return _multi.emd_grid_multi(P, cost, **multi)
        return _multi.sinkhorn_grid_multi(P, cost, reg, **multi, **opts)
    if regularized == "unreg":
        return engine.emd_grid(P, cost)
    return engine.sinkhorn_grid(P, cost, reg, **opts)


def _emd_frame(EMD, samples_id):
    return _labelled_square_frame(EMD, samples_id, "sampleID")


def return_real_labels(df, category="status", sample_col=1):
    """First status value of every sample, samples in first-appearance order (Trajectory.py:617-642)."""
    scodes, samples = _first_appearance_codes(df[df.columns[sample_col]])
    return _labels_from_codes(scodes, len(samples), df[category].to_numpy())


def _labels_from_codes(scodes, n_samples, cond):
    first_row = np.full(n_samples, -1, dtype=np.int64)
    idx = np.flatnonzero(scodes >= 0)
    # first occurrence of each sample code: assign the row numbers back to front, the earliest row is written last
    first_row[scodes[idx][::-1]] = idx[::-1]
    return [cond[r] for r in first_row]


def _values_at(series, rows):
    """``[series.iloc[r] for r in rows]`` as ``Series.unique()[0]`` hands the values out, without materialising a
    categorical column (1.8 M Python references for a few hundred lookups)."""
    if isinstance(series.dtype, pd.CategoricalDtype):
        values = series.array
        codes = values.codes[np.asarray(rows, dtype=np.int64)]
        cats = values.categories.to_numpy()             # (one array lookup: `categories[c]` per element is 0.4 us of pandas each)
        out = list(cats[np.where(codes >= 0, codes, 0)]) if len(cats) else [np.nan] * len(codes)
        for i in np.flatnonzero(codes < 0):
            out[i] = np.nan
        return out
    values = series.to_numpy()
    return [values[int(r)] for r in rows]


_libc = None


def _empty_like_huge(X):
    """``np.empty_like(X)`` whose pages the kernel may back with 2 MB huge pages (``madvise(MADV_HUGEPAGE)`` where transparent
    huge pages are in ``madvise`` mode): the private copy of a 216 MB embedding is 52 000 first-touch page faults otherwise,
    which is most of what the copy costs."""
    global _libc
    huge = 2 << 20
    if X.nbytes < 8 * huge or not X.flags.c_contiguous:
        return np.empty_like(X)
    try:
        if _libc is None:
            _libc = ctypes.CDLL(None, use_errno=True)
        raw = np.empty(X.nbytes + huge, dtype=np.uint8)
        off = (-raw.ctypes.data) % huge
        _libc.madvise(ctypes.c_void_p(raw.ctypes.data + off), ctypes.c_size_t((X.nbytes // huge) * huge), 14)       # MADV_HUGEPAGE; advisory
        return raw[off:off + X.nbytes].view(X.dtype).reshape(X.shape)
    except Exception:
        return np.empty_like(X)


class _DeviceWorker:
    """ONE long-lived helper thread that runs the device chain of ``wasserstein_distance`` (the library caches its
    workspace per calling thread: a fresh thread per call would rebuild it every time)."""

    class _Job:
        def __init__(self):
            self.done = threading.Event()
            self.error = None

        def wait(self):
            self.done.wait()

    def __init__(self):
        import queue
        self.q = queue.SimpleQueue()
        self.thread = threading.Thread(target=self._loop, name="pilot_amd_device_chain", daemon=True)
        self.thread.start()

    def _loop(self):
        from . import _lib
        while True:
            job, device, fn, args = self.q.get()
            try:
                _lib.check(_lib.load().pilot_ot_set_device(device))      # (the caller's current device)
                fn(*args)
            except BaseException as e:                                     # handed to the submitting thread
                job.error = e
            finally:
                job.done.set()

    def submit(self, device, fn, *args):
        job = self._Job()
        self.q.put((job, device, fn, args))
        return job


_worker = None
_worker_lock = threading.Lock()


def _device_worker():
    global _worker
    with _worker_lock:
        if _worker is None or not _worker.thread.is_alive():
            _worker = _DeviceWorker()
        return _worker


_SIL_ARI_MESSAGE = ("return_sil_ari=True: the ARI is the Rand index of a Leiden clustering of the finished matrix "
                    "(pilotpy.tl.Clustering, Trajectory.py:525-588: scanpy neighbors + leiden) -- a consumer of adata.uns['EMD'] "
                    "outside this engine's scope (SURVEY.md section 2 #6), so it is run by the reference's OWN function and "
                    "pilotpy (with scanpy / leidenalg) is not importable here: %s.  Run wasserstein_distance without it; the "
                    "silhouette alone is tl.Sil_computing(EMD / EMD.max(), adata.uns['real_labels']) (INTEGRATION.md, 'Sil / ARI')")


def _reference_clustering():
    """``pilotpy.tools.Trajectory.Clustering`` (Trajectory.py:525-588) when the reference package and its scanpy / leidenalg
    stack are installed next to this engine -- imported lazily, only for ``return_sil_ari=True``; nothing of it is restated
    here.  Raises NotImplementedError (before any device work) when it is not importable."""
    try:
        from pilotpy.tools.Trajectory import Clustering
    except Exception as e:            # ImportError, or whatever the eager import chain of pilotpy raises
        raise NotImplementedError(_SIL_ARI_MESSAGE % (repr(e),)) from e
    return Clustering


def Sil_computing(EMD, real_labels, metric="cosine"):
    """Silhouette score of a labelling of the samples, the rows of ``EMD`` being the points (Trajectory.py:592-612:
    ``sklearn.metrics.silhouette_score(EMD, real_labels, metric=metric)``; callers pass ``EMD / EMD.max()``,
    plot/ploting.py:324).  Row-to-row distances and the score are chained on the device (the matrix goes up once, N per-sample
    scores come back); metric "cosine" or "euclidean".  ``EMD`` may also be an ``engine.DeviceMatrix`` (a result still in HBM)."""
    return engine.silhouette_of_rows(EMD, real_labels, metric=metric)


def diffusion_kernel(adata, epsilon=1, knn=64):
    """The dense part of ``pl.trajectory`` (plot/ploting.py:95-110): ``EMD / EMD.max()``, Euclidean distances between its
    rows, and pydiffmap's k-nearest-neighbour Gaussian kernel ``exp(-d^2 / (4 epsilon))``, all on the device.  Returns
    ``(EMD_normalised_row_distances, kernel_matrix)``; the eigen-decomposition stays with pydiffmap / scipy."""
    return engine.diffusion_kernel_of_rows(adata.uns["EMD"], k=knn, epsilon=epsilon)


def wasserstein_distance(adata, emb_matrix="X_PCA", clusters_col="cell_types", sample_col="sampleID",
                         status="status", metric="cosine", regulizer=0.2, normalization=True,
                         regularized="unreg", reg=0.1, res=0.01, steper=0.01, data_type="scRNA",
                         return_sil_ari=False, engine_options=None):
    """Wasserstein distance among samples (Trajectory.py:36-115); results go to ``adata.uns``:
    ``data, annot, proportions, cost, EMD_df, EMD, real_labels``.

    ``engine_options`` (not in the reference): dict forwarded to the device engine, e.g.
    ``{"precision": "fp64"}``; ``{"n_devices": G}`` (or ``{"devices": [0, 1, ...]}``) row-shards the pair grid over G GPUs
    of this node with one RCCL all-gather -- same bits as the single-GPU matrix.
    """
    # (Trajectory.py:108-113) the Leiden / ARI tail is the reference's own function: found, or refused, BEFORE any device work
    reference_clustering = _reference_clustering() if return_sil_ari else None
    if metric not in engine._lib.METRICS:
        raise NotImplementedError("metric %r: the device kernel implements scipy's pdist names %s" % (metric, sorted(engine._lib.METRICS)))
    global path_to_results
    # the embedding frame first (extract_data_anno_*_from_h5ad, :234-299): its bytes start moving at once
    if data_type == "scRNA":
        X = adata.obsm[emb_matrix]
        data = pd.DataFrame(X, columns=["PCA_" + str(i) for i in range(1, X.shape[1] + 1)])
    else:
        var_names = list(adata.var_names)
        data = pd.DataFrame(adata[:, var_names].X, columns=var_names)
    # While the label columns are numbered (native passes, side by side), helper threads move bytes: the embedding to the device
    # (H2D) and into the private copy that adata.uns['data'] holds, like the reference's.
    X = data.to_numpy()
    upload = engine.EmbeddingUpload(X)
    own = _empty_like_huge(X)
    n_copy = min(8, max(1, (os.cpu_count() or 2) - 2)) if X.size >= (1 << 22) else (1 if X.nbytes >= (4 << 20) else 0)
    bounds = np.linspace(0, X.shape[0], max(n_copy, 1) + 1).astype(np.int64)
    copiers = [threading.Thread(target=np.copyto, args=(own[a:b], X[a:b]), name="pilot_amd_data_copy")
               for a, b in zip(bounds[:-1], bounds[1:])] if n_copy else []
    if not n_copy:
        np.copyto(own, X)          # (a small embedding: a thread costs more than the copy)
    for t in copiers:
        t.start()
    dev = {}

    def device_chain(ccodes, scodes, n_samples, n_types):
        # runs with the GIL released almost throughout (ctypes calls): proportions + first rows + medians, pdist, pair grid
        try:
            dev["P"], dev["first"], centroids = upload.prepass(ccodes, scodes, n_samples, n_types, regulizer=regulizer,
                                                               normalization=normalization, n_total=len(ccodes))
            cost = engine.pdist_square(centroids, metric=metric)
            dev["cost"] = cost
            dev["EMD"] = _pair_grid(dev["P"], cost / cost.max(), regularized, reg, engine_options) if n_samples else np.zeros((0, 0))
        except BaseException as e:          # re-raised on the calling thread
            dev["error"] = e
    chain = None
    try:
        # the two label columns are numbered ONCE (first-appearance order, Trajectory.py:402,412), straight from adata.obs,
        # and shared by the three steps that the reference runs as separate pandas scans; the cell-type column on a helper
        # thread beside the sample column on this one when the columns are long enough to pay for the thread ...
        side = {}
        if len(adata.obs) >= (1 << 17):
            def number_types():
                try:
                    side["cells"] = _first_appearance_codes(adata.obs[clusters_col])
                except BaseException as e:
                    side["error"] = e
            helper = threading.Thread(target=number_types, name="pilot_amd_label_codes")
            helper.start()
            try:
                scodes, samples = _first_appearance_codes(adata.obs[sample_col])
            finally:
                helper.join()
            if "error" in side:
                raise side["error"]
            ccodes, cells = side["cells"]
        else:
            ccodes, cells = _first_appearance_codes(adata.obs[clusters_col])
            scodes, samples = _first_appearance_codes(adata.obs[sample_col])
        chain = _device_worker().submit(upload.device, device_chain, ccodes, scodes, len(samples), len(cells))
        # ... and while the device works, this thread does the GIL-bound part: the annotation frame (a copy of three object
        # columns of adata.obs, Trajectory.py:257-262)
        annot = _annot_frame(adata.obs, clusters_col, sample_col, status)
        path_to_results = set_path_for_results()
    finally:
        if chain is not None:
            chain.wait()
        upload.close()
        for t in copiers:
            t.join()
    if "error" in dev or chain.error is not None:
        raise dev.get("error") or chain.error
    # One private N x K block, a row view per sample (634 separate copies were a third of a millisecond).  Same values, keys and
    # order as the reference's dict (Trajectory.py:432-436); unlike its independent arrays the rows share a base array: `arr.base`
    # is the block and `arr.flags.owndata` is False, so a consumer that keeps one row keeps the block alive.
    proportions = dict(zip(samples, np.array(dev["P"], dtype=np.float64, order="C", copy=True)))
    first_rows = dev["first"]
    adata.uns["data"] = pd.DataFrame(own, columns=data.columns, copy=False)
    adata.uns["annot"] = annot
    adata.uns["proportions"] = proportions
    cost, cost_df = _cost_frame(dev["cost"], cells)
    adata.uns["cost"] = cost_df
    EMD = dev["EMD"]
    adata.uns["EMD_df"] = _emd_frame(EMD, list(proportions.keys()))
    adata.uns["EMD"] = EMD
    # first status value of every sample (return_real_labels, :617-642): the first row of a sample came out of the
    # device pass over the codes
    # (the values as ``Series.unique()[0]`` hands them out; ``.iloc`` per sample on a categorical column was 3 - 6 ms at 634 samples)
    real_labels = _values_at(annot["status"], first_rows)
    if reference_clustering is not None:      # Trajectory.py:108-113: the reference's Clustering on the finished matrix, the silhouette on the device
        _predicted, ARI, real_labels = reference_clustering(EMD / EMD.max(), annot, metric=metric, res=res, steper=steper)
        adata.uns["real_labels"] = real_labels
        adata.uns["Sil"] = Sil_computing(EMD / EMD.max(), real_labels, metric=metric)
        adata.uns["ARI"] = ARI
        return
    adata.uns["real_labels"] = real_labels


def Precomputed_distance(adata, distances, cost_df, features_matrix, emb_matrix="X_PCA",
                         clusters_col="cell_types", sample_col="sampleID", status="status", data_type="scRNA"):
    """Store externally computed distances in ``adata.uns`` (Trajectory.py:1687-1727; the reference
    reads an undefined ``data_type`` at :1716 -- here it is an explicit argument)."""
    if data_type == "scRNA":
        data, annot = extract_data_anno_scRNA_from_h5ad(adata, emb_matrix=emb_matrix, clusters_col=clusters_col,
                                                        sample_col=sample_col, status=status)
    else:
        data, annot = extract_data_anno_pathomics_from_h5ad(adata, var_names=list(adata.var_names),
                                                            clusters_col=clusters_col, sample_col=sample_col,
                                                            status=status)
    adata.uns["data"] = data
    adata.uns["annot"] = annot
    adata.uns["proportions"] = features_matrix
    adata.uns["cost"] = cost_df
    adata.uns["EMD"] = distances
    adata.uns["real_labels"] = return_real_labels(annot)


def cell_level_wasserstein(adata, emb_matrix="X_PCA", sample_col="sampleID", status="status", reg=0.1, scale=None,
                           num_iter_max=1000, stop_thr=1e-9):
    """EXTENSION -- not part of pilotpy (BASELINE config 5): Wasserstein-2 distance between
    samples computed on their raw cell clouds instead of on cell-type proportions.

    Every sample is the uniform measure on its cells in ``adata.obsm[emb_matrix]``; the ground cost is the squared
    Euclidean distance divided by ``scale`` (default: twice the mean squared distance of the cells to the global
    centroid, i.e. the expected squared distance between two random cells); entropic OT with regularisation ``reg`` is
    solved in the log domain on the device for all ordered sample pairs.  Writes ``adata.uns['EMD_cell']`` (ndarray
    N x N of transport costs), ``adata.uns['EMD_cell_df']`` and ``adata.uns['real_labels']``; returns nothing, like
    ``wasserstein_distance``.
    """
    X = np.asarray(adata.obsm[emb_matrix], dtype=np.float32)
    obs = adata.obs[[sample_col, status]].copy()
    obs.columns = ["sampleID", "status"]
    obs = obs.reset_index(drop=True)
    scodes, samples = _first_appearance_codes(obs["sampleID"])
    if (scodes < 0).any():
        raise ValueError("cells without a sample id")
    order = np.argsort(scodes, kind="stable")                 # group the cells of a sample, keep their order
    counts = np.bincount(scodes, minlength=len(samples))
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    Xg = np.ascontiguousarray(X[order])
    if scale is None:
        mu = Xg.mean(axis=0, dtype=np.float64)
        scale = 2.0 * float(((Xg.astype(np.float64) - mu) ** 2).sum(axis=1).mean())
    W = engine.cell_w2_grid(Xg, offsets, scale, reg, num_iter_max=num_iter_max, stop_thr=stop_thr)
    df = pd.DataFrame(W.T, columns=list(samples))
    df["sampleID"] = list(samples)
    df = df.set_index("sampleID")
    adata.uns["EMD_cell"] = W
    adata.uns["EMD_cell_df"] = df
    adata.uns["EMD_cell_scale"] = scale
    adata.uns["real_labels"] = return_real_labels(pd.DataFrame({"cell_type": 0, "sampleID": obs["sampleID"], "status": obs["status"]}))
