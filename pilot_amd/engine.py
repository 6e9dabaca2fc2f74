"""NumPy-level face of the device engine (host buffers in, host buffers out).

These are the array-level operators ``tl.wasserstein_d`` / ``tl.cost_matrix`` are built on; they go
straight to ``libpilot_ot.so`` through ctypes.  No torch, no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib

# POT defaults of ot.sinkhorn2 / sinkhorn_stabilized (what Trajectory.py:515 runs with)
NUM_ITER_MAX = 1000
STOP_THR = 1e-9
TAU = 1e3
CHECK_PERIOD = 20


def _as_f64(x, name):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if not np.all(np.isfinite(a)):
        raise ValueError("%s contains NaN or inf" % name)
    return a


def n_rows_of(N, row_begin, row_end, row_step):
    return len(range(row_begin, N if row_end is None else row_end, row_step))


def sinkhorn_grid(P, M, reg, num_iter_max=NUM_ITER_MAX, stop_thr=STOP_THR, tau=TAU,
                  check_period=CHECK_PERIOD, precision="auto", f32_floor_ulps=0.0,
                  row_begin=0, row_end=None, row_step=1, return_info=False):
    """Entropic OT cost <Gamma, M> for ordered pairs (selected rows x all N columns).

    Device replacement of the loop at pilotpy/tools/Trajectory.py:512-515; each pair follows
    ``ot.sinkhorn2(P[i], P[j], M, reg, method="sinkhorn_stabilized")``.

    P : (N, K) proportion vectors; M : (K, K) cost (already divided by its max).
    Returns float64 (n_rows, N) and, with ``return_info``, a dict of iters / err / flags arrays.
    """
    P = _as_f64(P, "P")
    M = _as_f64(M, "M")
    if P.ndim != 2 or M.ndim != 2 or M.shape[0] != M.shape[1] or M.shape[0] != P.shape[1]:
        raise ValueError("shape mismatch: P %s, M %s" % (P.shape, M.shape))
    N, K = P.shape
    if precision not in _lib.PREC:
        raise ValueError("precision must be one of %s" % sorted(_lib.PREC))
    row_end = N if row_end is None else int(row_end)
    n_rows = n_rows_of(N, row_begin, row_end, row_step)
    emd = np.empty((n_rows, N), dtype=np.float64)
    L = _lib.load()
    sym = int(np.array_equal(M, M.T))
    if return_info:
        iters = np.empty((n_rows, N), dtype=np.int32)
        err = np.empty((n_rows, N), dtype=np.float64)
        flags = np.empty((n_rows, N), dtype=np.int32)
        pi, pe, pf = _lib.iptr(iters), _lib.dptr(err), _lib.iptr(flags)
    else:                                  # only the matrix travels back
        pi = pe = pf = None
    _lib.check(L.pilot_ot_sinkhorn_grid(
        _lib.dptr(P), N, K, _lib.dptr(M), float(reg), int(num_iter_max), float(stop_thr), float(tau),
        int(check_period), _lib.PREC[precision], float(f32_floor_ulps), sym,
        int(row_begin), row_end, int(row_step), _lib.dptr(emd), pi, pe, pf))
    if return_info:
        return emd, dict(iters=iters, err=err, flags=flags)
    return emd


def pdist_square(centroids, metric="cosine"):
    """Square pairwise-distance matrix of the K centroids (device replacement of
    ``squareform(pdist(centroids, metric))``, pilotpy/tools/Trajectory.py:468-469)."""
    X = _as_f64(centroids, "centroids")
    if X.ndim != 2:
        raise ValueError("centroids must be 2-D (K, D)")
    if metric not in _lib.METRICS:
        raise NotImplementedError("metric %r: the device kernel implements scipy's pdist names %s" % (metric, sorted(_lib.METRICS)))
    K, D = X.shape
    out = np.zeros((K, K), dtype=np.float64)
    aux = None
    if metric == "mahalanobis":
        # scipy's own preparation (scipy/spatial/distance.py::_validate_mahalanobis_kwargs): VI = inv(cov(X^T))^T on the host
        if K <= D:
            raise ValueError("The number of observations (%d) is too small; the covariance matrix is singular. For observations "
                             "with %d dimensions, at least %d observations are required." % (K, D, D + 1))
        aux = np.ascontiguousarray(np.linalg.inv(np.atleast_2d(np.cov(X.T))).T, dtype=np.float64)
    _lib.check(_lib.load().pilot_ot_cost_matrix_ex(_lib.dptr(X), K, D, _lib.METRICS[metric], _lib.dptr(aux) if aux is not None else None,
                                                   _lib.dptr(out)))
    return out


def proportions(cell_code, sample_code, n_samples, n_types, regulizer=0.2, normalization=True, n_total=None):
    """N x K smoothed cell-type proportions from per-cell integer codes (device histogram; replaces the pandas
    loops of Cluster_Representations, pilotpy/tools/Trajectory.py:400-430).  Bit-identical to the reference."""
    cc = np.ascontiguousarray(cell_code, dtype=np.int32)
    sc = np.ascontiguousarray(sample_code, dtype=np.int32)
    if cc.shape != sc.shape or cc.ndim != 1:
        raise ValueError("cell_code and sample_code must be 1-D arrays of equal length")
    n_total = cc.size if n_total is None else int(n_total)
    P = np.zeros((n_samples, n_types), dtype=np.float64)
    _lib.check(_lib.load().pilot_ot_proportions(_lib.iptr(cc), _lib.iptr(sc), cc.size, n_total, int(n_samples),
                                                int(n_types), float(regulizer), int(bool(normalization)), _lib.dptr(P)))
    return P


def proportions_and_first_rows(cell_code, sample_code, n_samples, n_types, regulizer=0.2, normalization=True, n_total=None):
    """:func:`proportions` plus, from the same pass over the codes, the first row of every sample (int64, -1: none) --
    the row ``return_real_labels`` reads (pilotpy/tools/Trajectory.py:617-642)."""
    cc = np.ascontiguousarray(cell_code, dtype=np.int32)
    sc = np.ascontiguousarray(sample_code, dtype=np.int32)
    if cc.shape != sc.shape or cc.ndim != 1:
        raise ValueError("cell_code and sample_code must be 1-D arrays of equal length")
    n_total = cc.size if n_total is None else int(n_total)
    P = np.zeros((n_samples, n_types), dtype=np.float64)
    first = np.full(n_samples, -1, dtype=np.int64)
    _lib.check(_lib.load().pilot_ot_proportions_ex(
        _lib.iptr(cc), _lib.iptr(sc), cc.size, n_total, int(n_samples), int(n_types), float(regulizer), int(bool(normalization)),
        _lib.dptr(P), first.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))))
    return P, first


def label_codes(ids, max_uniques=1 << 16, n_threads=None):
    """``(codes int32, first_rows int64)``: the labels of one column numbered in order of first appearance, -1 = missing
    (host pass of ``libpilot_ot.so``, ``pilot_ot_label_codes``: what ``Series.unique()`` and the per-label masks of
    pilotpy/tools/Trajectory.py:402-425 amount to).  ``ids``: 1-D contiguous int8/int16/int32 (negative = missing: the codes
    of a pandas Categorical) or uint64 (opaque identities, 0 = missing).  ``None`` when the column holds more than
    ``max_uniques`` distinct labels (the caller then takes another route)."""
    ids = np.ascontiguousarray(ids)
    if ids.ndim != 1 or ids.dtype not in (np.int8, np.int16, np.int32, np.uint64):
        raise ValueError("ids must be a 1-D int8 / int16 / int32 / uint64 array")
    if n_threads is None:
        n_threads = max(1, min(4, (os.cpu_count() or 2) // 2))
    codes = np.empty(ids.size, dtype=np.int32)
    first = np.empty(int(max_uniques), dtype=np.int64)
    n_u = ctypes.c_int(0)
    rc = _lib.load().pilot_ot_label_codes(ctypes.c_void_p(ids.ctypes.data), ids.itemsize, ids.size, int(max_uniques), int(n_threads),
                                          _lib.iptr(codes), first.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), ctypes.byref(n_u))
    if rc == _lib.ENOTSUP:
        return None
    _lib.check(rc)
    return codes, first[:n_u.value]


class EmbeddingUpload:
    """The C x D embedding on its way to the device: the copy runs on a helper thread (ctypes releases the GIL) while the
    caller factorises the label columns; :meth:`medians` joins it and runs the device radix select.

    A SMALL embedding (below ``SMALL_BYTES``: the reference test's own cohort is 1.3 MB) is not worth a thread, a device
    allocation and the ``hipFree`` that synchronises the device at the end -- 0.4 ms of a 2.8 ms call: it goes up inside
    :meth:`medians`, through the calling thread's buffer pool (``pilot_ot_centroid_medians`` on the host array)."""

    SMALL_BYTES = 4 << 20

    def __init__(self, X):
        import threading
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("X must be 2-D (cells, dims)")
        if X.dtype == np.float32:
            self.dt = 0
        else:
            X = X.astype(np.float64, copy=False)
            self.dt = 1
        self.X = np.ascontiguousarray(X)
        self.L = _lib.load()
        self.h = ctypes.c_void_p()
        self.err = None
        dev = ctypes.c_int(0)
        _lib.check(self.L.pilot_ot_get_device(ctypes.byref(dev)))
        self.device = dev.value
        self.thread = None
        if self.X.nbytes < self.SMALL_BYTES:
            return

        def work():
            try:
                _lib.check(self.L.pilot_ot_set_device(self.device))      # (a new host thread starts on device 0)
                _lib.check(self.L.pilot_ot_embedding_upload(ctypes.c_void_p(self.X.ctypes.data), self.dt, self.X.shape[0],
                                                            self.X.shape[1], ctypes.byref(self.h)))
            except BaseException as e:      # re-raised by the caller's thread in medians()
                self.err = e

        self.thread = threading.Thread(target=work, name="pilot_ot_embedding_upload")
        self.thread.start()

    def medians(self, cell_code, n_types):
        if self.thread is None:
            return centroid_medians(self.X, cell_code, n_types)
        self.thread.join()
        if self.err is not None:
            raise self.err
        cc = np.ascontiguousarray(cell_code, dtype=np.int32)
        if cc.shape != (self.X.shape[0],):
            raise ValueError("cell_code must have one entry per row of X")
        out = np.zeros((int(n_types), self.X.shape[1]), dtype=np.float64)
        _lib.check(self.L.pilot_ot_centroid_medians_dev(self.h, _lib.iptr(cc), int(n_types), _lib.dptr(out)))
        return out

    def prepass(self, cell_code, sample_code, n_samples, n_types, regulizer=0.2, normalization=True, n_total=None):
        """``(P, first_rows, centroids)`` -- :func:`proportions_and_first_rows` and :meth:`medians` from ONE upload of the two
        code columns (``pilot_ot_prepass_dev``): the same bits as the separate calls, a third of their transfers."""
        if self.thread is None:
            P, first = proportions_and_first_rows(cell_code, sample_code, n_samples, n_types, regulizer=regulizer,
                                                  normalization=normalization, n_total=n_total)
            return P, first, centroid_medians(self.X, cell_code, n_types)
        self.thread.join()
        if self.err is not None:
            raise self.err
        cc = np.ascontiguousarray(cell_code, dtype=np.int32)
        sc = np.ascontiguousarray(sample_code, dtype=np.int32)
        if cc.shape != (self.X.shape[0],) or sc.shape != cc.shape:
            raise ValueError("cell_code and sample_code must have one entry per row of X")
        n_total = cc.size if n_total is None else int(n_total)
        P = np.zeros((int(n_samples), int(n_types)), dtype=np.float64)
        first = np.full(int(n_samples), -1, dtype=np.int64)
        cen = np.zeros((int(n_types), self.X.shape[1]), dtype=np.float64)
        _lib.check(self.L.pilot_ot_prepass_dev(self.h, _lib.iptr(cc), _lib.iptr(sc), n_total, int(n_samples), int(n_types),
                                               float(regulizer), int(bool(normalization)), _lib.dptr(P),
                                               first.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), _lib.dptr(cen)))
        return P, first, cen

    def close(self):
        if self.thread is not None:
            self.thread.join()
        if self.h:
            self.L.pilot_ot_embedding_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def centroid_medians(X, cell_code, n_types):
    """K x D per-cell-type column-wise medians of the C x D embedding (device radix select; replaces
    ``data[annot.cell_type == k].median(axis=0)``, pilotpy/tools/Trajectory.py:465-466).  float32 / float64 input is
    processed in its own dtype, like pandas."""
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError("X must be 2-D (cells, dims)")
    if X.dtype == np.float32:
        dt = 0
    else:
        X = X.astype(np.float64, copy=False)
        dt = 1
    X = np.ascontiguousarray(X)
    cc = np.ascontiguousarray(cell_code, dtype=np.int32)
    if cc.shape != (X.shape[0],):
        raise ValueError("cell_code must have one entry per row of X")
    out = np.zeros((int(n_types), X.shape[1]), dtype=np.float64)
    _lib.check(_lib.load().pilot_ot_centroid_medians(ctypes.c_void_p(X.ctypes.data), dt, X.shape[0], X.shape[1],
                                                     _lib.iptr(cc), int(n_types), _lib.dptr(out)))
    return out


def _cell_inputs(X, offsets):
    X = np.ascontiguousarray(X, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if X.ndim != 2 or offsets.ndim != 1 or offsets.size < 2 or offsets[0] != 0 or offsets[-1] != X.shape[0]:
        raise ValueError("X must be (C, D) and offsets (N + 1,) with offsets[0] = 0, offsets[-1] = C")
    if not np.all(np.isfinite(X)):
        raise ValueError("X contains NaN or inf")
    return X, offsets


class CellCohort:
    """Device-resident cell clouds for the cell-level W2 extension (``pilot_ot_cell_cohort_*``): the cells stay in HBM as
    fp16 / bf16 operand pieces across calls, a call moves only result rows."""

    def __init__(self, X, offsets):
        X, offsets = _cell_inputs(X, offsets)
        self.N, self.D = offsets.size - 1, X.shape[1]
        self.cells_per_patient = np.diff(offsets)
        self.L = _lib.load()
        self.h = ctypes.c_void_p()
        _lib.check(self.L.pilot_ot_cell_cohort_create(ctypes.c_void_p(X.ctypes.data), ctypes.c_void_p(offsets.ctypes.data),
                                                      self.N, self.D, ctypes.byref(self.h)))
        self.last_kernel_ms = None

    def w2_grid(self, scale, reg, num_iter_max=1000, stop_thr=1e-9, check_period=10, f32_floor_ulps=0.0,
                row_begin=0, row_end=None, row_step=1, return_info=False):
        row_end = self.N if row_end is None else int(row_end)
        n_rows = n_rows_of(self.N, row_begin, row_end, row_step)
        w2 = np.zeros((n_rows, self.N), dtype=np.float64)
        iters = np.zeros((n_rows, self.N), dtype=np.int32)
        err = np.zeros((n_rows, self.N), dtype=np.float64)
        ms = ctypes.c_float(0.0)
        _lib.check(self.L.pilot_ot_cell_w2_grid_cohort(self.h, float(scale), float(reg), int(num_iter_max), float(stop_thr),
                                                       int(check_period), float(f32_floor_ulps), int(row_begin), row_end,
                                                       int(row_step), _lib.dptr(w2), _lib.iptr(iters), _lib.dptr(err),
                                                       ctypes.byref(ms)))
        self.last_kernel_ms = float(ms.value)
        pieces = ctypes.c_int(0)
        _lib.check(self.L.pilot_ot_cell_cohort_pieces(self.h, ctypes.byref(pieces)))
        self.last_pieces = pieces.value           # 2: fp16 operand pieces (3 piece products per tile), 3: bf16 (6)
        if return_info:
            return w2, dict(iters=iters, err=err)
        return w2

    def close(self):
        if self.h:
            self.L.pilot_ot_cell_cohort_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cell_w2_grid(X, offsets, scale, reg, num_iter_max=1000, stop_thr=1e-9, check_period=10, f32_floor_ulps=0.0,
                 row_begin=0, row_end=None, row_step=1, return_info=False, devices=None):
    """EXTENSION (not in the reference; BASELINE config 5): entropic W2 cost between patients' raw cell clouds.

    X: (C, D) float32 embedding with every patient's cells contiguous; offsets: (N + 1,) row ranges.  Pair (i, j):
    uniform weights, cost |x - y|^2 / scale, log-domain Sinkhorn with POT ``sinkhorn_log`` control flow; returns the
    (n_rows, N) matrix of <Gamma, C>.  ``devices=[...]``: the full grid with its rows dealt round-robin over several GPUs."""
    X, offsets = _cell_inputs(X, offsets)
    N = offsets.size - 1
    if devices is not None:
        if row_begin != 0 or row_end not in (None, N) or row_step != 1:
            raise ValueError("devices=[...] computes the full grid")
        dev = np.ascontiguousarray([int(d) for d in devices], dtype=np.int32)
        w2 = np.zeros((N, N), dtype=np.float64)
        iters = np.zeros((N, N), dtype=np.int32)
        err = np.zeros((N, N), dtype=np.float64)
        _lib.check(_lib.load().pilot_ot_cell_w2_grid_multi(
            ctypes.c_void_p(X.ctypes.data), ctypes.c_void_p(offsets.ctypes.data), N, X.shape[1], float(scale), float(reg),
            int(num_iter_max), float(stop_thr), int(check_period), float(f32_floor_ulps), _lib.iptr(dev), len(dev),
            _lib.dptr(w2), _lib.iptr(iters), _lib.dptr(err)))
        return (w2, dict(iters=iters, err=err)) if return_info else w2
    co = CellCohort(X, offsets)
    try:
        return co.w2_grid(scale, reg, num_iter_max=num_iter_max, stop_thr=stop_thr, check_period=check_period,
                          f32_floor_ulps=f32_floor_ulps, row_begin=row_begin, row_end=row_end, row_step=row_step,
                          return_info=return_info)
    finally:
        co.close()


class DevicePlan:
    """Device-resident pair-grid problem: P, M and the outputs live in HBM across calls.

    Used by ``bench.py`` (inputs resident before the timed region) and by the multi-GPU driver.
    Device memory comes from the library's own allocator (no torch needed); pointers can equally
    be torch ``data_ptr()`` values when the caller wants torch to own the buffers.
    """

    def __init__(self, P, M, n_rows_max=None):
        P = _as_f64(P, "P")
        M = _as_f64(M, "M")
        self.N, self.K = P.shape
        self.sym = int(np.array_equal(M, M.T))
        self.L = _lib.load()
        self._bufs = []
        self.plan = ctypes.c_void_p()
        _lib.check(self.L.pilot_ot_plan_create(self.N, self.K, ctypes.byref(self.plan)))
        # the device entry point cannot see max(M): the plan carries it, and every range decision (what AUTO means, the
        # fp16-split domain, the hand-over thresholds) is taken on max(M) / reg -- M need not be normalised
        self.max_cost = float(M.max()) if M.size else 1.0
        if not (self.max_cost > 0.0 and np.isfinite(self.max_cost)):
            self.max_cost = 1.0            # (an all-zero / NaN cost: every range decision as for the normalised cost, never a stale value)
        _lib.check(self.L.pilot_ot_plan_set_max_cost(self.plan, self.max_cost))
        self.n_rows_max = self.N if n_rows_max is None else n_rows_max
        n_out = self.n_rows_max * self.N
        self.dP = self._alloc(P.nbytes)
        self.dM = self._alloc(M.nbytes)
        self.dE = self._alloc(8 * n_out)
        self.dErr = self._alloc(8 * n_out)
        self.dIt = self._alloc(4 * n_out)
        self.dFl = self._alloc(4 * n_out)
        _lib.check(self.L.pilot_ot_memcpy_h2d(self.dP, P.ctypes.data, P.nbytes))
        _lib.check(self.L.pilot_ot_memcpy_h2d(self.dM, M.ctypes.data, M.nbytes))

    def _alloc(self, nbytes):
        p = ctypes.c_void_p()
        _lib.check(self.L.pilot_ot_dev_alloc(ctypes.byref(p), int(nbytes)))
        self._bufs.append(p)
        return p

    def run(self, reg, row_begin=0, row_end=None, row_step=1, precision="auto", num_iter_max=NUM_ITER_MAX,
            stop_thr=STOP_THR, tau=TAU, check_period=CHECK_PERIOD, f32_floor_ulps=0.0, stream=None,
            d_emd=None):
        """Enqueue one pass over the selected rows (asynchronous)."""
        row_end = self.N if row_end is None else row_end
        if n_rows_of(self.N, row_begin, row_end, row_step) > self.n_rows_max:
            raise ValueError("row selection exceeds the plan's n_rows_max")
        _lib.check(self.L.pilot_ot_sinkhorn_grid_dev(
            self.plan, self.dP, self.dM, float(reg), int(num_iter_max), float(stop_thr), float(tau),
            int(check_period), _lib.PREC[precision], float(f32_floor_ulps), self.sym,
            int(row_begin), int(row_end), int(row_step),
            self.dE if d_emd is None else ctypes.c_void_p(d_emd), self.dIt, self.dErr, self.dFl,
            ctypes.c_void_p(stream) if stream else None))

    def enable_timing(self, enable=True):
        """HIP events around the pair-grid kernels of every call (``True``) or of every n-th call (``enable=n``)."""
        _lib.check(self.L.pilot_ot_plan_enable_timing(self.plan, int(enable)))

    def enable_graph(self, enable=True):
        """Replay repeated identical `run` calls as one hipGraph launch (pilot_ot_plan_enable_graph)."""
        _lib.check(self.L.pilot_ot_plan_enable_graph(self.plan, int(enable)))

    def kernel_times_ms(self, max_n=64):
        """(main_ms, track_ms) float arrays of the most recent timed calls (sync the stream first)."""
        a = (ctypes.c_float * max_n)()
        b = (ctypes.c_float * max_n)()
        n = ctypes.c_int(0)
        _lib.check(self.L.pilot_ot_plan_kernel_times(self.plan, max_n, a, b, ctypes.byref(n)))
        return np.array(a[:n.value]), np.array(b[:n.value])

    def sync(self, stream=None):
        _lib.check(self.L.pilot_ot_stream_sync(ctypes.c_void_p(stream) if stream else None))

    def device_matrix(self):
        """The N x N result of the last full-grid run as it sits in HBM (sync first), for the device-side consumers."""
        if self.n_rows_max != self.N:
            raise ValueError("the plan holds a row shard, not the full matrix")
        return DeviceMatrix(self.dE, self.N, owner=self)

    def fetch(self, n_rows=None):
        n_rows = self.n_rows_max if n_rows is None else n_rows
        n = n_rows * self.N
        emd = np.empty((n_rows, self.N), dtype=np.float64)
        iters = np.empty((n_rows, self.N), dtype=np.int32)
        err = np.empty((n_rows, self.N), dtype=np.float64)
        flags = np.empty((n_rows, self.N), dtype=np.int32)
        _lib.check(self.L.pilot_ot_memcpy_d2h(emd.ctypes.data, self.dE, 8 * n))
        _lib.check(self.L.pilot_ot_memcpy_d2h(iters.ctypes.data, self.dIt, 4 * n))
        _lib.check(self.L.pilot_ot_memcpy_d2h(err.ctypes.data, self.dErr, 8 * n))
        _lib.check(self.L.pilot_ot_memcpy_d2h(flags.ctypes.data, self.dFl, 4 * n))
        return emd, dict(iters=iters, err=err, flags=flags)

    def close(self):
        for p in self._bufs:
            self.L.pilot_ot_dev_free(p)
        self._bufs = []
        if self.plan:
            self.L.pilot_ot_plan_destroy(self.plan)
            self.plan = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def equal_masses(P, rtol=1e-12):
    """ot.emd2 rescales b to the mass of a, so emd2(a, b) = sum(a) * W(a / sum a, b / sum b): the matrix of a symmetric cost
    is symmetric only when every histogram carries the same mass (to rounding)."""
    s = np.asarray(P, dtype=np.float64).sum(1)
    return s.size == 0 or float(s.max() - s.min()) <= rtol * max(float(np.abs(s).max()), 1e-300)


def emd_grid(P, M, row_begin=0, row_end=None, row_step=1, mode="auto", return_info=False):
    """Exact OT cost for ordered pairs (device replacement of the ``ot.emd2`` loop,
    pilotpy/tools/Trajectory.py:507-511).

    mode: "all" solves every (row, column) pair; "upper" only column >= row (rest left 0);
    "mirror" = upper + device-side mirroring (full square grid only); "auto" picks "mirror"
    when M is exactly symmetric, all histograms carry the same mass (see equal_masses) and the full grid is requested,
    else "all".
    """
    P = _as_f64(P, "P")
    M = _as_f64(M, "M")
    if P.ndim != 2 or M.ndim != 2 or M.shape[0] != M.shape[1] or M.shape[0] != P.shape[1]:
        raise ValueError("shape mismatch: P %s, M %s" % (P.shape, M.shape))
    N, K = P.shape
    row_end = N if row_end is None else int(row_end)
    full = (row_begin == 0 and row_end == N and row_step == 1)
    if mode == "auto":
        mode = "mirror" if (full and np.array_equal(M, M.T) and equal_masses(P)) else "all"
    modes = {"all": _lib.EMD_ALL, "upper": _lib.EMD_UPPER, "mirror": _lib.EMD_MIRROR}
    if mode not in modes:
        raise ValueError("mode must be one of %s" % sorted(modes))
    n_rows = n_rows_of(N, row_begin, row_end, row_step)
    emd = np.zeros((n_rows, N), dtype=np.float64)
    n_aug = np.zeros((n_rows, N), dtype=np.int32)
    _lib.check(_lib.load().pilot_ot_emd_grid(_lib.dptr(P), N, K, _lib.dptr(M), modes[mode], int(row_begin), row_end,
                                             int(row_step), _lib.dptr(emd), _lib.iptr(n_aug)))
    if (n_aug < 0).any():
        raise _lib.PilotOTError("exact-EMD kernel: augmentation guard tripped on %d pairs" % int((n_aug < 0).sum()))
    if return_info:
        return emd, dict(n_aug=n_aug)
    return emd


# ---- consumers of the finished matrix ------------------------------------------------------------
def row_distances(E, metric="euclidean", normalize_by_max=False):
    """Distances between the ROWS of the N x N matrix E (of E / E.max() with ``normalize_by_max``) on the device: the points
    pilotpy's diffusion map (pilotpy/plot/ploting.py:95-110) and silhouette scores (pilotpy/tools/Trajectory.py:592-612)
    work with.  metric: "euclidean" (scipy cdist) or "cosine" (sklearn cosine_distances)."""
    E = _as_f64(E, "E")
    if E.ndim != 2 or E.shape[0] != E.shape[1]:
        raise ValueError("E must be square, got %s" % (E.shape,))
    if metric not in _lib.ROW_METRICS:
        raise NotImplementedError("row metric %r: the device kernel implements %s" % (metric, sorted(_lib.ROW_METRICS)))
    D = np.empty_like(E)
    _lib.check(_lib.load().pilot_ot_row_distances(_lib.dptr(E), E.shape[0], int(bool(normalize_by_max)),
                                                  _lib.ROW_METRICS[metric], _lib.dptr(D)))
    return D


def silhouette_precomputed(D, labels, return_samples=False):
    """``sklearn.metrics.silhouette_score(D, labels, metric="precomputed")`` on the device (labels of any hashable type)."""
    D = _as_f64(D, "D")
    if D.ndim != 2 or D.shape[0] != D.shape[1]:
        raise ValueError("D must be square, got %s" % (D.shape,))
    uniq, codes = np.unique(np.asarray(labels), return_inverse=True)
    if codes.shape != (D.shape[0],):
        raise ValueError("one label per sample expected")
    codes = np.ascontiguousarray(codes, dtype=np.int32)
    score = ctypes.c_double(0.0)
    samples = np.empty(D.shape[0], dtype=np.float64)
    _lib.check(_lib.load().pilot_ot_silhouette(_lib.dptr(D), _lib.iptr(codes), D.shape[0], len(uniq), ctypes.byref(score),
                                               _lib.dptr(samples)))
    return (score.value, samples) if return_samples else score.value


def _label_codes(labels, n):
    uniq, codes = np.unique(np.asarray(labels), return_inverse=True)
    if codes.shape != (n,):
        raise ValueError("one label per sample expected")
    return np.ascontiguousarray(codes, dtype=np.int32), len(uniq)


def _matrix_arg(E):
    """(pointer, is_device, N) of a square matrix given as a numpy array or as a device-resident result (DeviceMatrix)."""
    if isinstance(E, DeviceMatrix):
        return ctypes.c_void_p(E.ptr), 1, E.N, None
    E = _as_f64(E, "E")
    if E.ndim != 2 or E.shape[0] != E.shape[1]:
        raise ValueError("E must be square, got %s" % (E.shape,))
    return ctypes.c_void_p(E.ctypes.data), 0, E.shape[0], E


class DeviceMatrix:
    """An N x N fp64 matrix that lives in HBM (e.g. ``DevicePlan.device_matrix()`` after a full-grid run, or
    ``multi.MultiPlan.device_matrix()``): the consumers below take it without a trip through host memory."""

    def __init__(self, ptr, N, owner=None):
        self.ptr = int(ptr.value if isinstance(ptr, ctypes.c_void_p) else ptr)
        self.N = int(N)
        self.owner = owner            # keeps the allocation alive


def silhouette_of_rows(E, labels, metric="cosine", normalize_by_max=False, return_samples=False):
    """``sklearn.metrics.silhouette_score(E, labels, metric=metric)`` with the ROWS of E as the points -- what
    ``Sil_computing`` does (pilotpy/tools/Trajectory.py:592-612): row distances and silhouette chained on the device, only the
    N per-sample scores come back.  E: numpy array or :class:`DeviceMatrix`."""
    ptr, on_dev, N, keep = _matrix_arg(E)
    if metric not in _lib.ROW_METRICS:
        raise NotImplementedError("row metric %r: the device kernel implements %s" % (metric, sorted(_lib.ROW_METRICS)))
    codes, n_clusters = _label_codes(labels, N)
    score = ctypes.c_double(0.0)
    samples = np.empty(N, dtype=np.float64)
    _lib.check(_lib.load().pilot_ot_silhouette_of_rows(ptr, on_dev, N, int(bool(normalize_by_max)), _lib.ROW_METRICS[metric],
                                                       _lib.iptr(codes), n_clusters, ctypes.byref(score), _lib.dptr(samples)))
    return (score.value, samples) if return_samples else score.value


def diffusion_kernel_of_rows(E, k=64, epsilon=1.0, return_distances=True):
    """The dense part of ``pl.trajectory`` (pilotpy/plot/ploting.py:95-110) chained on the device: E / E.max() -> Euclidean row
    distances -> pydiffmap's k-nearest-neighbour Gaussian kernel.  Returns ``(D, Kmat)`` (``D`` None unless asked for)."""
    ptr, on_dev, N, keep = _matrix_arg(E)
    D = np.empty((N, N), dtype=np.float64) if return_distances else None
    Kmat = np.empty((N, N), dtype=np.float64)
    _lib.check(_lib.load().pilot_ot_diffusion_kernel_of_rows(ptr, on_dev, N, int(k), float(epsilon),
                                                             _lib.dptr(D) if D is not None else None, _lib.dptr(Kmat)))
    return D, Kmat


def knn_gaussian_kernel(D, k=64, epsilon=1.0):
    """Kernel matrix of pydiffmap's ``DiffusionMap.from_sklearn(epsilon=, k=)`` from row distances D: exp(-d^2 / (4 eps)) on
    every row's k nearest rows (itself included), 0 elsewhere (pilotpy/plot/ploting.py:109-110)."""
    D = _as_f64(D, "D")
    if D.ndim != 2 or D.shape[0] != D.shape[1]:
        raise ValueError("D must be square, got %s" % (D.shape,))
    Kmat = np.empty_like(D)
    _lib.check(_lib.load().pilot_ot_knn_kernel(_lib.dptr(D), D.shape[0], int(k), float(epsilon), _lib.dptr(Kmat)))
    return Kmat
