"""CPU oracle for the pilot_amd hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of ``oracle/pilot_oracle.c`` plus literal numpy/pandas restatements
of the host-side steps of ``pilotpy.tl.wasserstein_distance``:

=========================  ====================================================
oracle function            reference it restates (``/root/reference/...``)
=========================  ====================================================
cluster_representations    pilotpy/tools/Trajectory.py:377-436
cost_matrix                pilotpy/tools/Trajectory.py:441-475
return_real_labels         pilotpy/tools/Trajectory.py:617-642
wasserstein_d              pilotpy/tools/Trajectory.py:479-523 (loop + layout)
sinkhorn2 / sinkhorn_grid  POT 0.9.x ot.sinkhorn2(method="sinkhorn_stabilized")
emd2 / emd_grid            POT 0.9.x ot.emd2 (value of the exact LP optimum)
=========================  ====================================================

PARITY UNPINNED for the two POT rows: POT (``pot>=0.9.1,<0.10.0``,
``/root/reference/setup.py:19``) is a third-party dependency that is neither in
the reference tree nor installable here, and the reference's only test
(``test/test_pilot.py:26-28,41``) asserts shapes, not values.  See the header
of ``pilot_oracle.c`` for what pins the restatement instead.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; ``pilot_amd`` never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpilot_oracle.so")
_lib = None

# POT defaults (ot.sinkhorn2 / sinkhorn_stabilized signature, POT 0.9.x)
NUM_ITER_MAX = 1000
STOP_THR = 1e-9
TAU = 1e3
PRINT_PERIOD = 20

FLAG_CONVERGED = 1
FLAG_NAN_REVERT = 2
FLAG_ABSORB_ON_LAST = 4
FLAG_ABSORBED = 8


def build(force: bool = False) -> str:
    """Compile pilot_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pilot_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "libpilot_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        L.pilot_oracle_sinkhorn2_stabilized.restype = ctypes.c_double
        L.pilot_oracle_sinkhorn2_stabilized.argtypes = [
            dp, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ip, dp]
        L.pilot_oracle_sinkhorn_grid.restype = ctypes.c_int
        L.pilot_oracle_sinkhorn_grid.argtypes = [
            dp, ctypes.c_int, ctypes.c_int, dp, ctypes.c_double, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp, ip]
        L.pilot_oracle_sinkhorn_grid_ex.restype = ctypes.c_int
        L.pilot_oracle_sinkhorn_grid_ex.argtypes = [
            dp, ctypes.c_int, ctypes.c_int, dp, ctypes.c_double, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp, ip]
        L.pilot_oracle_emd2.restype = ctypes.c_double
        L.pilot_oracle_emd2.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_int, dp]
        L.pilot_oracle_emd_grid.restype = ctypes.c_int
        L.pilot_oracle_emd_grid.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
        L.pilot_oracle_cell_w2.restype = ctypes.c_double
        L.pilot_oracle_cell_w2.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ip, dp]
        L.pilot_oracle_emd_grid_fast.restype = ctypes.c_int
        L.pilot_oracle_emd_grid_fast.argtypes = L.pilot_oracle_emd_grid.argtypes
        L.pilot_oracle_emd_grid_ns.restype = ctypes.c_int
        L.pilot_oracle_emd_grid_ns.argtypes = L.pilot_oracle_emd_grid.argtypes
        L.pilot_oracle_emd2_ns.restype = ctypes.c_double
        L.pilot_oracle_emd2_ns.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ip]
        _lib = L
    return _lib


def _dptr(x):
    return x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _iptr(x):
    return x.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


# --------------------------------------------------------------------------- POT
def sinkhorn2(a, b, M, reg, numItermax=NUM_ITER_MAX, stopThr=STOP_THR, tau=TAU,
              print_period=PRINT_PERIOD, legacy_loop=False, return_info=False):
    """ot.sinkhorn2(a, b, M, reg, method="sinkhorn_stabilized") -- Trajectory.py:515."""
    a, b, M = _f64(a), _f64(b), _f64(M)
    info = np.zeros(4, dtype=np.int32)
    err = ctypes.c_double(0.0)
    val = lib().pilot_oracle_sinkhorn2_stabilized(
        _dptr(a), _dptr(b), _dptr(M), a.size, b.size, float(reg), int(numItermax), float(tau),
        float(stopThr), int(print_period), int(bool(legacy_loop)), _iptr(info), ctypes.byref(err))
    if return_info:
        return val, dict(iters=int(info[0]), n_absorb=int(info[1]), last_absorb=int(info[2]),
                         flags=int(info[3]), err=err.value)
    return val


def _rows(N, row_begin, row_end, row_step):
    row_end = N if row_end is None else row_end
    return row_end, len(range(row_begin, row_end, row_step))


def sinkhorn_grid(P, M, reg, numItermax=NUM_ITER_MAX, stopThr=STOP_THR, tau=TAU,
                  print_period=PRINT_PERIOD, legacy_loop=False, row_begin=0, row_end=None,
                  row_step=1, n_threads=1, return_info=False, stop_floor_ulps=0.0):
    """All ordered pairs (selected rows x all columns) -- Trajectory.py:512-515.

    ``stop_floor_ulps`` > 0 is NOT POT: stopThr floored per pair at that many f32 ulps of ||b||_2, the f32 GPU kernels'
    stopping rule, so that bench.py can time the CPU on the update counts the GPU ran (``cpu_baseline_equal_updates``)."""
    P, M = _f64(P), _f64(M)
    N, K = P.shape
    row_end, nrows = _rows(N, row_begin, row_end, row_step)
    emd = np.zeros((nrows, N))
    iters = np.zeros((nrows, N), dtype=np.int32)
    err = np.zeros((nrows, N))
    flags = np.zeros((nrows, N), dtype=np.int32)
    rc = lib().pilot_oracle_sinkhorn_grid_ex(
        _dptr(P), N, K, _dptr(M), float(reg), int(numItermax), float(tau), float(stopThr),
        int(print_period), int(bool(legacy_loop)), float(stop_floor_ulps), row_begin, row_end, row_step, int(n_threads),
        _dptr(emd), _iptr(iters), _dptr(err), _iptr(flags))
    if rc != 0:
        raise ValueError("pilot_oracle_sinkhorn_grid: bad arguments")
    if return_info:
        return emd, dict(iters=iters, err=err, flags=flags)
    return emd


def emd2(a, b, M, return_plan=False):
    """ot.emd2(a, b, M) -- Trajectory.py:511."""
    a, b, M = _f64(a), _f64(b), _f64(M)
    G = np.zeros((a.size, b.size))
    val = lib().pilot_oracle_emd2(_dptr(a), _dptr(b), _dptr(M), a.size, b.size, _dptr(G))
    return (val, G) if return_plan else val


def emd2_ns(a, b, M, return_pivots=False):
    """ot.emd2(a, b, M) by the network simplex leg (pilot_oracle.c::pilot_oracle_emd2_ns) -- same LP value."""
    a, b, M = _f64(a), _f64(b), _f64(M)
    piv = ctypes.c_int(0)
    val = lib().pilot_oracle_emd2_ns(_dptr(a), _dptr(b), _dptr(M), a.size, b.size, None, ctypes.byref(piv))
    return (val, piv.value) if return_pivots else val


def emd_grid(P, M, row_begin=0, row_end=None, row_step=1, n_threads=1, fast=False):
    """All ordered pairs, exact OT -- Trajectory.py:507-511.  ``fast``: the quicker successive-shortest-path solver
    (the HIP kernel's algorithm on one CPU thread), ``fast="ns"``: the network simplex leg (the algorithm family of POT's
    own solver; the CPU baseline of ``bench.py --mode emd``) -- same LP value."""
    P, M = _f64(P), _f64(M)
    N, K = P.shape
    row_end, nrows = _rows(N, row_begin, row_end, row_step)
    emd = np.zeros((nrows, N))
    fn = lib().pilot_oracle_emd_grid_ns if fast == "ns" else (lib().pilot_oracle_emd_grid_fast if fast else lib().pilot_oracle_emd_grid)
    rc = fn(_dptr(P), N, K, _dptr(M), row_begin, row_end, row_step,
                                     int(n_threads), _dptr(emd))
    if rc != 0:
        raise ValueError("pilot_oracle_emd_grid: bad arguments")
    return emd


def sinkhorn_log_converged(a, b, M, reg, tol=1e-14, max_iter=200000):
    """Independent log-domain Sinkhorn run to its fixed point (NOT POT's stopping rule).

    The entropic optimum is unique, so on pairs where POT's rule converges the
    stabilized restatement must agree with this to ~stopThr.  Returns <Gamma, M>.
    """
    from scipy.special import logsumexp
    a, b, M = _f64(a), _f64(b), _f64(M)
    la, lb = np.log(a), np.log(b)
    f = np.zeros_like(a)
    g = np.zeros_like(b)
    for _ in range(max_iter):
        g = reg * (lb - logsumexp((f[:, None] - M) / reg, axis=0))
        f_new = reg * (la - logsumexp((g[None, :] - M) / reg, axis=1))
        done = np.max(np.abs(f_new - f)) < tol
        f = f_new
        if done:
            break
    G = np.exp((f[:, None] + g[None, :] - M) / reg)
    return float(np.sum(G * M))


# ----------------------------------------------------------------- host-side steps
def cluster_representations(cell_type, sample_id, regulizer=0.2, normalization=True):
    """Trajectory.py:377-436, restated literally (first-appearance orders, C-1 prior,
    Python ``sum`` for the normaliser).  Returns (ordered dict sample->float64[K], cells)."""
    import pandas as pd
    cell_type = pd.Series(np.asarray(cell_type, dtype=object))
    sample_id = pd.Series(np.asarray(sample_id, dtype=object))
    cells = cell_type.unique()                      # :402
    n_total = len(cell_type)
    prior = np.ones(len(cells))
    for k, c in enumerate(cells):                   # :405-407   n_k / (C - 1)
        prior[k] = int((cell_type == c).sum()) / (n_total - 1)
    prior = prior * regulizer                       # :409
    out = {}
    for s in sample_id.unique():                    # :412-425
        mask = (sample_id == s).to_numpy()
        vec = np.zeros(len(cells))
        sub = cell_type[mask]
        for k, c in enumerate(cells):
            vec[k] = int((sub == c).sum())
        out[s] = vec
    if normalization:                               # :428-430
        for s in list(out):
            out[s] = (out[s] + prior) / (sum(out[s]) + sum(prior))
    return out, cells


def cost_matrix(data, cell_type, metric="cosine"):
    """Trajectory.py:441-475: per-type column-wise MEDIAN centroids (pandas .median on the
    frame's own dtype, then ``list`` -> Python floats), scipy pdist + squareform."""
    import pandas as pd
    import scipy.spatial.distance as ssd
    df = data if isinstance(data, pd.DataFrame) else pd.DataFrame(np.asarray(data))
    ct = pd.Series(np.asarray(cell_type, dtype=object))
    cells = ct.unique()
    centroids = []
    for c in cells:                                 # :465-466
        centroids.append(list(df[(ct == c).to_numpy()].median(axis=0)))
    dis = ssd.squareform(ssd.pdist(centroids, metric=metric), force="no", checks=True)  # :468-469
    return dis, np.asarray(centroids, dtype=np.float64), cells


def return_real_labels(sample_id, status):
    """Trajectory.py:617-642: first status value per sample, first-appearance order."""
    import pandas as pd
    sample_id = pd.Series(np.asarray(sample_id, dtype=object))
    status = pd.Series(np.asarray(status, dtype=object))
    return [status[(sample_id == s).to_numpy()].unique()[0] for s in sample_id.unique()]


def wasserstein_d(clu_rep, cost, regularized="unreg", reg=0.1, n_threads=1):
    """Trajectory.py:479-523: EMD[i, j] over all ordered pairs in dict order; the
    DataFrame is ``pd.DataFrame.from_dict(EMD).T`` (i.e. holds EMD transposed)."""
    import pandas as pd
    samples_id = list(clu_rep.keys())
    P = np.stack([np.asarray(clu_rep[s], dtype=np.float64) for s in samples_id])
    if regularized == "unreg":
        EMD = emd_grid(P, cost, n_threads=n_threads)
    else:
        EMD = sinkhorn_grid(P, cost, reg, n_threads=n_threads)
    emd = pd.DataFrame.from_dict(EMD).T
    emd.columns = samples_id
    emd["sampleID"] = samples_id
    emd = emd.set_index("sampleID")
    return EMD, emd


def cell_w2(X, Y, scale, reg, numItermax=1000, stopThr=1e-9, check_period=10, return_info=False):
    """fp64 oracle of the cell-level extension (pilot_ot_cell_w2_grid; NOT in the reference): POT 0.9.x
    ``ot.bregman.sinkhorn_log`` control flow on uniform weights and cost |x - y|^2 / scale; returns <Gamma, C>."""
    from scipy.special import logsumexp
    from scipy.spatial.distance import cdist
    X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    n, m = len(X), len(Y)
    M = cdist(X, Y, "sqeuclidean") / scale
    Mr = -M / reg
    loga, logb = np.full(n, -np.log(n)), np.full(m, -np.log(m))
    b = np.full(m, 1.0 / m)
    u, v = np.zeros(n), np.zeros(m)
    err, iters = 1.0, 0
    for ii in range(numItermax):
        v = logb - logsumexp(Mr + u[:, None], axis=0)
        u = loga - logsumexp(Mr + v[None, :], axis=1)
        iters = ii + 1
        if ii % check_period == 0:
            err = float(np.linalg.norm(np.exp(Mr + u[:, None] + v[None, :]).sum(0) - b))
            if err < stopThr:
                break
    val = float(np.sum(np.exp(Mr + u[:, None] + v[None, :]) * M))
    return (val, dict(iters=iters, err=err)) if return_info else val


def cell_w2_c(X, Y, scale, reg, numItermax=1000, stopThr=1e-9, check_period=10, n_threads=8, return_info=False):
    """:func:`cell_w2` in C with OpenMP (pilot_oracle.c::pilot_oracle_cell_w2): the same control flow and max-shifted
    log-sum-exps; affordable for converged pairs of thousands of cells.  Thread-count independent."""
    X, Y = _f64(X), _f64(Y)
    it = ctypes.c_int(0)
    err = ctypes.c_double(0.0)
    val = lib().pilot_oracle_cell_w2(_dptr(X), X.shape[0], _dptr(Y), Y.shape[0], X.shape[1], float(scale), float(reg),
                                     int(numItermax), float(stopThr), int(check_period), int(n_threads), ctypes.byref(it),
                                     ctypes.byref(err))
    return (val, dict(iters=it.value, err=err.value)) if return_info else val


def cell_w2_grid(X, offsets, scale, reg, row_begin=0, row_end=None, row_step=1, **kw):
    N = len(offsets) - 1
    rows = range(row_begin, N if row_end is None else row_end, row_step)
    out = np.zeros((len(rows), N))
    for r, i in enumerate(rows):
        for j in range(N):
            out[r, j] = cell_w2(X[offsets[i]:offsets[i + 1]], X[offsets[j]:offsets[j + 1]], scale, reg, **kw)
    return out
