"""ctypes binding of ``libpilot_ot.so`` (C ABI: ``include/pilot_ot.h``).

This is the only way the Python host reaches the device.  There is no CPU code path behind
it: if the shared library is missing or no gfx950 device is visible, the compute calls raise.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PILOT_AMD_LIB") or os.path.join(_HERE, "libpilot_ot.so")      # (PILOT_AMD_LIB: an alternative build, A/B experiments)

OK, EINVAL, EHIP, ENOTSUP, ERCCL = 0, -1, -2, -3, -4
PREC = {"auto": 0, "fp32": 1, "f32": 1, "float32": 1, "fp64": 2, "f64": 2, "float64": 2, "bf16x3": 3, "generic": 5, "f16x2": 6}
METRICS = {"cosine": 0, "euclidean": 1, "sqeuclidean": 2, "cityblock": 3, "chebyshev": 4, "correlation": 5,
           "minkowski": 6, "seuclidean": 7, "braycurtis": 8, "canberra": 9, "hamming": 10, "jaccard": 11, "dice": 12, "yule": 13,
           "russellrao": 14, "sokalsneath": 15, "rogerstanimoto": 16, "sokalmichener": 17, "kulczynski1": 18, "jensenshannon": 19,
           "mahalanobis": 20}

EMD_ALL, EMD_UPPER, EMD_MIRROR = 0, 1, 2
FLAG_CONVERGED, FLAG_NAN, FLAG_ABSORB_LAST, FLAG_ABSORBED, FLAG_F64 = 1, 2, 4, 8, 16

# every symbol include/pilot_ot.h declares (tests check the library exports all of them)
SYMBOLS = [
    "pilot_ot_version", "pilot_ot_last_error", "pilot_ot_device_count", "pilot_ot_set_device", "pilot_ot_get_device",
    "pilot_ot_device_name", "pilot_ot_dev_alloc", "pilot_ot_dev_free", "pilot_ot_memcpy_h2d",
    "pilot_ot_memcpy_d2h", "pilot_ot_stream_sync", "pilot_ot_cost_matrix", "pilot_ot_cost_matrix_dev",
    "pilot_ot_cost_matrix_ex", "pilot_ot_cost_matrix_dev_ex",
    "pilot_ot_sinkhorn_grid", "pilot_ot_plan_create", "pilot_ot_plan_destroy", "pilot_ot_plan_set_max_cost",
    "pilot_ot_sinkhorn_grid_dev", "pilot_ot_auto_precision", "pilot_ot_auto_precision_for", "pilot_ot_resolve_precision", "pilot_ot_emd_grid", "pilot_ot_emd_grid_dev",
    "pilot_ot_plan_enable_timing", "pilot_ot_plan_kernel_times", "pilot_ot_plan_enable_graph", "pilot_ot_shutdown",
    "pilot_ot_proportions", "pilot_ot_proportions_ex", "pilot_ot_centroid_medians", "pilot_ot_embedding_upload",
    "pilot_ot_embedding_destroy", "pilot_ot_centroid_medians_dev", "pilot_ot_prepass_dev", "pilot_ot_prepass_device_ms", "pilot_ot_label_codes", "pilot_ot_test_switch", "pilot_ot_cell_w2_grid",
    "pilot_ot_cell_cohort_create", "pilot_ot_cell_cohort_destroy", "pilot_ot_cell_cohort_pieces", "pilot_ot_cell_w2_grid_cohort", "pilot_ot_cell_w2_grid_multi",
    "pilot_ot_mirror_upper_dev",
    "pilot_ot_row_distances", "pilot_ot_row_distances_dev", "pilot_ot_silhouette", "pilot_ot_knn_kernel",
    "pilot_ot_silhouette_dev", "pilot_ot_knn_kernel_dev", "pilot_ot_silhouette_of_rows", "pilot_ot_diffusion_kernel_of_rows",
    "pilot_ot_multi_create", "pilot_ot_multi_destroy", "pilot_ot_multi_set_inputs", "pilot_ot_multi_sinkhorn",
    "pilot_ot_multi_emd", "pilot_ot_multi_sync", "pilot_ot_multi_fetch", "pilot_ot_multi_device_matrix",
    "pilot_ot_multi_times", "pilot_ot_multi_rccl_info", "pilot_ot_sinkhorn_grid_multi", "pilot_ot_emd_grid_multi",
    "pilot_ot_comm_unique_id", "pilot_ot_comm_init_rank", "pilot_ot_comm_destroy", "pilot_ot_comm_info",
    "pilot_ot_comm_all_gather_rows", "pilot_ot_comm_all_reduce_max",
]
GATHER = {"auto": 0, "rccl": 1, "copy": 2}
ROW_METRICS = {"euclidean": 0, "cosine": 1}
UNIQUE_ID_BYTES = 128

_lib = None


class PilotOTError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libpilot_ot.so; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PilotOTError(
            "libpilot_ot.so not found at %s -- build it with `make -C pilot_amd/csrc` "
            "(or __graft_entry__.build()); pilot_amd has no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_dbl, c_vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p
    dp, ip = ctypes.POINTER(c_dbl), ctypes.POINTER(c_int)
    L.pilot_ot_version.restype = c_int
    L.pilot_ot_last_error.restype = ctypes.c_char_p
    L.pilot_ot_device_count.argtypes = [ip]
    L.pilot_ot_set_device.argtypes = [c_int]
    L.pilot_ot_get_device.argtypes = [ip]
    L.pilot_ot_device_name.argtypes = [ctypes.c_char_p, c_int]
    L.pilot_ot_dev_alloc.argtypes = [ctypes.POINTER(c_vp), ctypes.c_ulonglong]
    L.pilot_ot_dev_free.argtypes = [c_vp]
    L.pilot_ot_memcpy_h2d.argtypes = [c_vp, c_vp, ctypes.c_ulonglong]
    L.pilot_ot_memcpy_d2h.argtypes = [c_vp, c_vp, ctypes.c_ulonglong]
    L.pilot_ot_stream_sync.argtypes = [c_vp]
    L.pilot_ot_cost_matrix.argtypes = [dp, c_int, c_int, c_int, dp]
    L.pilot_ot_cost_matrix_dev.argtypes = [c_vp, c_int, c_int, c_int, c_vp, c_vp]
    L.pilot_ot_cost_matrix_ex.argtypes = [dp, c_int, c_int, c_int, dp, dp]
    L.pilot_ot_cost_matrix_dev_ex.argtypes = [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]
    L.pilot_ot_sinkhorn_grid.argtypes = [dp, c_int, c_int, dp, c_dbl, c_int, c_dbl, c_dbl, c_int, c_int,
                                         c_dbl, c_int, c_int, c_int, c_int, dp, ip, dp, ip]
    L.pilot_ot_plan_create.argtypes = [c_int, c_int, ctypes.POINTER(c_vp)]
    L.pilot_ot_plan_set_max_cost.argtypes = [c_vp, c_dbl]
    L.pilot_ot_plan_destroy.argtypes = [c_vp]
    L.pilot_ot_sinkhorn_grid_dev.argtypes = [c_vp, c_vp, c_vp, c_dbl, c_int, c_dbl, c_dbl, c_int, c_int, c_dbl,
                                             c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.pilot_ot_auto_precision.argtypes = [c_dbl]
    L.pilot_ot_auto_precision_for.argtypes = [c_dbl, c_int, c_int]
    L.pilot_ot_resolve_precision.argtypes = [c_int, c_dbl, c_int, c_int, c_dbl]
    L.pilot_ot_plan_enable_timing.argtypes = [c_vp, c_int]
    L.pilot_ot_plan_enable_graph.argtypes = [c_vp, c_int]
    L.pilot_ot_plan_kernel_times.argtypes = [c_vp, c_int, ctypes.POINTER(ctypes.c_float),
                                             ctypes.POINTER(ctypes.c_float), ip]
    L.pilot_ot_proportions.argtypes = [ip, ip, ctypes.c_longlong, ctypes.c_longlong, c_int, c_int, c_dbl, c_int, dp]
    L.pilot_ot_centroid_medians.argtypes = [c_vp, c_int, ctypes.c_longlong, c_int, ip, c_int, dp]
    L.pilot_ot_proportions_ex.argtypes = [ip, ip, ctypes.c_longlong, ctypes.c_longlong, c_int, c_int, c_dbl, c_int, dp,
                                          ctypes.POINTER(ctypes.c_longlong)]
    L.pilot_ot_embedding_upload.argtypes = [c_vp, c_int, ctypes.c_longlong, c_int, ctypes.POINTER(c_vp)]
    L.pilot_ot_embedding_destroy.argtypes = [c_vp]
    L.pilot_ot_centroid_medians_dev.argtypes = [c_vp, ip, c_int, dp]
    L.pilot_ot_prepass_dev.argtypes = [c_vp, ip, ip, ctypes.c_longlong, c_int, c_int, c_dbl, c_int, dp, ctypes.POINTER(ctypes.c_longlong), dp]
    L.pilot_ot_prepass_device_ms.argtypes = [ctypes.POINTER(ctypes.c_float)]
    L.pilot_ot_test_switch.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.pilot_ot_label_codes.argtypes = [c_vp, c_int, ctypes.c_longlong, c_int, c_int, ip, ctypes.POINTER(ctypes.c_longlong), ip]
    L.pilot_ot_cell_w2_grid.argtypes = [c_vp, c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_dbl, c_int, c_dbl, c_int, c_int, c_int,
                                        dp, ip, dp]
    L.pilot_ot_cell_cohort_create.argtypes = [c_vp, c_vp, c_int, c_int, ctypes.POINTER(c_vp)]
    L.pilot_ot_cell_cohort_destroy.argtypes = [c_vp]
    L.pilot_ot_cell_cohort_pieces.argtypes = [c_vp, ip]
    L.pilot_ot_cell_w2_grid_cohort.argtypes = [c_vp, c_dbl, c_dbl, c_int, c_dbl, c_int, c_dbl, c_int, c_int, c_int, dp, ip, dp,
                                               ctypes.POINTER(ctypes.c_float)]
    L.pilot_ot_cell_w2_grid_multi.argtypes = [c_vp, c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_dbl, c_int, c_dbl, ip, c_int, dp, ip, dp]
    L.pilot_ot_emd_grid.argtypes = [dp, c_int, c_int, dp, c_int, c_int, c_int, c_int, dp, ip]
    L.pilot_ot_emd_grid_dev.argtypes = [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]
    fp = ctypes.POINTER(ctypes.c_float)
    L.pilot_ot_mirror_upper_dev.argtypes = [c_vp, c_int, c_vp]
    L.pilot_ot_row_distances.argtypes = [dp, c_int, c_int, c_int, dp]
    L.pilot_ot_row_distances_dev.argtypes = [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]
    L.pilot_ot_silhouette.argtypes = [dp, ip, c_int, c_int, dp, dp]
    L.pilot_ot_knn_kernel.argtypes = [dp, c_int, c_int, c_dbl, dp]
    L.pilot_ot_silhouette_dev.argtypes = [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]
    L.pilot_ot_knn_kernel_dev.argtypes = [c_vp, c_int, c_int, c_dbl, c_vp, c_vp]
    L.pilot_ot_silhouette_of_rows.argtypes = [c_vp, c_int, c_int, c_int, c_int, ip, c_int, dp, dp]
    L.pilot_ot_diffusion_kernel_of_rows.argtypes = [c_vp, c_int, c_int, c_int, c_dbl, dp, dp]
    L.pilot_ot_multi_create.argtypes = [c_int, c_int, ip, c_int, c_int, ctypes.POINTER(c_vp)]
    L.pilot_ot_multi_destroy.argtypes = [c_vp]
    L.pilot_ot_multi_set_inputs.argtypes = [c_vp, dp, dp]
    L.pilot_ot_multi_sinkhorn.argtypes = [c_vp, c_dbl, c_int, c_dbl, c_dbl, c_int, c_int, c_dbl, c_int]
    L.pilot_ot_multi_emd.argtypes = [c_vp, c_int]
    L.pilot_ot_multi_sync.argtypes = [c_vp]
    L.pilot_ot_multi_fetch.argtypes = [c_vp, dp, ip, dp, ip]
    L.pilot_ot_multi_device_matrix.argtypes = [c_vp, c_int, ctypes.POINTER(c_vp)]
    L.pilot_ot_multi_times.argtypes = [c_vp, fp, fp]
    L.pilot_ot_multi_rccl_info.argtypes = [c_vp, ip, ip]
    L.pilot_ot_comm_info.argtypes = [c_vp, ip, ip]
    L.pilot_ot_sinkhorn_grid_multi.argtypes = [dp, c_int, c_int, dp, c_dbl, c_int, c_dbl, c_dbl, c_int, c_int, c_dbl,
                                               c_int, ip, c_int, c_int, dp, ip, dp, ip]
    L.pilot_ot_emd_grid_multi.argtypes = [dp, c_int, c_int, dp, c_int, ip, c_int, c_int, dp, ip]
    L.pilot_ot_comm_unique_id.argtypes = [ctypes.c_char_p]
    L.pilot_ot_comm_init_rank.argtypes = [ctypes.c_char_p, c_int, c_int, ctypes.POINTER(c_vp)]
    L.pilot_ot_comm_destroy.argtypes = [c_vp]
    L.pilot_ot_comm_all_gather_rows.argtypes = [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]
    L.pilot_ot_comm_all_reduce_max.argtypes = [c_vp, c_vp, c_int, c_vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name != "pilot_ot_last_error":
            fn.restype = c_int
    _lib = L
    return L


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = load().pilot_ot_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise ValueError("pilot_ot: " + msg)
    if rc == ENOTSUP:
        raise NotImplementedError("pilot_ot: " + msg)
    if rc == ERCCL:
        raise PilotOTError("pilot_ot (RCCL): " + msg)
    raise PilotOTError("pilot_ot (HIP): " + msg)


def test_switch(name=None, value=None) -> None:
    """TEST HOOK (``pilot_ot_test_switch``): force a kernel variant; ``value=None`` clears the switch, ``name=None`` all of them."""
    check(load().pilot_ot_test_switch(None if name is None else name.encode(), None if value is None else str(value).encode()))


def device_count() -> int:
    n = ctypes.c_int(0)
    check(load().pilot_ot_device_count(ctypes.byref(n)))
    return n.value


def device_name() -> str:
    buf = ctypes.create_string_buffer(128)
    check(load().pilot_ot_device_name(buf, 128))
    return buf.value.decode()


def dptr(x: np.ndarray):
    return x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def iptr(x: np.ndarray):
    return x.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
