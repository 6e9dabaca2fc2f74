"""The C-ABI shared library: loads, exports every symbol include/pilot_ot.h declares, validates
arguments, and FAILS LOUDLY without a GPU (no CPU fallback).  CPU only: no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from pilot_amd import _lib, engine


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pilot_ot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pilot_ot_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.SYMBOLS) == names
    assert L.pilot_ot_version() == 100


def test_header_compiles_as_plain_c(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "pilot_ot.h"\nint main(void){return PILOT_OT_OK;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "t.o")], check=True)


def test_argument_validation_needs_no_device():
    L = _lib.load()
    P = np.full((4, 3), 1 / 3.0); M = np.zeros((3, 3)); out = np.zeros((4, 4))
    dp = _lib.dptr
    # reg <= 0
    rc = L.pilot_ot_sinkhorn_grid(dp(P), 4, 3, dp(M), -1.0, 1000, 1e-9, 1e3, 20, 0, 0.0, 1, 0, 4, 1, dp(out), None, None, None)
    assert rc == _lib.EINVAL and b"reg" in L.pilot_ot_last_error()
    # bad row range
    rc = L.pilot_ot_sinkhorn_grid(dp(P), 4, 3, dp(M), 0.1, 1000, 1e-9, 1e3, 20, 0, 0.0, 1, 0, 9, 1, dp(out), None, None, None)
    assert rc == _lib.EINVAL and b"row range" in L.pilot_ot_last_error()
    # NULL pointer
    rc = L.pilot_ot_sinkhorn_grid(None, 4, 3, dp(M), 0.1, 1000, 1e-9, 1e3, 20, 0, 0.0, 1, 0, 4, 1, dp(out), None, None, None)
    assert rc == _lib.EINVAL
    # K beyond every kernel's range is ENOTSUP, not a silent fallback (K <= 128: MFMA kernels, <= 2048: reference-semantics kernel)
    rc = L.pilot_ot_sinkhorn_grid(dp(P), 4, 4000, dp(M), 0.1, 1000, 1e-9, 1e3, 20, 0, 0.0, 1, 0, 4, 1, dp(out), None, None, None)
    assert rc == _lib.ENOTSUP
    # reg so small that exp(-M/reg) leaves the f64 range is a valid call (it runs the reference-semantics kernel):
    # on this box it gets as far as needing a device
    M1 = np.ones((3, 3)) - np.eye(3)
    rc = L.pilot_ot_sinkhorn_grid(dp(P), 4, 3, dp(M1), 1e-3, 1000, 1e-9, 1e3, 20, 0, 0.0, 1, 0, 4, 1, dp(out), None, None, None)
    assert rc in (_lib.OK, _lib.EHIP)
    rc = L.pilot_ot_cost_matrix(dp(P), 4, 3, 99, dp(out))
    assert rc in (_lib.EINVAL, _lib.EHIP)
    assert L.pilot_ot_auto_precision(10.0) == 6 and L.pilot_ot_auto_precision(30.0) == 3 and L.pilot_ot_auto_precision(100.0) == 2     # f16x2 / bf16x3 / f64
    # the fp16-split configuration reaches max(M)/reg = 16 (sinkhorn_kernels.hpp, H_MAX_COST_OVER_REG), by name too
    assert L.pilot_ot_auto_precision(16.0) == 6 and L.pilot_ot_auto_precision(16.1) == 3
    L.pilot_ot_resolve_precision.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    assert L.pilot_ot_resolve_precision(6, 15.0, 50, 1, 1e3) == 6 and L.pilot_ot_resolve_precision(6, 17.0, 50, 1, 1e3) == 3
    assert L.pilot_ot_resolve_precision(6, 10.0, 50, 1, 5e3) == 3                # tau beyond the scaled fp16 domain


def test_python_wrappers_validate_shapes():
    with pytest.raises(ValueError):
        engine.sinkhorn_grid(np.ones((3, 4)), np.ones((3, 3)), 0.1)
    with pytest.raises(ValueError):
        engine.sinkhorn_grid(np.ones((3, 3)), np.ones((3, 3)), 0.1, precision="fp16")
    with pytest.raises(ValueError):
        engine.sinkhorn_grid(np.full((3, 3), np.nan), np.ones((3, 3)), 0.1)
    with pytest.raises(NotImplementedError):
        engine.pdist_square(np.ones((3, 3)), metric="wminkowski")


def test_compute_fails_loudly_without_a_gpu():
    """On the CPU box there is no device: the product path must raise, never compute on the host."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    P = np.full((4, 3), 1 / 3.0); M = np.ones((3, 3)) - np.eye(3)
    with pytest.raises(_lib.PilotOTError):
        engine.sinkhorn_grid(P, M, 0.1)
    with pytest.raises(_lib.PilotOTError):
        engine.emd_grid(P, M)
    with pytest.raises(_lib.PilotOTError):
        engine.pdist_square(P)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pilot_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no CPU fallback", ""), fn
