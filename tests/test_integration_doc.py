"""INTEGRATION.md prints the ctypes stub a maintainer of the reference would paste into pilotpy/tools/Trajectory.py.  These
tests EXECUTE that text, so the document cannot drift from include/pilot_ot.h: the CPU test holds its argtypes to the ones
pilot_amd/_lib.py declares (which test_abi.py holds to the header), the GPU test runs its wasserstein_d and cost_matrix lines on
BASELINE config 1 and compares with pilot_amd.engine bit for bit."""
import ctypes
import os
import re

import numpy as np
import pandas as pd
import pytest

from pilot_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"## The stub a maintainer adds.*?```python\n(.*?)```", text, re.S)
    assert m, "INTEGRATION.md lost its python stub"
    return m.group(1)


def _load_stub():
    src = _stub_source().replace('ctypes.CDLL("libpilot_ot.so")', "ctypes.CDLL(%r)" % _lib.LIB_PATH)
    ns = {"pd": pd}                       # (Trajectory.py imports pandas as pd already)
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    return ns, src


def test_documented_argtypes_are_the_bindings_argtypes():
    ns, _ = _load_stub()
    L = _lib.load()
    for name in ("pilot_ot_sinkhorn_grid", "pilot_ot_emd_grid", "pilot_ot_cost_matrix"):
        doc = list(getattr(ns["_L"], name).argtypes)
        ours = list(getattr(L, name).argtypes)
        assert len(doc) == len(ours), name
        for i, (a, b) in enumerate(zip(doc, ours)):
            assert ctypes.sizeof(a) == ctypes.sizeof(b) and (a is b or a._type_ == getattr(b, "_type_", None)), (name, i, a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["unreg", "reg"])
def test_documented_stub_runs_and_matches_the_engine(mode):
    from pilot_amd import engine
    from pilot_amd.synthetic import CONFIGS, make_problem
    ns, src = _load_stub()
    P, M = make_problem(**CONFIGS["c1"])
    rep = {"s%d" % i: P[i] for i in range(P.shape[0])}
    EMD, frame = ns["wasserstein_d"](rep, M, regularized=mode, reg=0.1)
    want = engine.emd_grid(P, M) if mode == "unreg" else engine.sinkhorn_grid(P, M, 0.1)
    np.testing.assert_array_equal(EMD, want)
    assert list(frame.index) == list(rep) and list(frame.columns) == list(rep) and frame.index.name == "sampleID"
    np.testing.assert_array_equal(frame.to_numpy(), EMD.T)
    # the cost_matrix lines (printed as comments: they go INSIDE the reference's function)
    lines = re.search(r"# in cost_matrix.*?\n((?:#   .*\n)+)", src).group(1)
    body = "\n".join(l[4:] for l in lines.splitlines())
    centroids = np.random.default_rng(0).standard_normal((7, 5))
    env = dict(ns, centroids=centroids, metric="cosine", METRIC_ID=_lib.METRICS)
    exec(compile(body, "INTEGRATION.md:cost_matrix", "exec"), env)
    np.testing.assert_array_equal(env["dis"], engine.pdist_square(centroids, "cosine"))
