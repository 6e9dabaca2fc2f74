#!/usr/bin/env python3
"""Exact OT at small K: the several-pairs-per-wave kernel (PILOT_OT_EMD_MULTI=1, default; =2: flows in the global slab) against the
one-pair-per-wave kernel (=0) and the oracle's network simplex, whole grids.  usage: emd_multi_probe.py [K ...] (default: a sweep + the
Kidney_IgAN_G cohort of tests/golden)"""
import os, sys, time
__import__("sys").path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))   # tools/switches.py
import switches
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from pilot_amd import engine, _lib
from oracle import oracle as O

def cohort(k):
    if k == "real":
        from conftest import GOLDEN_REAL, load_golden
        g = load_golden(GOLDEN_REAL)
        return g["proportions"], g["cost"] / g["cost"].max()
    from pilot_amd.synthetic import make_problem
    return make_problem(600, int(k), 8, seed=int(k), cells_per_patient=200)

def run(P, M, mode, reps=5):
    switches.set("PILOT_OT_EMD_MULTI", mode)
    N, K = P.shape
    plan = engine.DevicePlan(P, M)
    def emd(): _lib.check(plan.L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, 2, 0, N, 1, plan.dE, plan.dIt, None))
    for _ in range(2): emd()
    plan.sync()
    t = time.perf_counter()
    for _ in range(reps): emd()
    plan.sync(); dt = (time.perf_counter() - t) / reps
    E = np.empty((N, N)); n_aug = np.empty((N, N), dtype=np.int32)
    _lib.check(plan.L.pilot_ot_memcpy_d2h(E.ctypes.data, plan.dE, 8 * N * N))
    _lib.check(plan.L.pilot_ot_memcpy_d2h(n_aug.ctypes.data, plan.dIt, 4 * N * N))
    plan.close()
    return dt * 1e3, E, n_aug

ks = sys.argv[1:] or ["2", "3", "4", "5", "8", "12", "real", "16", "17", "24", "30", "32"]
for k in ks:
    P, M = cohort(k)
    N, K = P.shape
    Eo = O.emd_grid(P, M, n_threads=16, fast="ns") if N * N * K <= 600 * 600 * 32 else None
    iu = np.triu_indices(N)
    line = "N=%d K=%d:" % (N, K)
    for mode in ([0, 1, 2] if K <= 16 else [0, 1]):
        ms, E, na = run(P, M, mode)
        bad = int((na[iu] < 0).sum()) + int((~np.isfinite(E)).sum())
        d = np.abs(E - Eo).max() if Eo is not None else float("nan")
        sym = np.abs(E - E.T).max()
        line += "  mode %d: %.3f ms, %.1f aug/pair, max|d| %.1e, |E-E^T| %.1e%s;" % (mode, ms, na[iu].mean(), d, sym, (" BAD %d" % bad) if bad else "")
    print(line, flush=True)
