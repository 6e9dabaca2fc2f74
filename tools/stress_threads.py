"""Stress of the per-thread caches and helper threads added in round 3: tl.wasserstein_distance from several Python threads at
once (each spawns its uploads / copies, all share the one device-chain worker), short-lived threads whose cached contexts
must be reclaimed (registry + orphan reaping), shutdown in between, and device-memory use before / after (no growth).
Usage: python tools/stress_threads.py"""
import os, sys, threading, time, ctypes
sys.path.insert(0, ".")
os.environ.setdefault("PILOT_AMD_NO_RESULTS_DIR", "1")
import numpy as np
from pilot_amd import engine, tl, multi, _lib
from pilot_amd.synthetic import make_cells, make_problem

L = _lib.load()
hip = ctypes.CDLL("libamdhip64.so")
def free_mem():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value

cohorts = [make_cells(n, k, d, seed=s, cells_per_patient=c) for (n, k, d, s, c) in ((20, 10, 10, 0, 200), (37, 7, 5, 1, 90), (64, 30, 30, 2, 300))]
refs = []
for ad in cohorts:
    out = {}
    for mode in ("reg", "unreg"):
        ad.uns = {}
        tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized=mode)
        out[mode] = ad.uns["EMD"].copy()
    refs.append(out)
errors = []
def tl_worker(tid, reps):
    try:
        import copy
        for it in range(reps):
            i = (tid + it) % len(cohorts)
            ad = copy.copy(cohorts[i]); ad.uns = {}
            mode = "reg" if (it + tid) % 2 else "unreg"
            tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized=mode)
            if not np.array_equal(ad.uns["EMD"], refs[i][mode]): errors.append((tid, it, "EMD differs"))
            if ad.uns["data"].to_numpy().base is cohorts[i].obsm["X_pca"]: errors.append((tid, it, "data is a view"))
    except Exception as e:
        errors.append((tid, repr(e)))
t0 = time.time()
m0 = free_mem()
for rnd in range(3):
    ths = [threading.Thread(target=tl_worker, args=(t, 12)) for t in range(5)]
    [t.start() for t in ths]; [t.join() for t in ths]
print("tl from 5 threads x 3 rounds:", "ok" if not errors else errors[:5], "(%.1f s)" % (time.time() - t0))
# short-lived threads using the host entry points: their contexts are reaped by later threads / shutdown
P, M = make_problem(50, 20, 6, seed=3, cells_per_patient=300)
Es = engine.sinkhorn_grid(P, M, 0.1)
def short(tid):
    try:
        if not np.array_equal(engine.sinkhorn_grid(P, M, 0.1), Es): errors.append((tid, "short sinkhorn"))
        if not np.array_equal(multi.sinkhorn_grid_multi(P, M, 0.1, devices=[0, 0]), Es): errors.append((tid, "short multi"))
        engine.pdist_square(np.random.default_rng(tid).random((9, 4)), "cosine")
    except Exception as e:
        errors.append((tid, repr(e)))
for k in range(40):
    t = threading.Thread(target=short, args=(k,)); t.start(); t.join()
print("40 short-lived threads:", "ok" if not errors else errors[:5])
L.pilot_ot_shutdown()
m1 = free_mem()
for k in range(40):
    t = threading.Thread(target=short, args=(k,)); t.start(); t.join()
tl_worker(0, 6)
L.pilot_ot_shutdown()
m2 = free_mem()
print("free device memory: start %.1f MB, after round one + shutdown %.1f MB, after round two + shutdown %.1f MB" % (m0 / 2**20, m1 / 2**20, m2 / 2**20))
leak = (m1 - m2) / 2**20
print("growth between the two shutdowns: %.1f MB %s" % (leak, "ok" if leak < 64 else "LEAK?"))
print("errors:", errors[:5] if errors else "none", " total %.1f s" % (time.time() - t0))
