"""The reference test's own cohort (Kidney_IgAN_G: 634 patients x 14 clusters x 14 features, fixture under tests/golden/) through
tl.wasserstein_distance and through the device-resident pair grid: wall time, kernel time, update counts, flags."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ.setdefault("PILOT_AMD_NO_RESULTS_DIR", "1")
import numpy as np
from conftest import GOLDEN_REAL, golden_adata, load_golden
from pilot_amd import engine, tl, _lib
g = load_golden(GOLDEN_REAL)
ad, cell_col = golden_adata(g)
for mode in ("unreg", "reg"):
    best = 1e9
    for r in range(4):
        ad.uns = {}
        t = time.perf_counter()
        tl.wasserstein_distance(ad, clusters_col=cell_col, sample_col="sampleID", status="status", data_type="Pathomics", regularized=mode, reg=0.1)
        if r: best = min(best, time.perf_counter() - t)
    want = g["emd_unreg"] if mode == "unreg" else g["emd_reg"]
    print("tl.wasserstein_distance(%s): %.2f ms end to end (24 227 glomeruli -> 634 x 634), max|EMD - reference fixture| on rows 0::3 = %.2e" % (
        mode, best * 1e3, np.abs(ad.uns["EMD"][::3] - want).max()), flush=True)
P = g["proportions"]; M = g["cost"] / g["cost"].max(); N = len(P)
plan = engine.DevicePlan(P, M); plan.enable_timing(True)
for _ in range(10): plan.run(0.1)
plan.sync()
a, b = plan.kernel_times_ms(10)
plan.enable_timing(False)
t = time.perf_counter()
for _ in range(10): plan.run(0.1)
plan.sync(); dt = (time.perf_counter() - t) / 10
_, info = plan.fetch()
it, fl = info["iters"], info["flags"]
def emd(): _lib.check(plan.L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, 2, 0, N, 1, plan.dE, plan.dIt, None))
emd(); plan.sync()
t = time.perf_counter()
for _ in range(5): emd()
plan.sync(); de = (time.perf_counter() - t) / 5
print("pair grid, device resident: Sinkhorn reg 0.1 %.3f ms per matrix (main %.3f, track %.3f) = %.3g pairs/s | updates mean %.1f max %d capped %d | flags %s | exact %.3f ms = %.3g pairs/s" % (
    dt * 1e3, a.mean(), b.mean(), N * N / dt, it.mean(), it.max(), (it >= 1000).sum(), {int(f): int((fl == f).sum()) for f in np.unique(fl)}, de * 1e3, N * N / de))
