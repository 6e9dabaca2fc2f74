"""Edge inputs for the Sinkhorn grid against the oracle: unequal masses, huge / tiny reg, constant cost, N = 1, tau variants."""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from pilot_amd import engine
from pilot_amd.synthetic import make_problem
def cmp(tag, P, M, reg, **kw):
    Eo, io = O.sinkhorn_grid(P, M, reg, return_info=True, n_threads=8, **{k: v for k, v in kw.items() if k != "precisions"})
    for prec in kw.get("precisions", ("auto", "fp64")):
        okw = {"num_iter_max": kw.get("numItermax", 1000), "tau": kw.get("tau", 1e3), "stop_thr": kw.get("stopThr", 1e-9)}
        Eg, ig = engine.sinkhorn_grid(P, M, reg, precision=prec, return_info=True, **okw)
        both = np.isfinite(Eo) & np.isfinite(Eg)
        print("%-34s %-5s nan o/g %d/%d  max|d| %.3e  rel %.3e  iters equal %d/%d" % (tag, prec, np.isnan(Eo).sum(), np.isnan(Eg).sum(),
              np.abs(Eg - Eo)[both].max(), (np.abs(Eg - Eo)[both] / np.maximum(np.abs(Eo[both]), 1e-300)).max(), (ig["iters"] == io["iters"]).sum(), Eo.size))
P, M = make_problem(16, 20, 6, seed=5, cells_per_patient=400)
cmp("baseline", P, M, 0.1)
S = P * np.linspace(0.5, 2.0, 16)[:, None]
cmp("unequal masses (rows x 0.5..2)", S, M, 0.1)
cmp("tiny masses (x 1e-6)", P * 1e-6, M, 0.1)
cmp("huge reg 100", P, M, 100.0)
cmp("reg 0.005", P, M, 0.005)
cmp("constant cost", P, np.ones_like(M) - np.eye(20), 0.1)
cmp("zero cost", P, np.zeros_like(M), 0.1)
cmp("one patient", P[:1], M, 0.1)
cmp("tau 5 (absorbs all the time)", P, M, 0.1, tau=5.0)
cmp("cap 3 updates", P, M, 0.1, numItermax=3)
cmp("stopThr 1e-3", P, M, 0.1, stopThr=1e-3)
