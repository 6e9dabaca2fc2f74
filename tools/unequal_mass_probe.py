import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from oracle import oracle as O
from pilot_amd import engine
from pilot_amd.synthetic import make_problem
P, M = make_problem(16, 20, 6, seed=5, cells_per_patient=400)
S = P * np.linspace(0.5, 2.0, 16)[:, None]
Eo, io = O.sinkhorn_grid(S, M, 0.1, return_info=True, n_threads=8)
for prec in ("auto", "fp32", "fp64"):
    Eg, ig = engine.sinkhorn_grid(S, M, 0.1, precision=prec, return_info=True)
    d = np.abs(Eg - Eo)
    bad = np.argwhere(d > 1e-5)
    print(prec, "bad pairs", len(bad), "flags gpu", np.unique(ig["flags"]), "flags oracle", np.unique(io["flags"]))
    for (i, j) in bad[:6]:
        print("   (%d,%d) oracle %.6f it %d fl %d err %.3e | gpu %.6f it %d fl %d err %.3e" % (i, j, Eo[i, j], io["iters"][i, j], io["flags"][i, j], io["err"][i, j], Eg[i, j], ig["iters"][i, j], ig["flags"][i, j], ig["err"][i, j]))
print("oracle iters hist", np.unique(io["iters"], return_counts=True))
