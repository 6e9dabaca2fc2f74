"""GPU probe: c3 at reg 0.01 -- mixed precision (AUTO) vs f64, timing and parity on 20 rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from pilot_amd import _lib, engine
from pilot_amd.synthetic import CONFIGS, make_problem
P, M = make_problem(**CONFIGS["c3"])
REG = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
rows = dict(row_begin=7, row_end=600, row_step=30)
Eo, io = O.sinkhorn_grid(P, M, REG, n_threads=64, return_info=True, **rows)
for prec in ("auto", "fp64", "bf16x3"):
    E, inf = engine.sinkhorn_grid(P, M, REG, precision=prec, return_info=True, **rows)
    f64 = (inf["flags"] & _lib.FLAG_F64) > 0
    last = ((io["flags"] & O.FLAG_ABSORB_ON_LAST) > 0) | ((inf["flags"] & _lib.FLAG_ABSORB_LAST) > 0)
    same = inf["iters"] == io["iters"]
    d = np.abs(E - Eo)
    print("%-7s f64 pairs %5d/%d  nan %d  max|d| all %.3e  not-last %.3e  same-iters %.3e (n=%d)  f64-only %.3e"
          % (prec, f64.sum(), E.size, np.isnan(E).sum(), np.nanmax(d), np.nanmax(d[~last]), np.nanmax(d[same & ~last]), same.sum(),
             np.nanmax(d[f64]) if f64.any() else 0.0))
    plan = engine.DevicePlan(P, M)
    for _ in range(3):
        plan.run(REG, precision=prec)
    plan.sync()
    t = time.perf_counter()
    for _ in range(5):
        plan.run(REG, precision=prec)
    plan.sync()
    Ef, inff = plan.fetch()
    print("        full grid %.2f ms per matrix; f64 pairs %d of %d" % ((time.perf_counter() - t) / 5 * 1e3, ((inff["flags"] & 16) > 0).sum(), Ef.size))
    plan.close()
