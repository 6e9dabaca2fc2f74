"""Where does the host-to-host time of engine.sinkhorn_grid go?  c3, reg 0.1."""
import sys, time, ctypes
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import _lib, engine
from pilot_amd.synthetic import CONFIGS, make_problem
P, M = make_problem(**CONFIGS["c3"])
N = P.shape[0]
L = _lib.load()
def t(f, n=30):
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(150): engine.sinkhorn_grid(P, M, 0.1)
print("engine.sinkhorn_grid (host in, host out)      %.3f ms" % t(lambda: engine.sinkhorn_grid(P, M, 0.1)))
print("  ... with return_info                        %.3f ms" % t(lambda: engine.sinkhorn_grid(P, M, 0.1, return_info=True)))
plan = engine.DevicePlan(P, M)
def dev():
    plan.run(0.1); plan.sync()
print("DevicePlan.run + sync (resident)              %.3f ms" % t(dev))
E = np.empty((N, N))
def d2h():
    _lib.check(L.pilot_ot_memcpy_d2h(E.ctypes.data, plan.dE, 8 * N * N))
print("pilot_ot_memcpy_d2h of the matrix (pageable)  %.3f ms" % t(d2h))
def h2d():
    _lib.check(L.pilot_ot_memcpy_h2d(plan.dP, P.ctypes.data, P.nbytes)); _lib.check(L.pilot_ot_memcpy_h2d(plan.dM, M.ctypes.data, M.nbytes))
print("pilot_ot_memcpy_h2d of P and M                %.3f ms" % t(h2d))
src = np.random.rand(N, N)
print("host memcpy of %d bytes                   %.3f ms" % (8 * N * N, t(lambda: np.copyto(E, src))))
print("np.empty((N, N)) + first touch                %.3f ms" % t(lambda: np.empty((N, N)).fill(0.0)))
