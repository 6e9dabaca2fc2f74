"""API stress: plan churn, interleaved shapes, concurrent host calls from several threads.  Usage: python tools/stress_api.py"""
import sys, threading, time
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem
L = _lib.load()
t0 = time.time()
ref = {}
shapes = [(20, 10), (64, 30), (33, 50), (7, 3), (100, 17)]
for (N, K) in shapes:
    P, M = make_problem(N, K, 5, seed=N + K, cells_per_patient=80)
    ref[(N, K)] = (P, M, engine.sinkhorn_grid(P, M, 0.1), engine.emd_grid(P, M))
for it in range(60):                                   # plan churn + interleaved shapes on the host entry points (cached context)
    N, K = shapes[it % len(shapes)]
    P, M, Es, Ee = ref[(N, K)]
    plan = engine.DevicePlan(P, M)
    plan.run(0.1); E, _ = plan.fetch()
    assert np.array_equal(E, Es), (it, N, K)
    plan.close()
    assert np.array_equal(engine.emd_grid(P, M), Ee)
    assert np.array_equal(engine.sinkhorn_grid(P, M, 0.1), Es)
print("plan churn ok (%.1f s)" % (time.time() - t0))
errors = []
def worker(tid):
    try:
        for it in range(25):
            N, K = shapes[(it + tid) % len(shapes)]
            P, M, Es, Ee = ref[(N, K)]
            if not np.array_equal(engine.sinkhorn_grid(P, M, 0.1), Es): errors.append((tid, it, "sinkhorn"))
            if not np.array_equal(engine.emd_grid(P, M), Ee): errors.append((tid, it, "emd"))
    except Exception as e:
        errors.append((tid, repr(e)))
ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
[t.start() for t in ths]; [t.join() for t in ths]
print("threads:", "ok" if not errors else errors[:5])
L.pilot_ot_shutdown()
P, M, Es, Ee = ref[(20, 10)]
assert np.array_equal(engine.sinkhorn_grid(P, M, 0.1), Es)
print("after shutdown ok; total %.1f s" % (time.time() - t0))
