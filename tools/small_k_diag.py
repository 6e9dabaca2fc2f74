"""Which pairs of a small-K grid leave the MFMA kernels for the POT-literal one, and why: PILOT_OT_DEBUG=1024 skips the
hand-over launch, so the pairs it would have solved keep the sentinel this script writes into the outputs first."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
__import__("sys").path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__))); import switches; switches.set("PILOT_OT_DEBUG", "1024")
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem
from oracle import oracle as O
N = 600
for K in [int(a) for a in sys.argv[1:]] or [2, 4]:
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=200)
    for prec in ("auto", "bf16x3", "fp32", "fp64"):
        plan = engine.DevicePlan(P, M)
        sent = np.full(N * N, -7, dtype=np.int32)
        _lib.check(plan.L.pilot_ot_memcpy_h2d(plan.dFl, sent.ctypes.data, sent.nbytes))
        plan.run(0.1, precision=prec); plan.sync()
        t = time.perf_counter()
        for _ in range(5): plan.run(0.1, precision=prec)
        plan.sync(); dt = (time.perf_counter() - t) / 5
        E, info = plan.fetch()
        fl, it = info["flags"], info["iters"]
        lost = np.argwhere(fl == -7)
        print("K=%d %-6s %.3f ms without the hand-over launch | pairs left to it: %d | flags %s" % (
            K, prec, dt * 1e3, len(lost), {int(f): int((fl == f).sum()) for f in np.unique(fl)}), flush=True)
        for (i, j) in lost[:4]:
            v, oi = O.sinkhorn2(P[i], P[j], M, 0.1, return_info=True)
            print("   pair (%d, %d): a=%s b=%s | oracle value %.6g iters %d flags %d n_absorb %d last_absorb %d" % (
                i, j, np.array2string(P[i], precision=5), np.array2string(P[j], precision=5), v, oi["iters"], oi["flags"], oi["n_absorb"], oi["last_absorb"]))
        plan.close()
