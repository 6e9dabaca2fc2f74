"""Logical multi-device shards (repeated device 0) for random N, shard counts and gather modes must reproduce the
single-device matrices bit for bit.  Usage: python tools/fuzz_multi.py [n] [seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine, multi
from pilot_amd.synthetic import make_problem
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    N = int(rng.choice([1, 2, 3, 7, 16, 17, 33, 64, 100])); K = int(rng.choice([2, 5, 17, 50, 70])); G = int(rng.choice([1, 2, 3, 4, 5, 8]))
    P, M = make_problem(N, K, 4, seed=int(rng.integers(0, 1000)), cells_per_patient=60)
    reg = float(rng.choice([0.1, 0.02, 1.0]))
    gather = str(rng.choice(["auto", "copy"]))
    msgs = []
    try:
        ref, iref = engine.sinkhorn_grid(P, M, reg, return_info=True)
        got, ig = multi.sinkhorn_grid_multi(P, M, reg, devices=[0] * G, gather=gather, return_info=True)
        if not (np.array_equal(ref, got) and np.array_equal(iref["iters"], ig["iters"])): msgs.append("sinkhorn differs %.3e" % np.abs(ref - got).max())
        refe = engine.emd_grid(P, M)
        gote = multi.emd_grid_multi(P, M, devices=[0] * G, gather=gather)
        if not np.array_equal(refe, gote): msgs.append("emd differs %.3e" % np.abs(refe - gote).max())
    except Exception as e:
        msgs.append("%s: %s" % (type(e).__name__, e))
    tag = "N=%d K=%d G=%d reg=%g gather=%s" % (N, K, G, reg, gather)
    if msgs: bad += 1; print("FAIL", tag, "|", "; ".join(msgs), flush=True)
    else: print("ok  ", tag, flush=True)
print("%d of %d cases failed" % (bad, n_cases))
