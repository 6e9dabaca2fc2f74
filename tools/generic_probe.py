"""GPU probe: the POT-literal kernel (K > 128) against the number of resident workgroups (each keeps K' and its transpose,
2 K^2 doubles, in a global scratch: does the working set fit L2?).  usage: python tools/generic_probe.py"""
import os, sys, time, subprocess
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    import numpy as np
    from pilot_amd import engine
    from pilot_amd.synthetic import make_problem
    K = int(sys.argv[1]); N = 200
    wgs = sys.argv[2] if len(sys.argv) > 2 else None
    if wgs:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import switches
        switches.set("PILOT_OT_GENERIC_WGS", wgs)
    P, M = make_problem(N, K, 8, seed=K, cells_per_patient=800)
    E = engine.sinkhorn_grid(P, M, 0.1)
    t = time.perf_counter(); E = engine.sinkhorn_grid(P, M, 0.1); dt = time.perf_counter() - t
    print("K=%d workgroups %s: %.1f ms" % (K, wgs or "512", dt * 1e3), flush=True)
else:
    for K in (130, 192, 256):
        for w in (16, 32, 64, 96, 128, 256, 512):
            subprocess.run([sys.executable, __file__, str(K), str(w)])
