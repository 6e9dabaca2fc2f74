"""What bounds small grids: c3 1/8 shard and c2, kernel ms with the duplicate path off (PILOT_OT_DEBUG=512), one workgroup per CU
(16), both (528); plus the update counts of the duplicate and the longest other pairs."""
import os, sys, time
__import__("sys").path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))   # tools/switches.py
import switches
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import make_problem, CONFIGS
for cfg, step in (("c3", 8), ("c3", 1), ("c2", 1)):
    P, M = make_problem(**CONFIGS[cfg])
    N = P.shape[0]
    pl = engine.DevicePlan(P, M); pl.enable_timing(True)
    for _ in range(50): pl.run(0.1)
    pl.sync()
    it = pl.fetch()[1]["iters"]
    d = np.diag(it)
    print("%s: duplicates need %d .. %d updates (mean %.0f); other pairs max %d, p99.9 %d" % (cfg, d.min(), d.max(), d.mean(), (it - np.diag(d)).max(), np.percentile(it, 99.9)))
    for dbg in ("0", "512", "16", "528"):
        switches.set("PILOT_OT_DEBUG", dbg)
        for _ in range(5): pl.run(0.1, row_begin=0, row_step=step)
        pl.sync()
        for _ in range(20): pl.run(0.1, row_begin=0, row_step=step)
        pl.sync()
        a, b = pl.kernel_times_ms(20)
        print("  %s rows 0::%d  PILOT_OT_DEBUG=%-3s kernel %.4f ms" % (cfg, step, dbg, a.mean()), flush=True)
    switches.set("PILOT_OT_DEBUG", None)
    pl.close()
