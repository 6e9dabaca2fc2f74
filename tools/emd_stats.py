"""Per-pair numbers an exact-OT grid reports in n_aug (augmentations, or the event / tick counter of an EMD_STAT / EMD_PROF build:
tools/emd_stat_builds.sh, tools/emd_prof_builds.sh), mean and maximum over the solved pairs of a config."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilot_amd import engine
from pilot_amd.synthetic import CONFIGS, make_problem
P, M = make_problem(**CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"])
E, info = engine.emd_grid(P, M, return_info=True)
n = info["n_aug"][np.triu_indices(P.shape[0], 1)]
print("counter per pair (upper triangle): mean %.1f  median %.1f  p99 %.1f  max %d" % (n.mean(), np.median(n), np.percentile(n, 99), n.max()))
t = time.perf_counter()
for _ in range(3): engine.emd_grid(P, M)
print("%.1f ms per matrix" % ((time.perf_counter() - t) / 3 * 1e3))
