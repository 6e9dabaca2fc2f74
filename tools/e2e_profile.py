#!/usr/bin/env python3
"""Where the time of tl.wasserstein_distance goes (GPU box): wall-clock stamps around the stages, 1.8 M cells (c3)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PILOT_AMD_NO_RESULTS_DIR", "1")
from pilot_amd import tl, engine
from pilot_amd.synthetic import make_cells, CONFIGS
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
ad = make_cells(c["n_patients"], c["n_types"], c["n_dims"], c["seed"], c["cells_per_patient"])
for categorical in (False, True):
    if categorical:
        for col in ad.obs.columns:
            ad.obs[col] = ad.obs[col].astype("category")
    for mode in ("reg", "unreg"):
        best = 1e9
        for rep in range(4):
            ad.uns = {}
            t0 = time.perf_counter()
            tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized=mode, reg=0.1)
            best = min(best, time.perf_counter() - t0)
        print("%s labels, %s: %.4f s end to end (best of 4)" % ("categorical" if categorical else "object", mode, best))
# stage stamps (object labels)
import pandas as pd
ad.obs = ad.obs.astype(object)
X = ad.obsm["X_pca"]
def stamp(name, fn):
    t = time.perf_counter(); r = fn(); print("  %-58s %.4f s" % (name, time.perf_counter() - t)); return r
annot = stamp("annot frame (obs[[3 columns]])", lambda: tl._annot_frame(ad.obs, "cell_types", "sampleID", "status"))
cc = stamp("factorise cell_type", lambda: tl._first_appearance_codes(annot["cell_type"]))
sc = stamp("factorise sampleID", lambda: tl._first_appearance_codes(annot["sampleID"]))
up = stamp("embedding upload (216 MB H2D, here on this thread)", lambda: (lambda u: (u.thread.join(), u)[1])(engine.EmbeddingUpload(X)))
stamp("private copy of the embedding, 4 threads", lambda: tl.np.copyto(np.empty_like(X), X))
pr = stamp("proportions + first rows (device)", lambda: engine.proportions_and_first_rows(cc[0], sc[0], len(sc[1]), len(cc[1])))
cen = stamp("medians (device, embedding resident)", lambda: up.medians(cc[0], len(cc[1])))
M = stamp("pdist (device)", lambda: engine.pdist_square(cen))
E = stamp("sinkhorn grid (host arrays in / out)", lambda: engine.sinkhorn_grid(pr[0], M / M.max(), 0.1))
stamp("EMD frame (from_dict(EMD).T ...)", lambda: tl.wasserstein_d({i: pr[0][i] for i in range(3)}, M / M.max(), "reg"))
up.close()
