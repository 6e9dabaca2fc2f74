#!/usr/bin/env python3
"""End-to-end wall time of tl.wasserstein_distance on a cell-level synthetic cohort (GPU box), with the
per-stage breakdown; the CPU oracle's restatement of the reference's pandas pre-pass is timed beside it."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PILOT_AMD_NO_RESULTS_DIR", "1")
from oracle import oracle as O
from pilot_amd import tl, engine
from pilot_amd.synthetic import make_cells, CONFIGS
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
c = CONFIGS[cfg]
t = time.perf_counter(); ad = make_cells(c["n_patients"], c["n_types"], c["n_dims"], c["seed"], c["cells_per_patient"]); print("synthetic cohort: %d cells x %d dims (%.1f s to generate)" % (ad.X.shape[0], ad.X.shape[1], time.perf_counter() - t))
for mode in ("reg", "unreg"):
    ad.uns = {}
    tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized=mode, reg=0.1)      # warm-up (library load, plan)
    ad.uns = {}
    t0 = time.perf_counter()
    tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized=mode, reg=0.1)
    t1 = time.perf_counter()
    print("tl.wasserstein_distance(%s): %.3f s end to end" % (mode, t1 - t0))
data, annot = tl.extract_data_anno_scRNA_from_h5ad(ad, "X_pca", "cell_types", "sampleID", "status")
for name, fn in (("extract frames", lambda: tl.extract_data_anno_scRNA_from_h5ad(ad, "X_pca", "cell_types", "sampleID", "status")),
                 ("Cluster_Representations (factorize + device histogram)", lambda: tl.Cluster_Representations(annot)),
                 ("cost_matrix (device medians + pdist, incl. H2D of the embedding)", lambda: tl.cost_matrix(annot, data)),
                 ("return_real_labels", lambda: tl.return_real_labels(annot))):
    t0 = time.perf_counter(); fn(); print("  %-70s %.3f s" % (name, time.perf_counter() - t0))
if ad.X.shape[0] <= 400000:
    t0 = time.perf_counter(); O.cluster_representations(annot["cell_type"], annot["sampleID"]); t1 = time.perf_counter()
    O.cost_matrix(data, annot["cell_type"]); t2 = time.perf_counter()
    print("  CPU oracle (pandas restatement of the reference): Cluster_Representations %.3f s, cost_matrix %.3f s" % (t1 - t0, t2 - t1))
