#!/usr/bin/env python3
"""Where the time of tl.wasserstein_distance goes (GPU box): wall-clock stamps around the stages, 1.8 M cells (c3), or
`real`: the reference test's own cohort (Kidney_IgAN_G, 24 227 glomeruli, 634 patients x 14 clusters; fixture under tests/golden)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PILOT_AMD_NO_RESULTS_DIR", "1")
from pilot_amd import tl, engine
from pilot_amd.synthetic import make_cells, CONFIGS
REAL = len(sys.argv) > 1 and sys.argv[1] == "real"
if REAL:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import GOLDEN_REAL, golden_adata, load_golden
    ad, cell_col = golden_adata(load_golden(GOLDEN_REAL))
    KW = dict(clusters_col=cell_col, sample_col="sampleID", status="status", data_type="Pathomics")
    REPS = 20
else:
    c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
    ad = make_cells(c["n_patients"], c["n_types"], c["n_dims"], c["seed"], c["cells_per_patient"])
    cell_col = "cell_types"
    KW = dict(emb_matrix="X_pca")
    REPS = 4
for categorical in ((None,) if REAL else (False, True)):
    if categorical:
        for col in ad.obs.columns:
            ad.obs[col] = ad.obs[col].astype("category")
    for mode in ("reg", "unreg"):
        best = 1e9
        for rep in range(REPS):
            ad.uns = {}
            t0 = time.perf_counter()
            tl.wasserstein_distance(ad, regularized=mode, reg=0.1, **KW)
            best = min(best, time.perf_counter() - t0)
        print("%s labels, %s: %.5f s end to end (best of %d)" % ("as stored" if REAL else ("categorical" if categorical else "object"), mode, best, REPS))
# stage stamps (object labels; the real cohort: labels as stored), each stage best of 5
import pandas as pd
if REAL:
    X = pd.DataFrame(ad[:, list(ad.var_names)].X).to_numpy()
else:
    ad.obs = ad.obs.astype(object)
    X = ad.obsm["X_pca"]
def stamp(name, fn, reps=5):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t)
    print("  %-58s %.5f s" % (name, best)); return r
if REAL:
    stamp("data frame (adata[:, var_names].X)", lambda: pd.DataFrame(ad[:, list(ad.var_names)].X, columns=list(ad.var_names)))
print("  host: %d cores, transparent huge pages: %s" % (os.cpu_count(), open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
                                                      if os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled") else "n/a"))
annot = stamp("annot frame (obs[[3 columns]])", lambda: tl._annot_frame(ad.obs, cell_col, "sampleID", "status"))
cc = stamp("number cell_type (native pass)", lambda: tl._first_appearance_codes(annot["cell_type"]))
sc = stamp("number sampleID (native pass)", lambda: tl._first_appearance_codes(annot["sampleID"]))
if not REAL:
    for col in ("cell_type", "sampleID"):
        cat = annot[col].astype("category")
        stamp("number %s, categorical" % col, lambda: tl._first_appearance_codes(cat))
up = stamp("embedding upload (H2D, here on this thread)", lambda: (lambda u: (u.thread.join() if u.thread else None, u)[1])(engine.EmbeddingUpload(X)), reps=1)
def private_copy(n_threads, huge):
    import threading
    own = tl._empty_like_huge(X) if huge else np.empty_like(X)
    b = np.linspace(0, X.shape[0], n_threads + 1).astype(np.int64)
    th = [threading.Thread(target=np.copyto, args=(own[i:j], X[i:j])) for i, j in zip(b[:-1], b[1:])]
    [t.start() for t in th]; [t.join() for t in th]
for nt in (1, 4, 8):
    stamp("private copy of the embedding, %d threads" % nt, lambda: private_copy(nt, False))
    stamp("private copy of the embedding, %d threads, madvise(HUGEPAGE)" % nt, lambda: private_copy(nt, True))
pr = stamp("proportions + first rows (device)", lambda: engine.proportions_and_first_rows(cc[0], sc[0], len(sc[1]), len(cc[1])))
cen = stamp("medians (device, embedding resident)", lambda: up.medians(cc[0], len(cc[1])))
stamp("pre-pass in one call (proportions + first rows + medians)", lambda: up.prepass(cc[0], sc[0], len(sc[1]), len(cc[1])))
M = stamp("pdist (device)", lambda: engine.pdist_square(cen))
E = stamp("sinkhorn grid (host arrays in / out)", lambda: engine.sinkhorn_grid(pr[0], M / M.max(), 0.1))
stamp("exact grid (host arrays in / out)", lambda: engine.emd_grid(pr[0], M / M.max()))
stamp("EMD frame", lambda: tl._emd_frame(E, list(sc[1])))
stamp("cost frame", lambda: tl._cost_frame(M, cc[1]))
stamp("proportions dict", lambda: dict(zip(sc[1], np.array(pr[0], copy=True))))
st = annot["status"].to_numpy()
stamp("real labels", lambda: [st[int(r)] for r in pr[1]])
up.close()
