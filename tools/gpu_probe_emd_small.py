import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from pilot_amd import engine, _lib
from pilot_amd.synthetic import make_problem, CONFIGS
import ctypes
P, M = make_problem(**CONFIGS["c1"])
N, K = P.shape
emd = np.zeros((N, N)); n_aug = np.zeros((N, N), dtype=np.int32)
rc = _lib.load().pilot_ot_emd_grid(_lib.dptr(P), N, K, _lib.dptr(M), 0, 0, N, 1, _lib.dptr(emd), _lib.iptr(n_aug))
print("rc", rc, flush=True)
Eo = O.emd_grid(P, M)
print("n_aug min/max", n_aug.min(), n_aug.max(), "trip codes", np.unique((-n_aug[n_aug < 0]) % 8), "n tripped", (n_aug < 0).sum(), flush=True)
ok = n_aug >= 0
print("max|d| on ok pairs", np.abs(emd - Eo)[ok].max() if ok.any() else None, flush=True)
print(n_aug[:4, :8])
