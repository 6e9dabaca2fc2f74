"""Exact OT: warm restarts of the shortest-path tree (numpy prototype of the kernels' algorithm, CPU).

When an augmentation dries its root or empties an arc the kernels bring the potentials up to date and search again from the
sources; the new search re-walks, level by level at label 0, the part of the old tree that is still valid.  With the path masks
of emd_multi_kernels.hpp the valid part can be KEPT (a label is invalid iff its path contains a dried root or the row of an
emptied arc), the labels of the open columns rebuilt from the kept rows, and the search goes on.  Same LP values; Dijkstra steps
-40 % (reference test's cohort 152 -> 91 per pair, c3 shape 240 -> 133), row relaxations unchanged (every kept row relaxes the
reopened columns again).  In the four-pairs-per-wave kernel the steps are a third of the instructions, in the one-pair kernel
two restarts' bookkeeping cost what the saved steps do: estimated -12 % / -4 %, not built (profiles/r05/ab_experiments.md)."""
import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import GOLDEN_REAL, load_golden
from oracle import oracle as O
INF = float("inf")
def solve(a, b, M, st, warm=True, always_rebuild=False):
    K = len(a)
    a = a.astype(float).copy(); b = b.astype(float).copy()
    sa, sb = a.sum(), b.sum(); b *= sa / sb
    tol = 1e-15 * (sa if sa > 0 else 1.0)
    F = np.zeros((K, K)); ra = a.copy(); rb = b.copy()
    pu = M.min(1).copy(); pv = np.zeros(K)
    for i in range(K):
        if M[i, i] - pu[i] == 0.0:
            f = min(ra[i], rb[i])
            if f > 0: F[i, i] = f; ra[i] -= f; rb[i] -= f
    prev_src = None; A = None; Apar = None
    keep = None      # state carried into a warm restart
    guard = 0
    while True:
        guard += 1
        assert guard < 100 * K
        src = ra > tol
        if not src.any(): break
        st["search"] += 1
        if prev_src is None or (src != prev_src).any():
            A = np.full(K, INF); Apar = np.full(K, -1)
            for i in np.nonzero(src)[0]:
                v = M[i] - pu[i]; lt = v < A; A[lt] = v[lt]; Apar[lt] = i; st["srcrelax"] += 1
            prev_src = src.copy()
        pending_targets = np.zeros(K, bool)
        if keep is None:
            dC = A.copy(); parC = Apar.copy(); closed = np.zeros(K, bool)
            reached = src.copy(); parR = np.full(K, -1); puN = pu.copy()
            pmRr = np.zeros((K, K), bool); pmRc = np.zeros((K, K), bool)   # path of row i: rows / cols
            for i in range(K): pmRr[i, i] = True
            pmCr = np.zeros((K, K), bool); pmCc = np.zeros((K, K), bool)
            for c in range(K):
                if Apar[c] >= 0: pmCr[c, Apar[c]] = True
                pmCc[c, c] = True
            pend_rows = []
        else:
            (dC, parC, closed, reached, parR, pmRr, pmRc, pmCr, pmCc, bad) = keep
            st["warm"] += 1
            # invalid: path contains a bad row
            inv_r = reached & (pmRr[:, bad].any(1))
            inv_r |= reached & ~src & (parR < 0)          # a dried source that is still marked as a root
            # rows whose root dried: their path contains that source row (in bad)
            reached = reached & ~inv_r
            reached |= src
            lab = np.isfinite(dC)
            inv_c = lab & (pmCr[:, bad].any(1))
            closed = closed & ~inv_c
            puN = pu.copy()                               # new potentials already applied: every kept row is at distance 0
            for i in np.nonzero(src)[0]:
                parR[i] = -1; pmRr[i] = False; pmRc[i] = False; pmRr[i, i] = True
            need = (~closed) & inv_c                       # open columns whose label came from an invalid row
            if need.any() or always_rebuild:
                st["rebuild"] += 1
                # labels of open columns again from the kept reached rows (sources through A)
                for c in np.nonzero(~closed)[0]:
                    dC[c] = A[c]; parC[c] = Apar[c]; pmCr[c] = False; pmCc[c] = False; pmCc[c, c] = True
                    if Apar[c] >= 0: pmCr[c, Apar[c]] = True
                pend_rows = [i for i in np.nonzero(reached & ~src)[0]]
            else:
                pend_rows = []
            # closed columns that still have demand are targets again
            pending_targets = closed & (rb > 0)
            # rows that ship to a kept closed column and are not reached: reached at distance 0
            hit = (F[:, closed] > 0).any(1) & ~reached
            for i in np.nonzero(hit)[0]:
                js = np.nonzero(closed & (F[i] > 0))[0]
                parR[i] = js[0]; puN[i] = pu[i]
                pmRr[i] = pmCr[js[0]]; pmRc[i] = pmCc[js[0]]; pmRr[i, i] = True
                pend_rows.append(i)
            reached |= hit
            keep = None
        step_bd = 0.0; dstar = 0.0; exhausted = False
        bad = np.zeros(K, bool)
        first = True
        while True:
            for i in pend_rows:
                st["relax"] += 1
                nd = M[i] - puN[i]
                lt = (nd < dC) & ~closed
                dC[lt] = nd[lt]; parC[lt] = i
                pmCr[lt] = pmRr[i]; pmCc[lt] = pmRc[i]
                for c in np.nonzero(lt)[0]: pmCc[c, c] = True
            pend_rows = []
            if first and pending_targets.any():
                tie = pending_targets; bd = 0.0; first = False
            else:
                first = False
                st["step"] += 1
                cur = np.maximum(dC - pv, step_bd); cur[closed] = INF
                bd = cur.min()
                if bd == INF: exhausted = True; break
                step_bd = bd
                tie = (cur == bd); closed |= tie
            broke = False
            for t in np.nonzero(tie & (rb > 0))[0]:
                hops = []; j = t
                while True:
                    i = parC[j]; jb = parR[i]; hops.append((i, j, jb))
                    assert pmCr[t, i] and pmCc[t, j]
                    if jb < 0: break
                    j = jb
                assert pmCr[t].sum() == len(hops) and pmCc[t].sum() == len(hops)
                s = hops[-1][0]
                delta = min(rb[t], ra[s])
                for (i, j, jb) in hops:
                    if jb >= 0: delta = min(delta, F[i, jb])
                stale = False
                for (i, j, jb) in hops:
                    F[i, j] += delta
                    if jb >= 0:
                        F[i, jb] -= delta
                        if F[i, jb] == 0: stale = True; bad[i] = True
                ra[s] -= delta; rb[t] -= delta
                st["aug"] += 1
                if not (ra[s] > tol): stale = True; bad[s] = True
                if rb[t] > 0: stale = True
                if stale: broke = True; break
            if broke: dstar = bd; break
            hit = (F[:, tie] > 0).any(1) & ~reached
            for i in np.nonzero(hit)[0]:
                js = np.nonzero(tie & (F[i] > 0))[0]
                parR[i] = js[0]; puN[i] = pu[i] - bd
                pmRr[i] = pmCr[js[0]]; pmRc[i] = pmCc[js[0]]; pmRr[i, i] = True
                pend_rows.append(i)
            reached |= hit
        if exhausted: break
        pu = np.where(reached, puN, pu - dstar)
        fC = np.where(closed, np.maximum(dC - pv, 0.0), INF)
        pv = pv + np.minimum(fC, dstar)
        if warm:
            keep = (dC, parC, closed, reached, parR, pmRr, pmRc, pmCr, pmCc, bad)
    return (F * M).sum()

def run(P, M, name, n):
    rng = np.random.default_rng(0)
    prs = [tuple(rng.integers(0, P.shape[0], 2)) for _ in range(n)]
    for mode in [dict(warm=False), dict(warm=True), dict(warm=True, always_rebuild=True)]:
        st = dict(search=0, step=0, aug=0, relax=0, srcrelax=0, warm=0, rebuild=0); err = 0
        for (i, j) in prs:
            c = solve(P[i], P[j], M, st, **mode)
            err = max(err, abs(c - O.emd2(P[i], P[j], M)))
        print(name, mode, {k: round(v / n, 1) for k, v in st.items()}, "err %.1e" % err, flush=True)
g = load_golden(GOLDEN_REAL)
run(g["proportions"], g["cost"] / g["cost"].max(), "kidney", 300)
from pilot_amd.synthetic import make_problem
P, M = make_problem(600, 50, 8, seed=50, cells_per_patient=200)
run(P, M, "c3like", 100)
P, M = make_problem(600, 12, 8, seed=12, cells_per_patient=200)
run(P, M, "K12", 300)
