"""Exact OT: would a better start than the diagonal save augmentations? (VERDICT r04 #1b; numpy prototype of the kernels' algorithm, CPU)

Potentials carried from pair (i, j) to (i, j + 1): dual-feasible for any pair, and the previous optimal support is a set of tight
arcs, so a greedy flow on it keeps complementary slackness -- against the fresh start (row minima + min(a_i, b_i) on the diagonal).
A greedy row-minimum start is not on the table: with the zero-diagonal costs of this path the only tight arcs under the initial
potentials ARE the diagonal.  Result (profiles/r05/emd_start_probe.txt): the reference test's cohort -20 % augmentations / -14 %
steps, the c3 shape +16 % steps -- below the 30 % that would pay for the dependency between consecutive pairs: not built."""
import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import GOLDEN_REAL, load_golden
from oracle import oracle as O
INF = float("inf")
def solve(a, b, M, stats, warm=None):
    K = len(a)
    a = a.astype(float).copy(); b = b.astype(float).copy()
    sa, sb = a.sum(), b.sum(); b *= sa / sb
    tol = 1e-15 * (sa if sa > 0 else 1.0)
    F = np.zeros((K, K)); ra = a.copy(); rb = b.copy()
    if warm is None:
        pu = M.min(1).copy(); pv = np.zeros(K)
        for i in range(K):
            if M[i, i] - pu[i] == 0.0:
                f = min(ra[i], rb[i])
                if f > 0: F[i, i] = f; ra[i] -= f; rb[i] -= f
    else:
        pu, pv, supp = warm
        pu = pu.copy(); pv = pv.copy()
        # greedy on the previous support (tight arcs), diagonal first
        arcs = [(i, i) for i in range(K) if supp[i, i]] + [(i, j) for i in range(K) for j in range(K) if supp[i, j] and i != j]
        for (i, j) in arcs:
            f = min(ra[i], rb[j])
            if f > 0: F[i, j] += f; ra[i] -= f; rb[j] -= f
    prev_src = None; A = None; Apar = None
    while True:
        src = ra > tol
        if not src.any(): break
        stats["search"] += 1
        if prev_src is None or (src != prev_src).any():
            A = np.full(K, INF); Apar = np.full(K, -1)
            for i in np.nonzero(src)[0]:
                v = M[i] - pu[i]; lt = v < A; A[lt] = v[lt]; Apar[lt] = i
            prev_src = src.copy()
        dC = A.copy(); parC = Apar.copy(); closed = np.zeros(K, bool)
        reached = src.copy(); parR = np.full(K, -1); puN = pu.copy()
        step_bd = 0.0; stale = False; dstar = 0.0; exhausted = False
        while True:
            stats["step"] += 1
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
                    if jb < 0: break
                    j = jb
                s = hops[-1][0]
                delta = min(rb[t], ra[s])
                for (i, j, jb) in hops:
                    if jb >= 0: delta = min(delta, F[i, jb])
                emptied = False
                for (i, j, jb) in hops:
                    F[i, j] += delta
                    if jb >= 0:
                        F[i, jb] -= delta
                        if F[i, jb] == 0: emptied = True
                ra[s] -= delta; rb[t] -= delta
                stats["aug"] += 1; stats["hops"] += len(hops)
                if emptied or not (ra[s] > tol) or rb[t] > 0: stale = True
                if stale: broke = True; break
            if broke: dstar = bd; break
            hit = (F[:, tie] > 0).any(1) & ~reached
            rows = np.nonzero(hit)[0]
            for i in rows:
                js = np.nonzero(tie & (F[i] > 0))[0]
                parR[i] = js[0]; puN[i] = pu[i] - bd
            reached |= hit
            for i in rows:
                stats["relax"] += 1
                nd = M[i] - (pu[i] - bd)
                lt = (nd < dC) & ~closed
                dC[lt] = nd[lt]; parC[lt] = i
        if exhausted: break
        pu = np.where(reached, puN, pu - dstar)
        fC = np.where(closed, np.maximum(dC - pv, 0.0), INF)
        pv = pv + np.minimum(fC, dstar)
    return (F * M).sum(), (pu, pv, F > 0)

def run(P, M, name, rows=6, ncol=60):
    rng = np.random.default_rng(0)
    for mode in ["fresh", "carry"]:
        st = dict(search=0, step=0, aug=0, hops=0, relax=0); n = 0; err = 0
        for i in rng.integers(0, P.shape[0], rows):
            warm = None
            for j in range(min(ncol, P.shape[0])):
                c, w = solve(P[i], P[j], M, st, warm if mode == "carry" else None)
                warm = w; n += 1
                err = max(err, abs(c - O.emd2(P[i], P[j], M)))
        print(name, mode, {k: round(v / n, 1) for k, v in st.items()}, "err %.1e" % err)
g = load_golden(GOLDEN_REAL)
run(g["proportions"], g["cost"] / g["cost"].max(), "kidney")
from pilot_amd.synthetic import make_problem
P, M = make_problem(600, 50, 8, seed=50, cells_per_patient=200)
run(P, M, "c3like", rows=3, ncol=30)
