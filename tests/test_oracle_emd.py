"""The exact-OT oracle against scipy's HiGHS LP solver (the LP optimum value is unique).  CPU only."""
import numpy as np
import pytest
from scipy.optimize import linprog

from conftest import GOLDEN_CASES, load_golden
from oracle import oracle as O
from pilot_amd.synthetic import CONFIGS, make_problem


def lp_emd(a, b, M):
    na, nb = len(a), len(b)
    A = np.zeros((na + nb, na * nb))
    for i in range(na):
        A[i, i * nb:(i + 1) * nb] = 1
    for j in range(nb):
        A[na + j, j::nb] = 1
    b2 = b * (a.sum() / b.sum())
    res = linprog(M.ravel(), A_eq=A, b_eq=np.concatenate([a, b2]), bounds=(0, None), method="highs")
    assert res.status == 0
    return res.fun


def test_against_highs_on_pilot_shaped_pairs():
    P, M = make_problem(**CONFIGS["c2"])
    rng = np.random.default_rng(1)
    for _ in range(15):
        i, j = rng.integers(0, P.shape[0], 2)
        assert abs(O.emd2(P[i], P[j], M) - lp_emd(P[i], P[j], M)) < 1e-10


def test_against_highs_on_ragged_and_degenerate_problems():
    rng = np.random.default_rng(2)
    for na, nb in [(1, 1), (1, 5), (4, 1), (3, 7), (8, 8), (12, 5)]:
        a = rng.dirichlet(np.ones(na)); b = rng.dirichlet(np.ones(nb))
        M = rng.random((na, nb))
        assert abs(O.emd2(a, b, M) - lp_emd(a, b, M)) < 1e-10
    # zero-mass bins (POT drops them), unequal total mass (POT rescales b)
    a = np.array([0.5, 0.0, 0.5, 0.0]); b = np.array([0.0, 0.3, 0.3, 0.0]) 
    M = rng.random((4, 4))
    b_eff = np.where(b > 0, b, 0)
    assert abs(O.emd2(a, b, M) - lp_emd(a, b_eff, M)) < 1e-10


def test_identical_histograms_cost_zero_and_plan_is_feasible():
    P, M = make_problem(**CONFIGS["c1"])
    for i in range(5):
        assert abs(O.emd2(P[i], P[i], M)) < 1e-15
    val, G = O.emd2(P[0], P[1], M, return_plan=True)
    np.testing.assert_allclose(G.sum(1), P[0], atol=1e-14)
    np.testing.assert_allclose(G.sum(0), P[1], atol=1e-14)
    assert (G >= 0).all() and abs((G * M).sum() - val) < 1e-15
    assert (G > 0).sum() <= 2 * len(P[0]) - 1 + 2      # (near-)basic solution


def test_grid_is_symmetric_with_zero_diagonal_for_pdist_costs():
    P, M = make_problem(**CONFIGS["c1"])
    E = O.emd_grid(P, M)
    assert np.abs(E - E.T).max() < 1e-14 and np.abs(np.diag(E)).max() < 1e-15
    # triangle inequality holds for a metric-like ground cost only approximately; sanity: bounded by max M
    assert E.max() <= M.max() + 1e-12


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_unreg_matrix_is_reproduced(name):
    g = load_golden(name)
    P = g["proportions"]
    M = g["cost"] / g["cost"].max()
    E = O.emd_grid(P, M, n_threads=4)
    np.testing.assert_allclose(E, g["emd_unreg"], rtol=0, atol=1e-14)
    np.testing.assert_array_equal(g["emd_unreg_df"], g["emd_unreg"].T)


def test_fast_solver_of_the_cpu_baseline_returns_the_same_lp_value():
    """oracle.emd_grid(fast=True) -- the HIP kernel's algorithm on one CPU thread, what `bench.py --mode emd` times as its CPU
    baseline -- against the plain oracle solver and against HiGHS: random non-symmetric costs, sparse histograms, ties."""
    from scipy.optimize import linprog
    rng = np.random.default_rng(11)
    for K in (1, 2, 3, 8, 21, 50):
        P = rng.random((7, K))
        P[rng.random((7, K)) < 0.35] = 0.0
        P[:, 0] += 0.05
        P /= P.sum(1, keepdims=True)
        M = np.round(rng.random((K, K)) * 8) / 8 if K % 2 else rng.random((K, K))       # (quantised: many ties)
        slow, fast = O.emd_grid(P, M), O.emd_grid(P, M, fast=True)
        assert np.abs(slow - fast).max() <= 1e-13
        if K <= 21:
            A = np.zeros((2 * K, K * K))
            for i in range(K):
                A[i, i * K:(i + 1) * K] = 1.0
                A[K + i, i::K] = 1.0
            res = linprog(M.ravel(), A_eq=A, b_eq=np.concatenate([P[0], P[3]]), bounds=(0, None), method="highs")
            assert abs(res.fun - fast[0, 3]) <= 1e-10
    Pc, Mc = make_problem(**CONFIGS["c2"])
    assert np.abs(O.emd_grid(Pc, Mc, row_end=3) - O.emd_grid(Pc, Mc, row_end=3, fast=True)).max() <= 1e-13


def test_network_simplex_leg_returns_the_same_lp_value():
    """oracle.emd_grid(fast="ns") / oracle.emd2_ns -- a network simplex (spanning-tree basis, block-search pricing), the CPU
    baseline of `bench.py --mode emd` -- against the successive-shortest-path oracle and HiGHS: rectangular problems, empty
    bins (degenerate pivots), quantised costs (ties), PILOT-shaped pairs."""
    rng = np.random.default_rng(5)
    for _ in range(200):
        na, nb = rng.integers(1, 13, 2)
        a, b, M = rng.random(na), rng.random(nb), rng.random((na, nb))
        if rng.random() < 0.4:
            M = np.round(M * 4) / 4
        if rng.random() < 0.4 and na > 1:
            a[rng.integers(0, na)] = 0.0
        if rng.random() < 0.4 and nb > 1:
            b[rng.integers(0, nb)] = 0.0
        v, pivots = O.emd2_ns(a, b, M, return_pivots=True)
        assert abs(v - O.emd2(a, b, M)) <= 1e-12 and pivots < 200 * (na + nb + 1)
    for na, nb in [(3, 7), (8, 8), (12, 5)]:
        a = rng.dirichlet(np.ones(na)); b = rng.dirichlet(np.ones(nb)); M = rng.random((na, nb))
        assert abs(O.emd2_ns(a, b, M) - lp_emd(a, b, M)) < 1e-10
    for cfg in ("c1", "c2"):
        P, M = make_problem(**CONFIGS[cfg])
        assert np.abs(O.emd_grid(P, M, row_end=4, fast="ns") - O.emd_grid(P, M, row_end=4)).max() <= 1e-13
    P, M = make_problem(**CONFIGS["c3"])
    assert np.abs(O.emd_grid(P, M, row_end=1, fast="ns", n_threads=4) - O.emd_grid(P, M, row_end=1, fast=True)).max() <= 1e-13
