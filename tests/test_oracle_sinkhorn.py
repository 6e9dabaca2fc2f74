"""The Sinkhorn oracle (oracle/pilot_oracle.c) against independent evidence.  CPU only.

POT itself cannot be run here ("parity unpinned"); what is pinned: POT's own docstring example, converged values
against an independent log-domain fixed point (unique entropic optimum), POT's documented control flow
(check every 20 updates, v-first, cap), and the committed golden numbers."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden
from oracle import oracle as O
from pilot_amd.synthetic import CONFIGS, make_problem


def test_pot_docstring_known_answers():
    """The only numbers POT itself publishes for this path: the example in the docstrings of ot.sinkhorn /
    ot.bregman.sinkhorn_stabilized / ot.sinkhorn2 (a = b = [.5, .5], M = [[0, 1], [1, 0]], reg = 1: plan
    [[0.36552929, 0.13447071], [0.13447071, 0.36552929]], sinkhorn2 = 0.26894142) and of ot.emd / ot.emd2 on the same
    inputs (plan diag(.5, .5), cost 0.0).  Printed to 8 digits there."""
    a = np.array([0.5, 0.5])
    M = np.array([[0.0, 1.0], [1.0, 0.0]])
    val, info = O.sinkhorn2(a, a, M, 1.0, return_info=True)
    assert info["flags"] & O.FLAG_CONVERGED
    assert abs(val - 0.26894142) < 5e-9
    assert abs(val - 2 * 0.13447071) < 2e-8
    assert O.emd2(a, a, M) == 0.0
    # the documented asymmetric case of ot.emd2's tests (test_ot.py::test_emd_1d_emd2_1d analogue on a line): cost of
    # moving [1, 0] to [0, 1] with unit distance is exactly 1
    assert O.emd2(np.array([1.0, 0.0]), np.array([0.0, 1.0]), M) == 1.0


def test_converged_pairs_match_independent_logdomain_solver():
    P, M = make_problem(**CONFIGS["c2"])
    rng = np.random.default_rng(0)
    for reg in (1.0, 0.1):
        for _ in range(12):
            i, j = rng.integers(0, P.shape[0], 2)
            val, info = O.sinkhorn2(P[i], P[j], M, reg, return_info=True)
            assert info["flags"] & O.FLAG_CONVERGED
            ref = O.sinkhorn_log_converged(P[i], P[j], M, reg)
            # stopped at marginal error <= 1e-9  ->  value within a few 1e-9 of the fixed point
            assert abs(val - ref) < 2e-8, (reg, i, j, val, ref)


def test_iteration_counts_follow_pot_check_period():
    P, M = make_problem(**CONFIGS["c2"])
    _, info = O.sinkhorn_grid(P, M, 0.1, row_step=25, return_info=True)
    it = info["iters"]
    conv = (info["flags"] & O.FLAG_CONVERGED) > 0
    # err is evaluated when ii % 20 == 0 -> converged pairs stop after 20k+1 updates
    assert np.all(it[conv] % 20 == 1)
    assert np.all(it[~conv] == 1000)          # POT >= 0.8: `for ii in range(numItermax)`
    assert conv.mean() > 0.9


def test_legacy_loop_runs_one_more_update_on_capped_pairs():
    P, M = make_problem(**CONFIGS["c1"])
    a, b = P[0], P[3]
    v_new, i_new = O.sinkhorn2(a, b, M, 0.002, return_info=True)
    v_old, i_old = O.sinkhorn2(a, b, M, 0.002, legacy_loop=True, return_info=True)
    if not (i_new["flags"] & O.FLAG_CONVERGED):
        assert i_new["iters"] == 1000 and i_old["iters"] == 1001
    assert abs(v_new - v_old) < 1e-3


def test_value_is_transport_cost_without_entropy_and_diag_nonzero():
    P, M = make_problem(**CONFIGS["c1"])
    E = O.sinkhorn_grid(P, M, 0.1)
    assert E.shape == (20, 20)
    assert np.all(np.diag(E) > 1e-3)           # entropic plan of (a, a) is not the identity
    assert np.all(E >= 0) and np.all(E <= M.max() + 1e-12)
    assert np.abs(E - E.T).max() < 1e-7        # converged pairs are symmetric up to stopThr


def test_absorption_matches_plain_scaling_in_exact_arithmetic():
    """tau-absorption is purely numerical: with tau=inf (never absorb) the value after the same
    number of updates is the same, except when the absorption lands on the final update."""
    P, M = make_problem(**CONFIGS["c1"])
    rng = np.random.default_rng(3)
    n_abs = 0
    for _ in range(20):
        i, j = rng.integers(0, 20, 2)
        v1, i1 = O.sinkhorn2(P[i], P[j], M, 0.02, return_info=True)
        v2, i2 = O.sinkhorn2(P[i], P[j], M, 0.02, tau=1e300, return_info=True)
        n_abs += i1["n_absorb"]
        if i1["flags"] & O.FLAG_ABSORB_ON_LAST:
            continue
        if i1["iters"] == i2["iters"]:
            assert abs(v1 - v2) < 1e-9 * max(1.0, abs(v1)) + 1e-12
    assert n_abs > 0, "test did not exercise the absorption branch"


def test_absorb_on_final_update_scales_plan_by_one_over_k2():
    """POT resets u, v to 1/K (not 1) when absorbing; if that happens on the very last update the
    returned plan is scaled by 1/K^2.  Restated faithfully (see pilot_oracle.c)."""
    P, M = make_problem(**CONFIGS["c3"])
    i, j = 275, 21          # found by scanning rows 0,25,.. of c3 at reg=0.01 (3 of 14400 pairs hit it)
    v, inf = O.sinkhorn2(P[i], P[j], M, 0.01, return_info=True)
    assert inf["flags"] & O.FLAG_ABSORB_ON_LAST and inf["iters"] == 1000
    v_one_more = O.sinkhorn2(P[i], P[j], M, 0.01, numItermax=1001)
    assert v < v_one_more / 100.0 and abs(v * 2500 - v_one_more) < 1e-3


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_reg_matrix_is_reproduced(name):
    g = load_golden(name)
    P = g["proportions"]
    M = g["cost"] / g["cost"].max()
    E = O.sinkhorn_grid(P, M, float(g["reg"]), n_threads=4)
    np.testing.assert_allclose(E, g["emd_reg"], rtol=0, atol=1e-13)
    # the reference's DataFrame is from_dict(EMD).T  (Trajectory.py:518)
    np.testing.assert_array_equal(g["emd_reg_df"], g["emd_reg"].T)


def test_grid_row_selection_and_threads_are_consistent():
    P, M = make_problem(**CONFIGS["c1"])
    full = O.sinkhorn_grid(P, M, 0.1)
    part = O.sinkhorn_grid(P, M, 0.1, row_begin=1, row_end=20, row_step=3, n_threads=3)
    np.testing.assert_array_equal(part, full[1:20:3])


def test_golden_fixtures_record_what_produced_their_ot_numbers(capsys):
    """Every fixture says whether a real POT or the oracle shim stood behind ``import ot`` when the reference's code ran
    (tests/golden/gen_golden.py prefers a real POT when one can be imported).  While this prints "oracle-shim" the parity of
    the OT arithmetic is UNPINNED; it may only be called pinned for fixtures that say "pot==...". """
    from conftest import GOLDEN_CASES, GOLDEN_OPTION_CASES, load_golden, load_golden_pack
    sources = {name: str(load_golden(name)["ot_source"]) for name in GOLDEN_CASES + GOLDEN_OPTION_CASES}
    sources.update({"random_pack[%d]" % i: str(g["ot_source"]) for i, g in enumerate(load_golden_pack())})
    with capsys.disabled():
        print("\ngolden fixtures, ot_source: %s" % sorted(set(sources.values())))
    assert all(s == "oracle-shim" or s.startswith("pot==") for s in sources.values()), sources


def test_pin_with_pot_tool_runs_end_to_end_in_self_test_mode():
    """tools/pin_with_pot.py is the one command that closes "parity unpinned" for anybody who has POT.  POT is not importable
    here, so the tool's plumbing is exercised against the oracle itself (--self-test never prints PINNED) -- including its list of
    c3 / reg 0.01 pairs whose tau-absorption lands on the last update, which the tool re-checks against the oracle's flags --
    and without POT it must say UNPINNED and exit 2."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pin_with_pot.py"), "--self-test", "--per-class", "3", "--rows", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SELF-TEST passed" in r.stdout and "PINNED:" not in r.stdout
    assert "'absorb-on-last'" in r.stdout and "'nan-revert': 36" in r.stdout      # (3 of the 12 sparse rows x 12 columns: all of them revert)
    try:
        import ot  # noqa: F401
    except Exception:
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "pin_with_pot.py")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 2 and "UNPINNED" in r.stdout
