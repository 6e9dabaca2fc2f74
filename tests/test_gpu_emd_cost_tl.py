"""Exact-EMD kernel, cost-matrix kernel and the tl.* surface on the GPU, against oracle + golden fixtures."""
import numpy as np
import pandas as pd
import pytest
import scipy.spatial.distance as ssd

from conftest import GOLDEN_CASES, GOLDEN_OPTION_CASES, GOLDEN_REAL, frame_digests, golden_adata, load_golden, load_golden_pack
from oracle import oracle as O
from pilot_amd import _lib, engine, tl
from pilot_amd.synthetic import CONFIGS, make_cells, make_problem

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------- exact EMD
@pytest.mark.parametrize("cfg,step", [("c1", 1), ("c2", 9), ("c3", 75)])
def test_emd_parity(cfg, step):
    P, M = make_problem(**CONFIGS[cfg])
    Eo = O.emd_grid(P, M, row_step=step, n_threads=16)
    Eg, info = engine.emd_grid(P, M, row_step=step, mode="all", return_info=True)
    assert np.abs(Eg - Eo).max() <= 1e-12
    assert (info["n_aug"] >= 0).all() and info["n_aug"].max() > 0      # >= 0: the warm start already solves a == b


@pytest.mark.parametrize("cfg,step", [("c3", 1), ("c4", 16)])
def test_emd_full_size_grid_against_the_network_simplex(cfg, step):
    """BASELINE.json's full sizes, EVERY pair of c3 (360 000; a sixteenth of c4's rows): the kernel's successive shortest
    paths against the oracle's network simplex -- a different algorithm, so an LP value that is off shows (the LP optimum
    is unique); and the properties that do not need an oracle: a symmetric matrix with a zero diagonal."""
    P, M = make_problem(**CONFIGS[cfg])
    Eg = engine.emd_grid(P, M)
    Eo = O.emd_grid(P, M, row_step=step, n_threads=16, fast="ns")
    assert np.abs(Eg[::step] - Eo).max() <= 1e-12
    assert np.array_equal(Eg, Eg.T) and np.abs(np.diag(Eg)).max() == 0.0


@pytest.mark.parametrize("K", [1, 2, 3, 5, 8, 11, 14, 15, 16])
def test_emd_four_pairs_per_wave_kernel_against_the_one_pair_kernel(K, switches):
    """K <= 16 (what real cohorts have: the reference test's own has 14 clusters): emd_multi_kernel solves four pairs per
    wavefront, one per 16-lane DPP row -- group minima by DPP butterflies, the node sets of a search as wave masks, augmentations
    through path masks instead of a walk.  Same algorithm and the same fp64 arithmetic per pair as the one-pair-per-wave
    kernel (PILOT_OT_EMD_MULTI=0): the same number of augmentations for every pair and the LP value to rounding, with the flow
    values in LDS (=1) or in the global slab (=2); a pair's bits depend neither on its slot mates nor on the row subset nor on
    the mode (all / upper / mirror).  Sparse one-cell-type patients (rows with up to K arcs), duplicates and unequal masses included."""
    rng = np.random.default_rng(K)
    N = 45
    P, M = make_problem(N, K, 6, seed=900 + K, cells_per_patient=300)
    if K == 1:
        M = np.zeros((1, 1))
    else:
        P[::5] = 0.0; P[::5, rng.integers(0, K, size=P[::5].shape[0])] = 1.0
    P[7] = P[8]
    P[9] *= 0.5
    Eo = O.emd_grid(P, M, n_threads=16, fast="ns")
    switches.setenv("PILOT_OT_EMD_MULTI", "0")
    E0, i0 = engine.emd_grid(P, M, mode="all", return_info=True)
    res = {}
    for mode in ("1", "2"):
        switches.setenv("PILOT_OT_EMD_MULTI", mode)
        E1, i1 = engine.emd_grid(P, M, mode="all", return_info=True)
        np.testing.assert_array_equal(i1["n_aug"], i0["n_aug"])
        assert np.abs(E1 - E0).max() <= 1e-14 and np.abs(E1 - Eo).max() <= 1e-12
        rows = np.arange(3, N, 7)
        np.testing.assert_array_equal(engine.emd_grid(P, M, row_begin=3, row_step=7, mode="all"), E1[rows])
        np.testing.assert_array_equal(engine.emd_grid(P, M, row_begin=3, row_step=7, mode="upper"),
                                      np.where(np.arange(N)[None, :] >= rows[:, None], E1[rows], 0.0))
        res[mode] = E1
    np.testing.assert_array_equal(res["1"], res["2"])
    switches.delenv("PILOT_OT_EMD_MULTI")
    np.testing.assert_array_equal(engine.emd_grid(P, M, mode="all"), res["1"])          # the default is one of the two


@pytest.mark.parametrize("K", [2, 5, 9, 13, 16])
def test_emd_whole_grids_at_small_k_against_the_network_simplex(K):
    """tools/emd_multi_probe.py at test size: EVERY pair of a 300-patient grid (45 150 solved pairs: two hundred pairs per 16-lane
    group, so every group refills, finishes and idles at the end of the queue many times) against the oracle's network simplex --
    a different algorithm; symmetric, zero diagonal, no guard tripped; a round-robin row shard in `upper` mode gives the same bits."""
    P, M = make_problem(300, K, 8, seed=K, cells_per_patient=200)
    E, info = engine.emd_grid(P, M, return_info=True)
    assert (info["n_aug"][np.triu_indices(300)] >= 0).all()
    assert np.abs(E - O.emd_grid(P, M, n_threads=16, fast="ns")).max() <= 1e-12
    assert np.array_equal(E, E.T) and np.abs(np.diag(E)).max() == 0.0
    rows = np.arange(1, 300, 4)
    np.testing.assert_array_equal(engine.emd_grid(P, M, row_begin=1, row_step=4, mode="upper"),
                                  np.where(np.arange(300)[None, :] >= rows[:, None], E[rows], 0.0))


def test_emd_four_pairs_per_wave_kernel_on_the_reference_cohort(switches):
    """Every pair of the reference test's own cohort (Kidney_IgAN_G: 634 patients x 14 clusters, up to 101 augmentations per
    pair) against the oracle's network simplex, and against the one-pair-per-wave kernel augmentation for augmentation."""
    g = load_golden(GOLDEN_REAL)
    P = g["proportions"]; M = g["cost"] / g["cost"].max()
    E1, i1 = engine.emd_grid(P, M, return_info=True)
    assert np.abs(E1 - O.emd_grid(P, M, n_threads=16, fast="ns")).max() <= 1e-12
    assert np.array_equal(E1, E1.T) and np.abs(np.diag(E1)).max() == 0.0
    switches.setenv("PILOT_OT_EMD_MULTI", "0")
    E0, i0 = engine.emd_grid(P, M, return_info=True)
    iu = np.triu_indices(P.shape[0])
    np.testing.assert_array_equal(i1["n_aug"][iu], i0["n_aug"][iu])
    assert np.abs(E1 - E0).max() <= 1e-14


def test_emd_modes_and_symmetry():
    P, M = make_problem(**CONFIGS["c2"])
    full = engine.emd_grid(P, M)                                   # auto -> mirror
    allp = engine.emd_grid(P, M, mode="all")
    up = engine.emd_grid(P, M, mode="upper")
    assert np.array_equal(full, full.T) and np.abs(np.diag(full)).max() < 1e-15
    assert np.abs(full - allp).max() < 1e-13
    np.testing.assert_array_equal(np.triu(up), np.triu(full))
    assert np.all(np.tril(up, -1) == 0)
    rows = engine.emd_grid(P, M, row_begin=2, row_step=8, mode="upper")
    np.testing.assert_array_equal(rows, up[2::8])
    with pytest.raises(ValueError):
        engine.emd_grid(P, M, row_begin=1, mode="mirror")


@pytest.mark.parametrize("K", [1, 2, 9, 16, 17, 32, 33, 64, 65, 100, 128, 129, 192, 193, 256])
def test_emd_every_k_regime(K):
    """K <= 16: four pairs per wave (emd_multi_kernels.hpp); K <= 64: one row/column per lane (labels with the column potential up to K = 32, without it beyond: emd_ul); K > 64:
    two (flow support masks of 2 x 64 bits per row); K > 128: three / four, cost matrix read from global memory, lazy
    restarts, row loops without bounds tests."""
    P, M = make_problem(12, K, 6, seed=200 + K, cells_per_patient=500)
    if K == 1:
        M = np.zeros((1, 1))
    Eo = O.emd_grid(P, M, n_threads=16)
    Eg = engine.emd_grid(P, M, mode="all")
    assert np.abs(Eg - Eo).max() <= 1e-12


@pytest.mark.parametrize("K", [257, 300, 520])
def test_emd_beyond_256_cell_types_runs_the_workgroup_kernel(K):
    """The reference's ot.emd2 loop has no limit on the number of cell types (Trajectory.py:507-511): beyond the
    one-wave-per-pair kernel's 256 a pair is solved by one workgroup (emd_generic_kernel.hpp).  Same LP value as the oracle;
    the symmetric shortcut (upper triangle + mirror), row shards and sparse / non-symmetric inputs included."""
    P, M = make_problem(7, K, 5, seed=K, cells_per_patient=4000)
    Eo = O.emd_grid(P, M, n_threads=16, fast=True)
    Eg, info = engine.emd_grid(P, M, return_info=True)                # auto: symmetric cost -> upper triangle + mirror
    assert np.abs(Eg - Eo).max() <= 1e-12 and (info["n_aug"] >= 0).all()
    Ea = engine.emd_grid(P, M, mode="all")
    assert np.abs(Ea - Eo).max() <= 1e-12
    np.testing.assert_array_equal(engine.emd_grid(P, M, mode="all", row_begin=2, row_end=7, row_step=3), Ea[2:7:3])      # row shards: same bits
    rng = np.random.default_rng(K)
    Mn = rng.random((K, K))
    Ps = P.copy()
    Ps[rng.random(P.shape) < 0.6] = 0.0
    Ps[:, 0] += 1e-3
    Ps /= Ps.sum(1, keepdims=True)
    assert np.abs(engine.emd_grid(Ps, Mn) - O.emd_grid(Ps, Mn, n_threads=16, fast=True)).max() <= 1e-12


def test_emd_nonsymmetric_cost_and_sparse_histograms():
    rng = np.random.default_rng(8)
    K = 24
    P = rng.dirichlet(0.2 * np.ones(K), size=15)
    P[P < 1e-3] = 0.0                                             # zero-mass bins (POT drops them)
    P /= P.sum(1, keepdims=True)
    M = rng.random((K, K))
    Eo = O.emd_grid(P, M, n_threads=16)
    Eg = engine.emd_grid(P, M)                                     # auto -> all (not symmetric)
    assert np.abs(Eg - Eo).max() <= 1e-12


@pytest.mark.parametrize("K,sparsity,nonzero_diag,seed", [(5, 0.5, False, 1), (33, 0.8, True, 2), (64, 0.3, False, 3),
                                                         (70, 0.9, True, 4), (128, 0.6, False, 5)])
def test_emd_randomized_stress(K, sparsity, nonzero_diag, seed):
    """Degenerate inputs for the augmenting-path solver: very sparse histograms (most bins empty, single-bin patients),
    duplicates, unequal total mass (POT rescales b), random non-metric costs with ties (quantised) and, optionally, a
    non-zero diagonal (no warm start).  Value against the CPU oracle, and the LP bounds 0 <= cost <= max(M)."""
    rng = np.random.default_rng(seed)
    N = 14
    P = rng.random((N, K)) ** 3
    P[rng.random((N, K)) < sparsity] = 0.0
    P[0] = 0.0; P[0, K // 2] = 1.0                                  # all mass in one bin
    P[1] = P[2]                                                     # duplicate patients
    P[P.sum(1) == 0, 0] = 1.0
    P /= P.sum(1, keepdims=True)
    P[3] *= 0.5                                                     # unequal mass: emd2 rescales the second histogram
    M = np.round(rng.random((K, K)) * 8) / 8                        # many exactly equal costs
    if not nonzero_diag:
        np.fill_diagonal(M, 0.0)
    Eo = O.emd_grid(P, M, n_threads=16)
    Eg, info = engine.emd_grid(P, M, mode="all", return_info=True)
    assert (info["n_aug"] >= 0).all()
    assert np.abs(Eg - Eo).max() <= 1e-12
    assert Eg.min() >= -1e-15 and Eg.max() <= M.max() * P.sum(1).max() + 1e-12


def test_emd_unequal_masses_are_not_mirrored():
    """emd2 rescales b to the mass of a: with unequal masses the matrix of a symmetric cost is NOT symmetric, and the
    automatic mode must solve every ordered pair (found by tools/fuzz_emd.py)."""
    from pilot_amd import multi
    rng = np.random.default_rng(5)
    N, K = 21, 12
    P = rng.dirichlet(np.ones(K), size=N) * rng.uniform(0.3, 3.0, size=(N, 1))
    M = rng.random((K, K)); M = M + M.T; np.fill_diagonal(M, 0.0)
    Eo = O.emd_grid(P, M, n_threads=8)
    assert np.abs(Eo - Eo.T).max() > 1e-3
    np.testing.assert_allclose(engine.emd_grid(P, M), Eo, rtol=0, atol=1e-11)
    np.testing.assert_allclose(multi.emd_grid_multi(P, M, devices=[0, 0]), Eo, rtol=0, atol=1e-11)
    Pn = P / P.sum(1, keepdims=True)                                  # equal masses: the mirrored form is used and agrees
    np.testing.assert_allclose(engine.emd_grid(Pn, M), O.emd_grid(Pn, M, n_threads=8), rtol=0, atol=1e-12)


def test_emd_upper_triangle_queue_over_row_subsets():
    """mode "upper" draws its pairs from a counter that enumerates only column >= row, row by row (row offsets inverted
    with a square root and fixed up): every row subset must give exactly the rows of the full matrix, nothing solved
    twice, nothing left out (untouched entries stay 0)."""
    rng = np.random.default_rng(77)
    N, K = 157, 9
    P = rng.dirichlet(np.ones(K), size=N)
    M = rng.random((K, K)); M = M + M.T; np.fill_diagonal(M, 0.0)
    full = engine.emd_grid(P, M, mode="all")
    cases = [(0, N, 1), (0, N, 2), (1, N, 2), (5, 140, 7), (156, 157, 1), (0, 1, 1), (3, 150, 149), (10, 11, 5), (0, N, 156)]
    cases += [(int(b), int(min(N, b + 1 + rng.integers(0, N - b))), int(s)) for b, s in zip(rng.integers(0, N, 12), rng.integers(1, 40, 12))]
    for rb, re_, rs in cases:
        rows = np.arange(rb, re_, rs)
        U, info = engine.emd_grid(P, M, row_begin=rb, row_end=re_, row_step=rs, mode="upper", return_info=True)
        want = np.where(np.arange(N)[None, :] >= rows[:, None], full[rows], 0.0)
        np.testing.assert_array_equal(U, want)
        assert ((info["n_aug"] > 0) <= (np.arange(N)[None, :] >= rows[:, None])).all()


@pytest.mark.parametrize("K,seed", [(40, 11), (64, 12), (90, 13), (150, 14)])
def test_emd_lattice_masses_and_integer_costs_vs_linprog(K, seed):
    """Everything ties: masses are multiples of 1/32 (many bins empty, sources run dry and arcs run empty together),
    costs are small integers.  The LP optimum from scipy's HiGHS on sampled pairs, the oracle on all of them; every
    code path of the search (several augmentations per search, lazy restarts for K > 64, single-hop paths) is hit."""
    from scipy.optimize import linprog
    rng = np.random.default_rng(seed)
    N = 40
    P = rng.multinomial(32, rng.dirichlet(0.3 * np.ones(K), size=N)[0], size=N).astype(np.float64) / 32.0
    P[5] = P[6]
    line = np.abs(np.arange(K)[:, None] - np.arange(K)[None, :]).astype(np.float64)
    M = np.minimum(line, 7.0)                                       # truncated line metric: integer, heavily tied
    Eo = O.emd_grid(P, M, n_threads=16)
    Eg, info = engine.emd_grid(P, M, return_info=True)
    assert (info["n_aug"][np.triu_indices(N)] >= 0).all()
    assert np.abs(Eg - Eo).max() <= 1e-12
    assert np.abs(Eg - Eg.T).max() == 0.0 and np.abs(np.diag(Eg)).max() == 0.0
    A_eq = np.zeros((2 * K, K * K))
    for i in range(K):
        A_eq[i, i * K:(i + 1) * K] = 1.0
        A_eq[K + i, i::K] = 1.0
    for (i, j) in [(0, 1), (2, 30), (5, 6), (17, 39)]:
        res = linprog(M.ravel(), A_eq=A_eq, b_eq=np.concatenate([P[i], P[j]]), bounds=(0, None), method="highs")
        assert res.status == 0 and abs(res.fun - Eg[i, j]) <= 1e-9


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_emd_golden(name):
    g = load_golden(name)
    E = engine.emd_grid(g["proportions"], g["cost"] / g["cost"].max())
    assert np.abs(E - g["emd_unreg"]).max() <= 1e-12


# ------------------------------------------------------------------------------- pre-pass (histogram, medians)
@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("categorical", [False, True])
def test_cluster_representations_bit_exact_vs_reference(name, categorical):
    g = load_golden(name)
    ad, cell_col = golden_adata(g, categorical=categorical)
    annot = ad.obs[[cell_col, "sampleID", "status"]].copy()
    annot.columns = ["cell_type", "sampleID", "status"]
    rep = tl.Cluster_Representations(annot, regulizer=0.2, normalization=True)
    assert isinstance(rep, dict) and [str(k) for k in rep.keys()] == list(g["samples"])
    got = np.stack(list(rep.values()))
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, g["proportions"])           # bit-exact with the reference's output
    for regulizer in (1.0, 0.05):
        rep = tl.Cluster_Representations(annot, regulizer=regulizer)
        ora, _ = O.cluster_representations(annot["cell_type"], annot["sampleID"], regulizer=regulizer)
        for k in rep:
            np.testing.assert_array_equal(rep[k], ora[k])
    raw = tl.Cluster_Representations(annot, normalization=False)
    ora, _ = O.cluster_representations(annot["cell_type"], annot["sampleID"], normalization=False)
    for k in raw:
        np.testing.assert_array_equal(raw[k], ora[k])


def test_proportions_kernel_large_and_ragged():
    rng = np.random.default_rng(0)
    C, N, K = 300_000, 97, 41
    cc = rng.integers(0, K, C).astype(np.int32); sc = rng.integers(0, N, C).astype(np.int32)
    cc[rng.random(C) < 0.01] = -1                                   # missing labels are skipped
    P = engine.proportions(cc, sc, N, K, regulizer=0.2, n_total=C)
    ok = cc >= 0
    counts = np.bincount(sc[ok].astype(np.int64) * K + cc[ok], minlength=N * K).reshape(N, K).astype(np.float64)
    prior = counts.sum(0) / (C - 1) * 0.2
    want = np.stack([(counts[n] + prior) / (sum(counts[n]) + sum(prior)) for n in range(N)])
    np.testing.assert_array_equal(P, want)


@pytest.mark.parametrize("layout", ["sample_major", "shuffled", "few_wide"])
def test_count_kernel_lds_window_and_global_fallback(layout):
    """count_kernel tallies a block's 2048 cells in an LDS window of sample rows when (rows x K) fits 8192 counters and
    falls back to global atomics when it does not: sample-major cohorts (the window path), a shuffled one with 700 samples
    (every chunk spans all samples: the fallback), and few samples x many types (window limited by K).  Same integers either
    way; the first row of every sample comes out of the same pass."""
    rng = np.random.default_rng(11)
    if layout == "sample_major":
        C, N, K = 250_000, 83, 50
        sc = np.sort(rng.integers(0, N, C)).astype(np.int32)
    elif layout == "shuffled":
        C, N, K = 250_000, 700, 50
        sc = rng.integers(0, N, C).astype(np.int32)
    else:
        C, N, K = 120_000, 5, 3000
        sc = np.sort(rng.integers(0, N, C)).astype(np.int32)
    cc = rng.integers(0, K, C).astype(np.int32)
    cc[rng.random(C) < 0.01] = -1
    sc[rng.random(C) < 0.005] = -1
    P, first = engine.proportions_and_first_rows(cc, sc, N, K, regulizer=0.2, n_total=C)
    ok = (cc >= 0) & (sc >= 0)
    counts = np.bincount(sc[ok].astype(np.int64) * K + cc[ok], minlength=N * K).reshape(N, K).astype(np.float64)
    prior = counts.sum(0) / (C - 1) * 0.2
    want = np.stack([(counts[n] + prior) / (sum(counts[n]) + sum(prior)) for n in range(N)])
    np.testing.assert_array_equal(P, want)
    want_first = np.full(N, -1, dtype=np.int64)
    idx = np.flatnonzero(sc >= 0)
    want_first[sc[idx][::-1]] = idx[::-1]
    np.testing.assert_array_equal(first, want_first)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("C,D,K,N", [(300_000, 30, 50, 100), (20_000, 14, 14, 600), (70_000, 70, 3, 9)])
def test_prepass_in_one_call_gives_the_three_calls_bits(dtype, C, D, K, N):
    """pilot_ot_prepass_dev: proportions, first rows and medians from ONE upload of the two code columns -- the general
    median path (grouped rows + radix select), the one-launch small-cohort path and a windowed D > 64, against the separate
    entry points and numpy."""
    rng = np.random.default_rng(C + D)
    X = (rng.standard_normal((C, D)) * 4).astype(dtype)
    X[rng.random((C, D)) < 0.05] = 0.0
    cc = rng.integers(0, K, C).astype(np.int32)
    sc = np.sort(rng.integers(0, N, C)).astype(np.int32)
    cc[rng.random(C) < 0.01] = -1
    X = np.concatenate([X, X[:1]])[:C]                      # (a private, contiguous array)
    up = engine.EmbeddingUpload(X)
    up.SMALL_BYTES = 0
    try:
        P, first, cen = up.prepass(cc, sc, N, K, regulizer=0.2, n_total=C)
        cen2 = up.medians(cc, K)
    finally:
        up.close()
    P2, first2 = engine.proportions_and_first_rows(cc, sc, N, K, regulizer=0.2, n_total=C)
    np.testing.assert_array_equal(P, P2)
    np.testing.assert_array_equal(first, first2)
    np.testing.assert_array_equal(cen, cen2)
    for k in range(K):
        np.testing.assert_array_equal(cen[k], np.median(X[cc == k], axis=0).astype(np.float64))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("C,D,K", [(1, 1, 1), (7, 3, 2), (1000, 30, 50), (4097, 65, 3), (3000, 150, 4), (500, 400, 2), (200_000, 30, 50)])
def test_centroid_medians_exact(dtype, C, D, K):
    rng = np.random.default_rng(C + D + K)
    X = (rng.standard_normal((C, D)) * 10).astype(dtype)
    X[rng.random((C, D)) < 0.05] = 0.0                               # ties, signed zeros
    X[rng.random((C, D)) < 0.02] *= -0.0
    cc = rng.integers(0, K, C).astype(np.int32)
    got = engine.centroid_medians(X, cc, K)
    for k in range(K):
        rows = X[cc == k]
        if len(rows) == 0:
            assert np.isnan(got[k]).all()
        else:
            np.testing.assert_array_equal(got[k], np.median(rows, axis=0).astype(np.float64))   # median in X's dtype


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_small_cohort_medians_in_one_launch_give_the_general_path_s_bits(dtype, switches):
    """Small cohorts (C x K x D <= 3.2e7 and no cell type beyond 8192 cells) select their medians in ONE launch, in LDS
    (small_medians_kernel); the general path (PILOT_OT_NO_SMALL_MEDIANS=1) is sixteen launches.  Same keys, ranks and final
    arithmetic: the same bits, with ties, signed zeros, infinities, an empty type, one-cell types, an even / odd split -- and a
    cohort with one type beyond the cap must take the general path by itself."""
    rng = np.random.default_rng(5)
    for C, D, K in ((24227, 14, 14), (5000, 3, 40), (9, 2, 4)):
        X = (rng.standard_normal((C, D)) * 3).astype(dtype)
        X[rng.random((C, D)) < 0.1] = 0.0
        X[rng.random((C, D)) < 0.03] *= -0.0
        X[rng.random((C, D)) < 0.01] = np.inf
        cc = rng.integers(0, K - 1, C).astype(np.int32)             # type K - 1 stays empty
        cc[:3] = [0, 1, 1]
        fast = engine.centroid_medians(X, cc, K)
        switches.setenv("PILOT_OT_NO_SMALL_MEDIANS", "1")
        slow = engine.centroid_medians(X, cc, K)
        switches.delenv("PILOT_OT_NO_SMALL_MEDIANS")
        np.testing.assert_array_equal(fast, slow)
        assert np.isnan(fast[K - 1]).all()
        for k in range(K - 1):
            np.testing.assert_array_equal(fast[k], np.median(X[cc == k], axis=0).astype(np.float64))
    C, D, K = 20000, 2, 2                                             # 12 000 cells of one type: beyond the LDS key buffer
    X = rng.standard_normal((C, D)).astype(dtype)
    cc = (np.arange(C) >= 12000).astype(np.int32)
    got = engine.centroid_medians(X, cc, K)
    for k in range(K):
        np.testing.assert_array_equal(got[k], np.median(X[cc == k], axis=0).astype(np.float64))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_centroid_medians_match_the_reference_pandas_medians(name):
    g = load_golden(name)
    ad, cell_col = golden_adata(g)
    codes, cells = tl._first_appearance_codes(ad.obs[cell_col])
    X = ad.X if str(g["data_type"]) != "scRNA" else ad.obsm["X_pca"]
    got = engine.centroid_medians(X, codes, len(cells))
    _, ora_cent, _ = O.cost_matrix(X, ad.obs[cell_col])
    np.testing.assert_array_equal(got, ora_cent)


# ------------------------------------------------------------------------------- cost matrix
@pytest.mark.parametrize("metric", ["cosine", "euclidean", "sqeuclidean", "cityblock", "chebyshev", "correlation",
                                    "minkowski", "seuclidean", "braycurtis", "canberra", "hamming"])
@pytest.mark.parametrize("K,D", [(2, 3), (50, 30), (100, 50), (130, 7)])
def test_pdist_kernel_vs_scipy(metric, K, D):
    X = np.random.default_rng(K * D).standard_normal((K, D))
    if metric in ("hamming", "canberra"):
        X = np.round(X)                              # equal coordinates and 0/0 terms actually occur
    ref = ssd.squareform(ssd.pdist(X, metric=metric))
    got = engine.pdist_square(X, metric=metric)
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-14)
    assert np.array_equal(got, got.T) and np.all(np.diag(got) == 0)


@pytest.mark.parametrize("metric", ["jaccard", "dice", "yule", "russellrao", "sokalsneath", "rogerstanimoto", "sokalmichener",
                                    "kulczynski1", "jensenshannon", "mahalanobis"])
@pytest.mark.parametrize("K,D", [(3, 2), (50, 30), (130, 7)])
def test_pdist_kernel_vs_scipy_the_remaining_names(metric, K, D):
    """The rest of scipy 1.15's pdist names (Trajectory.py:468 forwards any of them): the boolean dissimilarities on rows with
    actual zeros, Jensen-Shannon on non-negative rows, Mahalanobis with scipy's own VI = inv(cov(X^T))^T (needs K > D).
    0/0 cases must give scipy's NaN / inf, not something else."""
    import warnings
    rng = np.random.default_rng(K * D + len(metric))
    X = rng.random((K, D))
    if metric not in ("jensenshannon", "mahalanobis"):
        X[rng.random((K, D)) < 0.45] = 0.0          # non-zero = True
        X[0] = 0.0                                     # an all-False row: the 0/0 cases
        if K > 2:
            X[2] = X[1]
    if metric == "mahalanobis" and K <= D:
        with pytest.raises(ValueError, match="observations"):
            engine.pdist_square(X, metric=metric)
        with pytest.raises(ValueError, match="observations"):
            ssd.pdist(X, metric=metric)
        return
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = ssd.squareform(ssd.pdist(X, metric=metric))
    got = engine.pdist_square(X, metric=metric)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-14, equal_nan=True)
    assert np.all(np.diag(got) == 0)


@pytest.mark.parametrize("D", [1, 2, 5, 30])
def test_jensenshannon_of_proportional_rows_is_scipy_s_own_rounding(D):
    """Proportional rows: the Jensen-Shannon distance is 0 up to rounding, and scipy returns the root of whatever the rounding left
    -- 0, 1e-8, or NaN where the sum came out below zero.  scipy's build multiplies by the reciprocal of a row's sum; the kernel does
    the same and lands on scipy's values and on scipy's NaNs (found by tools/fuzz_prepass.py: rows [0.4f] and [0.9f], D = 1)."""
    import warnings
    rng = np.random.default_rng(D)
    base = np.abs(rng.standard_normal((40, D))).astype(np.float32).astype(np.float64)
    X = np.concatenate([base, base * float(np.float32(0.9)), base * float(np.float32(2.25)), base[:5] * 0.0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = ssd.squareform(ssd.pdist(X, metric="jensenshannon"))
    got = engine.pdist_square(X, metric="jensenshannon")
    off = ~np.eye(len(X), dtype=bool)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    if D <= 2:
        assert np.isnan(ref[off]).any()                       # (the case exists at the small dimensions)
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_cost_matrix_golden(name):
    g = load_golden(name)
    ad, cell_col = golden_adata(g)
    data, annot = (tl.extract_data_anno_scRNA_from_h5ad(ad, "X_pca", cell_col, "sampleID", "status")
                   if str(g["data_type"]) == "scRNA" else
                   tl.extract_data_anno_pathomics_from_h5ad(ad, list(ad.var_names), cell_col, "sampleID", "status"))
    dis, df = tl.cost_matrix(annot, data, metric="cosine")
    np.testing.assert_allclose(dis, g["cost"], rtol=0, atol=1e-14)
    assert isinstance(df, pd.DataFrame) and df.index.name == str(g["cost_index_name"]) == "cell_types"
    assert [str(c) for c in df.columns] == list(g["cells"]) == [str(c) for c in df.index]


# ------------------------------------------------------------------------------- tl surface
@pytest.mark.parametrize("name", GOLDEN_CASES + [GOLDEN_REAL])
@pytest.mark.parametrize("mode", ["unreg", "reg"])
def test_wasserstein_distance_end_to_end_vs_reference_fixture(name, mode, tmp_path, monkeypatch):
    """Mirror of the reference's own test (test/test_pilot.py:9-28) plus numbers: same call, same uns keys,
    same Python types, values within tolerance of what the reference's code produced (golden).  The last case IS the
    reference test's input: Tutorial/Datasets/Kidney_IgAN_G.h5ad with data_type='Pathomics' (634 patients, 14 clusters)."""
    monkeypatch.chdir(tmp_path)
    g = load_golden(name)
    ad, cell_col = golden_adata(g, categorical=(name.startswith("ragged")))
    kw = dict(clusters_col=cell_col, sample_col="sampleID", status="status", regularized=mode, reg=float(g["reg"]))
    if str(g["data_type"]) == "scRNA":
        tl.wasserstein_distance(ad, emb_matrix="X_pca", engine_options={"precision": "fp64"}, **kw)
    else:
        tl.wasserstein_distance(ad, data_type="Pathomics", engine_options={"precision": "fp64"}, **kw)
    u = ad.uns
    assert sorted(u.keys()) == list(g["uns_keys"])
    E = u["EMD"]
    assert isinstance(E, np.ndarray) and E.dtype == np.float64
    assert E.shape[0] == E.shape[1] == len(u["real_labels"]) == len(g["samples"])       # test_pilot.py:26-28
    want = g["emd_unreg"] if mode == "unreg" else g["emd_reg"]
    assert np.abs(E[::int(g["row_step"])] - want).max() <= 1e-12        # (the 634-patient case stores rows 0, 3, 6, ..)
    assert frame_digests(u) == (str(g["data_sha256"]), str(g["annot_sha256"]))     # uns['data'], uns['annot']: the reference's bytes
    df = u["EMD_df"]
    assert isinstance(df, pd.DataFrame) and df.index.name == "sampleID"
    assert [str(s) for s in df.index] == list(g["samples"]) == [str(s) for s in df.columns]
    np.testing.assert_array_equal(df.to_numpy(), E.T)              # from_dict(EMD).T, Trajectory.py:518
    assert isinstance(u["proportions"], dict) and [str(k) for k in u["proportions"]] == list(g["samples"])
    np.testing.assert_array_equal(np.stack(list(u["proportions"].values())), g["proportions"])
    np.testing.assert_allclose(u["cost"].to_numpy(), g["cost"], atol=1e-14)              # stored UN-normalised
    assert [str(x) for x in u["real_labels"]] == list(g["real_labels"])
    assert isinstance(u["annot"], pd.DataFrame) and isinstance(u["data"], pd.DataFrame)
    assert (E + E).shape == E.shape and (E / E.max()).max() == 1.0                        # test_pilot.py:30, ploting.py:95


@pytest.mark.parametrize("name", GOLDEN_OPTION_CASES)
@pytest.mark.parametrize("mode", ["unreg", "reg"])
def test_wasserstein_distance_with_other_options_vs_reference_fixture(name, mode, tmp_path, monkeypatch):
    """The same call as the reference was run with (metric, regulizer, reg from the fixture): every output it left."""
    monkeypatch.chdir(tmp_path)
    g = load_golden(name)
    ad, cell_col = golden_adata(g)
    tl.wasserstein_distance(ad, emb_matrix="X_pca", clusters_col=cell_col, sample_col="sampleID", status="status",
                            metric=str(g["metric"]), regulizer=float(g["regulizer"]), regularized=mode, reg=float(g["reg"]),
                            engine_options={"precision": "fp64"})
    u = ad.uns
    assert sorted(u.keys()) == list(g["uns_keys"])
    assert frame_digests(u) == (str(g["data_sha256"]), str(g["annot_sha256"]))
    assert [str(k) for k in u["proportions"]] == list(g["samples"])
    np.testing.assert_array_equal(np.stack(list(u["proportions"].values())), g["proportions"])
    assert [str(c) for c in u["cost"].columns] == list(g["cells"]) == [str(c) for c in u["cost"].index]
    np.testing.assert_allclose(u["cost"].to_numpy(), g["cost"], rtol=0, atol=1e-13 * max(1.0, g["cost"].max()))
    want = g["emd_unreg"] if mode == "unreg" else g["emd_reg"]
    assert np.abs(u["EMD"] - want).max() <= 1e-12
    np.testing.assert_array_equal(u["EMD_df"].to_numpy(), u["EMD"].T)
    assert [str(x) for x in u["real_labels"]] == list(g["real_labels"])
    if mode == "reg":                                              # and the default (f32-valued) precision
        tl.wasserstein_distance(ad, emb_matrix="X_pca", clusters_col=cell_col, sample_col="sampleID", status="status",
                                metric=str(g["metric"]), regulizer=float(g["regulizer"]), regularized=mode, reg=float(g["reg"]))
        assert np.abs(ad.uns["EMD"] - want).max() <= 1e-5


def test_wasserstein_distance_over_the_random_reference_pack(tmp_path, monkeypatch):
    """Twelve random cohorts x random options, each produced by the reference's own code: the same call must leave the
    same proportions (bit for bit), cost, matrices and labels, in both modes."""
    monkeypatch.chdir(tmp_path)
    for g in load_golden_pack():
        for mode in ("unreg", "reg"):
            ad, cell_col = golden_adata(g)
            tl.wasserstein_distance(ad, emb_matrix="X_pca", clusters_col=cell_col, sample_col="sampleID", status="status",
                                    metric=str(g["metric"]), regulizer=float(g["regulizer"]), regularized=mode, reg=float(g["reg"]),
                                    engine_options={"precision": "fp64"})
            u = ad.uns
            tag = "%s %s reg=%g regulizer=%g" % (mode, g["metric"], float(g["reg"]), float(g["regulizer"]))
            assert [str(k) for k in u["proportions"]] == list(g["samples"]), tag
            np.testing.assert_array_equal(np.stack(list(u["proportions"].values())), g["proportions"], err_msg=tag)
            assert [str(c) for c in u["cost"].columns] == list(g["cells"]), tag
            np.testing.assert_allclose(u["cost"].to_numpy(), g["cost"], rtol=0, atol=1e-13 * max(1.0, g["cost"].max()), err_msg=tag)
            want = g["emd_unreg"] if mode == "unreg" else g["emd_reg"]
            assert np.abs(u["EMD"] - want).max() <= 1e-11, tag
            assert [str(x) for x in u["real_labels"]] == list(g["real_labels"]), tag


def test_wasserstein_distance_default_precision_c2_shape(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    ad = make_cells(100, 30, 30, seed=1, cells_per_patient=300)
    tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized="reg", reg=0.1)
    P = np.stack(list(ad.uns["proportions"].values()))
    M = ad.uns["cost"].to_numpy(); M = M / M.max()
    Eo = O.sinkhorn_grid(P, M, 0.1, row_step=10, n_threads=16)
    assert np.abs(ad.uns["EMD"][::10] - Eo).max() <= 1e-5
    ad.uns = {}
    tl.wasserstein_distance(ad, emb_matrix="X_pca")                 # reference default: exact OT
    Eo = O.emd_grid(P, M, row_step=10, n_threads=16)
    assert np.abs(ad.uns["EMD"][::10] - Eo).max() <= 1e-12


def test_unsupported_metric_and_sil_ari_raise(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    ad = make_cells(6, 4, 3, seed=2, cells_per_patient=30)
    with pytest.raises(NotImplementedError, match="pdist names"):
        tl.wasserstein_distance(ad, emb_matrix="X_pca", metric="wminkowski")       # (removed from scipy; a callable is not a name either)
    assert ad.uns == {}
    with pytest.raises(NotImplementedError):
        tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized="reg", return_sil_ari=True)
    assert ad.uns == {}


def test_return_sil_ari_branch_end_to_end_with_a_stub_clustering(tmp_path, monkeypatch):
    """return_sil_ari=True (Trajectory.py:108-113): the Leiden clustering is the reference's OWN function, looked up lazily; with a
    stand-in for it (scanpy / leidenalg are not in this image) the whole branch runs -- Clustering gets EMD / EMD.max() and the
    annotation frame, its labels replace uns['real_labels'], the silhouette comes from the device, and the ARI is stored
    (only the lookup had a test)."""
    import sys
    import types
    from sklearn.metrics import silhouette_score
    monkeypatch.chdir(tmp_path)
    ad = make_cells(20, 10, 10, seed=0, cells_per_patient=200)
    seen = {}

    def clustering(EMD, annot, metric="cosine", res=0.01, steper=0.01):
        seen.update(EMD=EMD.copy(), annot=annot, metric=metric, res=res, steper=steper)
        labels = ["case" if i % 2 else "ctrl" for i in range(EMD.shape[0])]
        return [0] * EMD.shape[0], 0.75, labels

    fake = types.ModuleType("pilotpy.tools.Trajectory")
    fake.Clustering = clustering
    for name, mod in (("pilotpy", types.ModuleType("pilotpy")), ("pilotpy.tools", types.ModuleType("pilotpy.tools")), ("pilotpy.tools.Trajectory", fake)):
        monkeypatch.setitem(sys.modules, name, mod)
    tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized="reg", reg=0.1, return_sil_ari=True, res=0.05, steper=0.02)
    E = ad.uns["EMD"]
    np.testing.assert_array_equal(seen["EMD"], E / E.max())
    assert seen["annot"] is ad.uns["annot"] and (seen["metric"], seen["res"], seen["steper"]) == ("cosine", 0.05, 0.02)
    assert ad.uns["ARI"] == 0.75 and ad.uns["real_labels"] == ["case" if i % 2 else "ctrl" for i in range(20)]
    assert abs(ad.uns["Sil"] - silhouette_score(E / E.max(), ad.uns["real_labels"], metric="cosine")) <= 1e-12
