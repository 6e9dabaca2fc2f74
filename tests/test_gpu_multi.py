"""Multi-GPU entry points of libpilot_ot.so on a 1-GPU box: RCCL with one rank (ncclCommInitAll / ncclCommInitRank,
ncclAllGather, ncclAllReduce all execute), and 2 / 3 / 8 logical shards on device 0 with the peer-copy gather.  In every
form the assembled matrix must equal the single-device matrix BIT FOR BIT (same kernel, same pair -> same arithmetic)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from pilot_amd import _lib, engine, multi, tl
from pilot_amd.synthetic import CONFIGS, make_cells, make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    P, M = make_problem(**CONFIGS["c2"])
    E, info = engine.sinkhorn_grid(P, M, 0.1, precision="fp32", return_info=True)
    X = engine.emd_grid(P, M)
    return P, M, E, info, X


@pytest.mark.parametrize("devices,gather", [([0], "rccl"), ([0], "copy"), ([0, 0], "auto"), ([0, 0, 0], "copy"),
                                            ([0] * 8, "copy")])
def test_sharded_matrix_is_bit_identical_to_the_single_device_one(c2, devices, gather):
    P, M, E, info, X = c2
    Em, im = multi.sinkhorn_grid_multi(P, M, 0.1, devices=devices, gather=gather, precision="fp32", return_info=True)
    np.testing.assert_array_equal(Em, E)
    np.testing.assert_array_equal(im["iters"], info["iters"])
    np.testing.assert_array_equal(im["flags"], info["flags"])
    np.testing.assert_array_equal(im["err"], info["err"])
    Xm, xi = multi.emd_grid_multi(P, M, devices=devices, gather=gather, return_info=True)
    np.testing.assert_array_equal(Xm, X)                 # upper triangle solved per shard, mirrored after the gather


def _n_devices():
    return int(_lib.device_count())


@pytest.mark.parametrize("gather", ["rccl", "copy"])
def test_real_devices_rccl_all_gather_bit_identical_to_one_device(c2, gather):
    """More than one PHYSICAL device: ncclCommInitAll over distinct devices, one host thread per shard on its own device, the
    grouped ncclAllGather over xGMI (or peer copies), the row interleave on every device.  Every device count the node offers up to
    eight, every grid entry point (Sinkhorn with its per-pair outputs, exact OT, the resident MultiPlan with c3 and its timers):
    the assembled matrix must be the single-device matrix bit for bit, and RCCL must report as many ranks as devices were asked for."""
    if _n_devices() < 2:
        pytest.skip("needs at least two visible GPUs (the 1-GPU boxes of this pool skip it; the first multi-GPU node runs it)")
    P, M, E, info, X = c2
    n = min(_n_devices(), 8)
    for g in sorted({2, n, (n // 2) or 2}):
        if g > n:
            continue
        devs = list(range(g))
        Em, im = multi.sinkhorn_grid_multi(P, M, 0.1, devices=devs, gather=gather, precision="fp32", return_info=True)
        np.testing.assert_array_equal(Em, E)
        for k in ("iters", "flags", "err"):
            np.testing.assert_array_equal(im[k], info[k])
        np.testing.assert_array_equal(multi.emd_grid_multi(P, M, devices=devs, gather=gather), X)
    P3, M3 = make_problem(**CONFIGS["c3"])
    ref = engine.sinkhorn_grid(P3, M3, 0.1, precision="auto")
    mp = multi.MultiPlan(P3, M3, devices=list(range(n)), gather=gather)
    for _ in range(3):
        mp.sinkhorn(0.1)
    np.testing.assert_array_equal(mp.fetch(), ref)
    grid_ms, gather_ms = mp.times_ms()
    assert grid_ms.shape == (n,) and (grid_ms > 0).all()
    ranks, user_ranks = mp.rccl_info()
    assert (ranks, user_ranks) == (([n] * n, list(range(n))) if gather == "rccl" else ([0] * n, [-1] * n))
    mp.emd()
    np.testing.assert_array_equal(mp.fetch(), engine.emd_grid(P3, M3))
    mp.close()


def test_rccl_with_repeated_devices_is_refused():
    P, M = make_problem(**CONFIGS["c1"])
    with pytest.raises(ValueError, match="distinct"):
        multi.sinkhorn_grid_multi(P, M, 0.1, devices=[0, 0], gather="rccl")
    with pytest.raises(ValueError, match="not visible"):
        multi.sinkhorn_grid_multi(P, M, 0.1, devices=[0, 63])


def test_multi_plan_resident_and_timed():
    P, M = make_problem(**CONFIGS["c3"])
    ref = engine.sinkhorn_grid(P, M, 0.1, precision="auto")
    mp = multi.MultiPlan(P, M, devices=[0, 0, 0, 0])
    for _ in range(2):
        mp.sinkhorn(0.1)
    E = mp.fetch()
    np.testing.assert_array_equal(E, ref)
    grid_ms, gather_ms = mp.times_ms()
    assert grid_ms.shape == (4,) and (grid_ms > 0).all() and gather_ms > -0.5
    mp.emd()
    np.testing.assert_array_equal(mp.fetch(), engine.emd_grid(P, M))
    mp.close()
    # non-symmetric cost, N not a multiple of the shard count
    rng = np.random.default_rng(0)
    P, _ = make_problem(37, 20, 6, seed=3, cells_per_patient=300)
    M = rng.random((20, 20)); M /= M.max()
    for prec in ("fp32", "bf16x3", "f16x2", "fp64"):
        np.testing.assert_array_equal(multi.sinkhorn_grid_multi(P, M, 0.2, devices=[0, 0, 0], precision=prec),
                                      engine.sinkhorn_grid(P, M, 0.2, precision=prec))
    np.testing.assert_array_equal(multi.emd_grid_multi(P, M, devices=[0, 0, 0]), engine.emd_grid(P, M))


def test_comm_one_rank_all_gather_and_all_reduce():
    """The one-process-per-GPU form with a world of one: ncclCommInitRank, ncclAllGather, ncclAllReduce really run."""
    L = _lib.load()
    comm = multi.Comm(0, 1, key="pytest_%d" % os.getpid())
    assert comm.all_reduce_max(3.25) == 3.25
    N = 11
    A = np.arange(N * N, dtype=np.float64).reshape(N, N)
    bufs = []
    for _ in range(3):
        p = ctypes.c_void_p()
        _lib.check(L.pilot_ot_dev_alloc(ctypes.byref(p), 8 * N * N))
        bufs.append(p)
    _lib.check(L.pilot_ot_memcpy_h2d(bufs[0], A.ctypes.data, A.nbytes))
    comm.all_gather_rows(bufs[0], N, N, bufs[1], bufs[2])
    comm.barrier()
    out = np.zeros_like(A)
    _lib.check(L.pilot_ot_memcpy_d2h(out.ctypes.data, bufs[2], A.nbytes))
    np.testing.assert_array_equal(out, A)
    with pytest.raises(ValueError, match="n_pad"):
        comm.all_gather_rows(bufs[0], N - 1, N, bufs[1], bufs[2])
    for p in bufs:
        L.pilot_ot_dev_free(p)
    comm.close()


def test_tl_wasserstein_distance_over_shards(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    ad = make_cells(20, 10, 10, seed=0, cells_per_patient=200)
    outs = {}
    for name, opts in (("one", None), ("rccl1", {"n_devices": 1, "gather": "rccl"}), ("shards3", {"devices": [0, 0, 0]})):
        for mode in ("unreg", "reg"):
            ad.uns = {}
            tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized=mode, reg=0.1, engine_options=opts)
            outs[name, mode] = ad.uns["EMD"].copy()
    for mode in ("unreg", "reg"):
        np.testing.assert_array_equal(outs["rccl1", mode], outs["one", mode])
        np.testing.assert_array_equal(outs["shards3", mode], outs["one", mode])


def test_bench_multi_paths_on_one_gpu():
    """bench.py --gpus 2 as ONE process over two logical shards, and under a 1-rank 'launcher' environment."""
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--logical-shards", "--steps", "3",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:500]          # ONE JSON line: RCCL's banner went to stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 1e7 and len(line["multi_gpu"]["grid_ms_per_shard"]) == 2
    # one-process-per-GPU path with a world of one: RCCL communicator through the temp-file rendezvous, all-gather, max
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-comm", "--steps", "3",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:500]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e7 and "rank0_kernel_ms" in line["multi_gpu"]


def test_more_shards_than_patients():
    """An 8-GPU node and a cohort of 3 patients: the surplus shards own empty row ranges (found by tools/fuzz_multi.py)."""
    from pilot_amd.synthetic import make_problem
    P, M = make_problem(3, 6, 4, seed=9, cells_per_patient=50)
    ref = engine.sinkhorn_grid(P, M, 0.1)
    for G in (4, 8):
        np.testing.assert_array_equal(multi.sinkhorn_grid_multi(P, M, 0.1, devices=[0] * G), ref)
        np.testing.assert_array_equal(multi.emd_grid_multi(P, M, devices=[0] * G), engine.emd_grid(P, M))


def test_unnormalised_cost_takes_the_same_kernels_on_one_and_on_several_devices():
    """the device entry point assumes cost / max; the multi-device forms must take the single-device host entry
    point's decisions from max(M) of the inputs -- max(M)/reg = 1000 runs the POT-literal kernel in both, whatever precision
    was asked for, and an explicit f16x2 outside its range becomes bf16x3 in both."""
    P, M = make_problem(12, 8, 4, seed=5, cells_per_patient=300)
    for scale, reg, prec in ((5.0, 0.005, "auto"), (5.0, 0.005, "fp64"), (5.0, 0.2, "f16x2"), (3.0, 0.1, "auto")):
        ref, ri = engine.sinkhorn_grid(P, scale * M, reg, precision=prec, return_info=True)
        got, gi = multi.sinkhorn_grid_multi(P, scale * M, reg, devices=[0, 0, 0], precision=prec, return_info=True)
        np.testing.assert_array_equal(got, ref)
        np.testing.assert_array_equal(gi["iters"], ri["iters"])
        np.testing.assert_array_equal(gi["flags"], ri["flags"])
        assert np.isfinite(ref).all()
        mp = multi.MultiPlan(P, scale * M, devices=[0, 0])
        mp.sinkhorn(reg, precision=prec)
        np.testing.assert_array_equal(mp.fetch(), ref)
        mp.close()


def test_shard_threads_serial_switch_and_back_to_back_calls(switches):
    """Every shard is enqueued by its own host thread; PILOT_OT_MULTI_SERIAL=1 enqueues from the calling thread instead.
    Same bits either way, and back-to-back asynchronous calls with the peer-copy gather (the next call's kernels must wait
    for shard 0's copies of the previous rows) leave the right matrix after every call."""
    P, M = make_problem(**CONFIGS["c2"])
    ref1 = engine.sinkhorn_grid(P, M, 0.1)
    ref2 = engine.sinkhorn_grid(P, M, 0.5)
    for serial in ("0", "1"):
        switches.setenv("PILOT_OT_MULTI_SERIAL", serial)
        mp = multi.MultiPlan(P, M, devices=[0] * 5)
        for _ in range(6):
            mp.sinkhorn(0.5)
            mp.sinkhorn(0.1)
        np.testing.assert_array_equal(mp.fetch(), ref1)
        mp.sinkhorn(0.5)
        np.testing.assert_array_equal(mp.fetch(), ref2)
        mp.emd()
        np.testing.assert_array_equal(mp.fetch(), engine.emd_grid(P, M))
        mp.close()


def test_rccl_reports_its_rank_count():
    """ncclCommCount / ncclCommUserRank are echoed so that a bench record can prove how many ranks RCCL saw."""
    P, M = make_problem(**CONFIGS["c1"])
    mp = multi.MultiPlan(P, M, devices=[0], gather="rccl")
    assert mp.rccl_info() == ([1], [0])
    mp.close()
    mp = multi.MultiPlan(P, M, devices=[0, 0], gather="copy")
    assert mp.rccl_info() == ([0, 0], [-1, -1])
    mp.close()
    comm = multi.Comm(0, 1, key="pytest_info_%d" % os.getpid())
    assert comm.info() == (1, 0)
    comm.close()


def test_errors_on_a_shard_thread_reach_the_caller():
    P, M = make_problem(**CONFIGS["c1"])
    mp = multi.MultiPlan(P, M, devices=[0, 0])
    with pytest.raises(ValueError, match="tau"):
        mp.sinkhorn(0.1, tau=0.5)
    mp.close()
