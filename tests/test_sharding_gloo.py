"""Multi-GPU row sharding, exercised with world_size-2 gloo on the CPU (the per-rank compute is
stood in for by the oracle -- the test is about the deal / all-gather / interleave / mirror)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from pilot_amd import sharding


def test_shard_rows_cover_every_row_once():
    for N in (1, 2, 7, 20, 600):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                b, e, s = sharding.shard_rows(N, r, world)
                rows += list(range(b, e, s))
                assert sharding.n_local_rows(N, r, world) == len(range(b, e, s)) <= sharding.n_padded_rows(N, world)
            assert sorted(rows) == list(range(N))


def test_interleave_undoes_the_deal():
    N, world = 11, 4
    full = np.arange(N * N, dtype=np.float64).reshape(N, N)
    n_pad = sharding.n_padded_rows(N, world)
    g = np.zeros((world, n_pad, N))
    for r in range(world):
        blk = full[r::world]
        g[r, :blk.shape[0]] = blk
    np.testing.assert_array_equal(sharding.interleave(g, N, world), full)


def test_mirror_upper():
    rng = np.random.default_rng(0)
    A = rng.random((6, 6)); S = np.triu(A) + np.triu(A, 1).T
    np.testing.assert_array_equal(sharding.mirror_upper(np.triu(A)), S)


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, %(root)r + "/tests")
    from oracle import oracle as O
    import gloo_harness as harness
    from pilot_amd.synthetic import make_problem
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    P, M = make_problem(23, 9, 6, seed=5, cells_per_patient=150)     # N=23: ragged over 2 ranks
    N = P.shape[0]
    full = harness.grid_sharded(lambda b, e, s: O.sinkhorn_grid(P, M, 0.1, row_begin=b, row_end=e, row_step=s),
                                 N, rank, world)
    ref = O.sinkhorn_grid(P, M, 0.1)
    assert full.shape == (N, N) and np.array_equal(full, ref), "sinkhorn shards differ"
    def upper(b, e, s):
        E = O.emd_grid(P, M, row_begin=b, row_end=e, row_step=s)
        rows = np.arange(b, e, s)[:, None]
        return np.where(np.arange(N)[None, :] >= rows, E, 0.0)
    full = harness.grid_sharded(upper, N, rank, world, symmetric_upper=True)
    ref = O.emd_grid(P, M)
    assert np.allclose(full, ref, atol=1e-14) and np.array_equal(full, full.T), "emd shards differ"
    t = torch.from_numpy(ref[rank::world].copy())
    again = harness.all_gather_rows(t, N)
    assert torch.equal(again, torch.from_numpy(ref))
    dist.barrier()
    if rank == 0:
        print("SHARDING_OK")
    dist.destroy_process_group()
""")


def test_world_size_two_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDING_OK" in out.stdout
