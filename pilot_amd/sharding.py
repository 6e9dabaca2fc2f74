"""Row sharding of the N x N pair grid over the GPUs of one node (one process per GPU).

The pair problems are independent given the replicated N x K proportions and K x K cost
(pilotpy/tools/Trajectory.py:505-515), so the only exchange step is assembling the finished
distance matrix: ONE all-gather of each rank's row block (RCCL over xGMI inside libpilot_ot.so:
``pilot_ot_multi_*`` / ``pilot_ot_comm_*``).  This module is the index arithmetic of that deal, shared
by ``bench.py`` and by the world-2 gloo test (whose torch harness lives in ``tests/gloo_harness.py``).

Rows are dealt ROUND-ROBIN (row r -> rank r mod world): Sinkhorn iteration counts are ragged and,
in exact mode with a symmetric cost, only columns >= row are solved, so contiguous blocks would be
unbalanced; cyclic rows balance both.  No torch here.
"""
from __future__ import annotations

import numpy as np


def shard_rows(N: int, rank: int, world: int):
    """(row_begin, row_end, row_step) of the rows owned by ``rank``: rank, rank+world, ... < N."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return min(rank, N), N, world          # (more ranks than rows: the surplus ranks own the empty range [N, N))


def n_local_rows(N: int, rank: int, world: int) -> int:
    return len(range(rank, N, world))


def n_padded_rows(N: int, world: int) -> int:
    """Rows every rank contributes to the all-gather (ranks short of a row pad with zeros)."""
    return (N + world - 1) // world


def interleave(gathered, N: int, world: int):
    """Undo the round-robin deal: ``gathered`` is (world, n_pad, N) with gathered[w, t] = row w + t*world."""
    n_pad = gathered.shape[1]
    # (world, n_pad, N) -> (n_pad, world, N) -> (n_pad*world, N): row index t*world + w
    return np.asarray(gathered).transpose(1, 0, 2).reshape(n_pad * world, N)[:N]


def mirror_upper(E):
    """Fill the strictly-lower triangle from the upper one (exact mode, symmetric cost)."""
    iu = np.triu_indices(E.shape[0], 1)
    E = np.array(E, copy=True)
    E.T[iu] = E[iu]
    return E
