"""Row sharding of the N x N pair grid over the GPUs of one node (one process per GPU).

The pair problems are independent given the replicated N x K proportions and K x K cost
(pilotpy/tools/Trajectory.py:505-515), so the only exchange step is assembling the finished
distance matrix: ONE all-gather of each rank's row block (RCCL over xGMI when the blocks live in
HBM -- ``torch.distributed`` backend "nccl" -- or gloo for host blocks / CPU tests).

Rows are dealt ROUND-ROBIN (row r -> rank r mod world): Sinkhorn iteration counts are ragged and,
in exact mode with a symmetric cost, only columns >= row are solved, so contiguous blocks would be
unbalanced; cyclic rows balance both.  ``torch`` is used only for the collective.
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
    """Undo the round-robin deal: ``gathered`` is (world, n_pad, N) with gathered[w, t] = row w + t*world.
    Works on numpy arrays and torch tensors alike (pure indexing)."""
    n_pad = gathered.shape[1]
    # (world, n_pad, N) -> (n_pad, world, N) -> (n_pad*world, N): row index t*world + w
    full = gathered.transpose(1, 0, 2) if isinstance(gathered, np.ndarray) else gathered.permute(1, 0, 2)
    full = full.reshape(n_pad * world, N)
    return full[:N]


def all_gather_rows(local, N: int, group=None):
    """All-gather the per-rank row blocks into the full N x N matrix on every rank.

    ``local``: torch tensor (n_local, N) -- on the GPU (nccl/RCCL) or the host (gloo).  Returns a
    torch tensor (N, N) on the same device.  With no process group initialised (single GPU) the
    block is returned as is.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n_pad = n_padded_rows(N, world)
    if local.shape[0] < n_pad:
        pad = torch.zeros((n_pad - local.shape[0], N), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    gathered = torch.empty((world * n_pad, N), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)      # rank-major concatenation
    return interleave(gathered.view(world, n_pad, N), N, world).contiguous()


def mirror_upper(E):
    """Fill the strictly-lower triangle from the upper one (exact mode, symmetric cost)."""
    if isinstance(E, np.ndarray):
        iu = np.triu_indices(E.shape[0], 1)
        E = E.copy()
        E.T[iu] = E[iu]
        return E
    import torch
    return torch.triu(E) + torch.triu(E, 1).transpose(0, 1)


def grid_sharded(compute_rows, N: int, rank: int, world: int, group=None, symmetric_upper=False):
    """Run ``compute_rows(row_begin, row_end, row_step) -> (n_local, N)`` on this rank's rows and
    assemble the full matrix on every rank.  ``compute_rows`` returns a torch tensor or a numpy
    array (converted for the collective).  ``symmetric_upper``: blocks hold only columns >= row."""
    import torch
    rb, re_, rs = shard_rows(N, rank, world)
    local = compute_rows(rb, re_, rs)
    as_numpy = isinstance(local, np.ndarray)
    if as_numpy:
        local = torch.from_numpy(np.ascontiguousarray(local))
    full = all_gather_rows(local, N, group=group)
    if symmetric_upper:
        full = mirror_upper(full)
    return full.numpy() if as_numpy else full
