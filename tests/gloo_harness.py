"""torch.distributed harness of the world-2 gloo test (test infrastructure: the product assembles the matrix with RCCL inside
libpilot_ot.so and never imports torch).  The deal / interleave arithmetic under test is pilot_amd.sharding's."""
import numpy as np
import torch
import torch.distributed as dist

from pilot_amd import sharding


def all_gather_rows(local, N, group=None):
    """All-gather the per-rank row blocks (torch tensor (n_local, N), host memory for gloo) into the full N x N matrix."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n_pad = sharding.n_padded_rows(N, world)
    if local.shape[0] < n_pad:
        local = torch.cat([local, torch.zeros((n_pad - local.shape[0], N), dtype=local.dtype)], dim=0)
    local = local.contiguous()
    gathered = torch.empty((world * n_pad, N), dtype=local.dtype)
    dist.all_gather_into_tensor(gathered, local, group=group)      # rank-major concatenation
    return torch.from_numpy(np.ascontiguousarray(sharding.interleave(gathered.view(world, n_pad, N).numpy(), N, world)))


def grid_sharded(compute_rows, N, rank, world, group=None, symmetric_upper=False):
    """Run ``compute_rows(row_begin, row_end, row_step) -> ndarray (n_local, N)`` on this rank's rows and assemble the full
    matrix on every rank; ``symmetric_upper``: blocks hold only columns >= row."""
    rb, re_, rs = sharding.shard_rows(N, rank, world)
    local = torch.from_numpy(np.ascontiguousarray(compute_rows(rb, re_, rs)))
    full = all_gather_rows(local, N, group=group).numpy()
    return sharding.mirror_upper(full) if symmetric_upper else full
