"""Point sharding of one scan across ranks: every rank voxel-filters the WHOLE scan (replicated), the
down-sampled cloud is split into contiguous blocks, the local map is replicated, every rank evaluates its block, and ONE sum
(fp64) of the 91 normal-equation scalars per IEKF iteration joins them.  On GPUs both the split (shard_range in
csrc/lii_device.h - the same arithmetic as shard_bounds below) and the exchange (node-local mailbox or RCCL, lii_comm_init)
happen inside libliinit_hip; this module is the host-side statement of the bookkeeping and a torch.distributed form of the
reduction, exercised by the CPU (gloo) tests with the oracle standing in for the kernels."""
from __future__ import annotations

import numpy as np


def shard_bounds(n_points: int, world_size: int, rank: int):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one point and cover [0, n) exactly."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    return (n_points * rank) // world_size, (n_points * (rank + 1)) // world_size


def all_reduce_normal_equations(local91, group=None):
    """Sum of the 91 scalars (78 upper-triangle H^T R^-1 H, 12 H^T R^-1 z, effective-point count) over the ranks."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local91, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


def unpack_normal_equations(ne91):
    """(H^T R^-1 H as a full 12 x 12, H^T R^-1 z, effective-point count)."""
    ne91 = np.asarray(ne91, dtype=np.float64)
    H = np.zeros((12, 12))
    iu = np.triu_indices(12)
    H[iu] = ne91[:78]
    H = H + H.T - np.diag(np.diag(H))
    return H, ne91[78:90].copy(), int(round(ne91[90]))
