"""(test support, not product: the library splits a scan itself - lii_comm_set_partition - and this restates that split on the host for tests/test_sharding_gloo.py)
Point sharding of one scan across ranks: every rank voxel-filters the WHOLE scan (replicated), the
down-sampled cloud is split into contiguous blocks (or, lii_comm_set_partition(h, 2), every rank filters the voxels whose key
hashes to it: voxel_keys / voxel_rank below), the local map is replicated, every rank evaluates its block, and ONE sum
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


def voxel_keys(xyz, leaf: float):
    """The key the fused voxel filter files a point under (vhash_insert_abs, csrc/lii_scan.hip): three 21-bit fields of
    floor(coordinate * (1 / leaf)) in float arithmetic around a bias of 2^20.  (Points beyond +- 1 048 000 voxels get a key of
    their own on the device and are not covered here.)"""
    xyz = np.asarray(xyz, np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    f = np.floor(xyz[:, :3] * inv).astype(np.int64) + (1 << 20)
    return (f[:, 2].astype(np.uint64) << np.uint64(42)) | (f[:, 1].astype(np.uint64) << np.uint64(21)) | f[:, 0].astype(np.uint64)


def voxel_rank(keys, world_size: int):
    """Which rank of a job split by voxel (lii_comm_set_partition(h, 2)) owns a voxel: the upper half of the key's 64-bit mix
    scaled to [0, world) - voxel_rank / vh_mix in csrc/lii_scan.hip, the same arithmetic."""
    k = np.asarray(keys, np.uint64).copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xFF51AFD7ED558CCD)
        k ^= k >> np.uint64(33)
    return (((k >> np.uint64(32)) * np.uint64(world_size)) >> np.uint64(32)).astype(np.int64)


def voxel_partition_bound(n_points: int, world_size: int) -> int:
    """What a rank's share of the down-sampled cloud may hold (voxel_partition_bound, csrc/lii_scan.hip)."""
    if world_size <= 1:
        return n_points
    return min(n_points, n_points // world_size + n_points // (4 * world_size) + 2048)


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
