"""world_size-2 (and 3) CPU test of the multi-GPU path: shard the points of a scan, evaluate each block independently
(here with the oracle standing in for the per-rank kernels), all-reduce the 91 scalars with torch.distributed/gloo and
compare with the unsharded evaluation.  The reduction is the only exchange step of the sharded path."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, by_voxel=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from conftest import make_state
    from harness import sharding
    from harness import synth
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hall = synth.Hall(size=(20.0, 16.0, 6.0), n_boxes=6, seed=3)
    map_pts = hall.surface_points(0.15, noise=0.01, seed=3)
    R = synth.rot_zyx(0.02, -0.01, 0.3)
    p = np.array([0.5, -0.4, 0.1])
    scan = synth.make_scan(hall, "tiny", R, p, noise=0.02, seed=5)
    st = O.state_boxplus(make_state(O, R, p), np.r_[0.004, -0.003, 0.005, 0.03, -0.02, 0.015, np.zeros(18)])
    tree = O.Tree("oracle")
    tree.build(map_pts)  # the map is replicated on every rank
    # every rank voxel-filters the WHOLE scan (replicated, so no voxel is ever split between ranks - SURVEY.md section 8e)
    # and registers its contiguous block of the down-sampled cloud: what libliinit_hip does with a communicator attached
    body = O.voxel_grid(scan, 0.1)[0]
    lo, hi = sharding.shard_bounds(len(body), world, rank)
    mine = body[lo:hi]
    if by_voxel:
        # ... or, split by voxel, the voxels whose key hashes to this rank: a rank needs the POINTS of its voxels only, so it
        # filters the part of the scan that falls into them - and gets the centroids the whole filter gives those voxels
        owner = sharding.voxel_rank(sharding.voxel_keys(scan, 0.1), world)
        mine = O.voxel_grid(scan[owner == rank], 0.1)[0]
        lo, hi = 0, len(mine)
    local = tree.iterate_once(mine, st, search=True, imu_en=True)
    total = sharding.all_reduce_normal_equations(local["out91"])
    full = tree.iterate_once(body, st, search=True, imu_en=True)
    q.put((rank, lo, hi, total, full["out91"], local["selected"].sum()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("by_voxel", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_normal_equations_match_unsharded(world, by_voxel):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, by_voxel)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    res.sort()
    # the blocks tile [0, n) and every rank holds the same reduced system, equal to the unsharded one
    if not by_voxel:
        assert res[0][1] == 0 and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    n_sel = sum(r[5] for r in res)
    for rank, lo, hi, total, full, _ in res:
        assert int(round(total[90])) == int(round(full[90])) == n_sel
        assert np.max(np.abs(total[:90] - full[:90])) <= 1e-12 * np.max(np.abs(full[:90]))
        assert np.array_equal(total, res[0][3])


def test_shard_bounds_properties():
    from harness import sharding
    for n in (0, 1, 7, 100_000, 131_072):
        for w in (1, 2, 3, 4, 8):
            b = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    # the split by voxel: every voxel has one owner, the shares are even, and the bound the launches are sized for holds
    rng = np.random.default_rng(5)
    keys = sharding.voxel_keys(rng.uniform(-40, 40, size=(200_000, 3)).astype(np.float32), 0.1)
    uniq = np.unique(keys)
    for w in (2, 3, 4, 8):
        owner = sharding.voxel_rank(uniq, w)
        assert owner.min() == 0 and owner.max() == w - 1
        share = np.bincount(owner, minlength=w)
        assert np.all(np.abs(share - len(uniq) / w) < 0.03 * len(uniq) / w)
        assert share.max() <= sharding.voxel_partition_bound(len(uniq), w)
    assert int(sharding.voxel_rank(np.array([0x40000100000], np.uint64), 8)[0]) in range(8)
    assert sharding.voxel_partition_bound(100_000, 1) == 100_000 and sharding.voxel_partition_bound(1000, 8) == 1000
    H, g, m = sharding.unpack_normal_equations(np.arange(91.0))
    assert H.shape == (12, 12) and np.allclose(H, H.T) and H[0, 11] == 11 and H[1, 1] == 12 and m == 90
