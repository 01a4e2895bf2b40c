"""The property the hash-grouped fold of lii_map_incremental rests on (lii_map.hip: AddHash): what
KD_TREE::Add_Points(points, downsample_on = true) leaves in the map does not depend on the ORDER of the batch - only the counter it
returns does.  Checked on the oracle's restated tree and, where it was built, on the UNMODIFIED reference ikd-Tree (oracle/_ref):
the same batch in three different orders gives the same point set.  (Ties of the squared distance to a voxel centre are broken by
batch order; random float coordinates do not tie.)  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O


def _sets(backend, base, batch, orders, box):
    out = []
    counters = []
    for perm in orders:
        t = O.Tree(backend, downsample=box)
        t.build(base)
        counters.append(t.add_points(batch[perm], True))
        pts = t.flatten()
        out.append(pts[np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))])
        t.close()
    return out, counters


@pytest.mark.parametrize("backend", ["oracle", "ref"])
def test_add_points_result_is_independent_of_the_batch_order(backend):
    if backend == "ref" and not O.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(11)
    box = 0.2
    base = rng.uniform(-3, 3, (6000, 3)).astype(np.float32)      # ~ 0.2 points per down-sample box: empty, single and shared boxes
    batch = np.concatenate([rng.uniform(-3, 3, (4000, 3)),       # spread out ...
                            rng.uniform(-0.5, 0.5, (3000, 3))]).astype(np.float32)  # ... and crowded (many batch points per box)
    n = len(batch)
    orders = [np.arange(n), np.arange(n)[::-1].copy(), rng.permutation(n)]
    sets, counters = _sets(backend, base, batch, orders, box)
    assert len(sets[0]) > 0
    for s in sets[1:]:
        assert s.shape == sets[0].shape
        assert np.array_equal(s, sets[0])
    # the counter Add_Points returns DOES depend on the order (it counts the running replacements) - which is why
    # lii_map_add_points, which reports it, keeps the sorted fold
    assert len(set(counters)) > 1
