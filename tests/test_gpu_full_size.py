"""BASELINE.json full-size configurations (OS1-128: 131 072 points; dense: ~500 k points; both against a 1 M-point map)
checked through size-independent properties, plus an exact comparison on a random sample of queries, plus the error
behaviour of the C-ABI."""
import numpy as np
import pytest

from conftest import make_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    import lidar_imu_init_amd as lii
    from harness import synth
    hall, map_pts = synth.bench_world(1_000_000, 0.15)
    reg = lii.Registrar(max_scan_points=520_000, max_map_points=1_100_000, filter_size_map=0.15)
    reg.map_build(map_pts)
    yield hall, map_pts, reg
    reg.close()


@pytest.mark.parametrize("sensor", ["os1_128", "dense500k"])
def test_full_size_properties(big, oracle, sensor):
    import lidar_imu_init_amd as lii
    from harness import sharding
    from harness import synth
    hall, map_pts, reg = big
    R = synth.rot_zyx(0.01, -0.02, 0.8)
    p = np.array([4.0, -3.0, 0.3])
    scan = synth.make_scan(hall, sensor, R, p, noise=0.02, seed=77)
    n = len(scan)
    assert n > (120_000 if sensor == "os1_128" else 450_000)
    st_true = make_state(oracle, R, p)
    st0 = oracle.state_boxplus(st_true, np.r_[0.003, -0.002, 0.004, 0.03, -0.02, 0.01, np.zeros(18)])
    reg.scan_upload(scan)
    assert reg.downsample_skip() == n
    out = reg.iekf_iterate(lii.State(st0), True, True)
    nb, cnt, sel = reg.neighbors(n)
    world = reg.scan_download(2)[:, :3]
    # (1) every neighbour is a map point; distances ascend, respect d2 <= 5 and equal the float32 reference formula
    full = cnt == 5
    assert full.mean() > 0.99
    d = nb[full] - world[full][:, None, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    assert np.all(np.diff(d2, axis=1) >= 0) and d2.max() <= 5.0
    keys = {tuple(x) for x in map(tuple, map_pts[:: 97].view(np.uint32))}  # membership of a slice of the map, both ways
    flat = nb[full].reshape(-1, 3)
    sample = flat[:: max(1, len(flat) // 20000)]
    mapset = set(map(bytes, np.ascontiguousarray(map_pts)))
    assert all(bytes(x) in mapset for x in np.ascontiguousarray(sample))
    assert len(keys) > 0
    # (2) idempotence: a second search at the same state reproduces neighbours and sums bit for bit
    out_b = reg.iekf_iterate(lii.State(st0), True, True)
    nb_b, cnt_b, sel_b = reg.neighbors(n)
    assert np.array_equal(nb, nb_b) and np.array_equal(cnt, cnt_b) and np.array_equal(sel, sel_b) and np.array_equal(out, out_b)
    # (3) exact agreement with the oracle's tree on a random sample of the queries
    tree = oracle.Tree("oracle")
    tree.build(map_pts)
    idx = np.random.default_rng(1).choice(n, 4000, replace=False)
    pts, rd2, rc = tree.knn(world[idx], threads=4)
    assert np.array_equal(cnt[idx], rc)
    m = rc == 5
    assert np.array_equal(nb[idx][m], pts[m])
    # (3b) ... and with the UNMODIFIED reference ikd-Tree (oracle/_ref) on EVERY query of the OS1-128 scan against the 1 M-point map
    if sensor == "os1_128" and oracle.ref_available():
        rt = oracle.Tree("ref")
        rt.build(map_pts)
        rp, _, rcnt = rt.knn(world, threads=8)
        rt.close()
        assert np.array_equal(cnt, rcnt)
        m = rcnt == 5
        assert np.array_equal(nb[m], rp[m])
    # (4) linearity over shards: the normal equations of two halves add up to those of the whole scan
    total = np.zeros(91)
    for r in range(2):
        lo, hi = sharding.shard_bounds(n, 2, r)
        reg.scan_upload(scan[lo:hi])
        reg.downsample_skip()
        total += reg.iekf_iterate(lii.State(st0), True, True)
    assert total[90] == out[90]
    assert np.max(np.abs(total[:90] - out[:90])) <= 1e-11 * np.max(np.abs(out[:90]))
    # (5) the full update recovers the true pose
    reg.scan_upload(scan)
    reg.downsample_skip()
    s = lii.State(st0)
    rep = reg.iekf_update(s, lii.State(st0), max_iterations=5, imu_en=False)
    assert rep["effect_num"] > 0.9 * n
    assert np.linalg.norm(s.pos_end - p) < 0.005
    assert np.linalg.norm(oracle.log_so3(R.T @ s.rot_end)) < 5e-4


def test_error_behaviour_and_edge_cases(oracle):
    import lidar_imu_init_amd as lii
    reg = lii.Registrar(max_scan_points=1000, max_map_points=2000, filter_size_map=0.2)
    st = lii.State()
    # registration before any scan / before any search is a call-order error, not a crash
    with pytest.raises(lii.LIIError) as e:
        reg.iekf_iterate(st, True, False)
    assert e.value.code == -5
    reg.scan_upload(np.zeros((10, 4), np.float32))
    reg.downsample_skip()
    with pytest.raises(lii.LIIError) as e:
        reg.iekf_iterate(st, False, False)
    assert e.value.code == -5
    # capacity errors
    with pytest.raises(lii.LIIError) as e:
        reg.scan_upload(np.zeros((1001, 4), np.float32))
    assert e.value.code == -4
    with pytest.raises(lii.LIIError) as e:
        reg.map_build(np.zeros((2001, 3), np.float32))
    assert e.value.code == -4
    # an EMPTY map: nothing is selected, the normal equations are zero and the update returns the prior unchanged
    reg.map_reset()
    assert reg.map_size() == 0
    scan = np.random.default_rng(0).uniform(-5, 5, (500, 4)).astype(np.float32)
    reg.scan_upload(scan)
    reg.downsample_skip()
    out = reg.iekf_iterate(st, True, False)
    assert np.all(out == 0)
    s = lii.State()
    rep = reg.iekf_update(s, lii.State(), max_iterations=4, imu_en=False)
    assert rep["effect_num"] == 0 and np.allclose(s.pod[:36], lii.State().pod[:36])
    # fewer than 5 map points: nobody gets 5 neighbours
    reg.map_build(np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0]], np.float32))
    out = reg.iekf_iterate(st, True, False)
    nb, cnt, sel = reg.neighbors(500)
    assert cnt.max() <= 3 and sel.sum() == 0 and out[90] == 0
    # a single-point scan
    reg.scan_upload(np.array([[0.05, 0.02, 0.01, 0.0]], np.float32))
    assert reg.downsample(0.05)[0] == 1
    reg.iekf_iterate(st, True, False)
    reg.close()
