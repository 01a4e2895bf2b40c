"""The ingest restatement (oracle/orc_ingest.hpp <- reference src/preprocess.cpp:50-335) against an independent,
vectorised numpy formulation of the same rules and against small hand-worked cases."""
import numpy as np
import pytest

from harness import synth, wire


def np_cut(pts, stamp_s, required, scan_count, uncut_below):
    """Sort + cut expressed with array operations (no running loop): boundaries b_c = int((c+1)*size/req) - 1."""
    order = np.argsort(pts[:, 3], kind="stable")
    p = pts[order].copy()
    size = len(p)
    req = 1 if scan_count < uncut_below else required
    frames, lfe, start, c = [], stamp_s * 1000.0, 1, 0
    while True:
        b = int(((c + 1) * size) // req) - 1
        if b < start or b > size - 1:
            break
        seg = p[start:b + 1].copy()
        delta = stamp_s * 1000.0 - lfe
        seg[:, 3] = (seg[:, 3].astype(np.float64) + delta).astype(np.float32)
        frames.append((lfe, seg))
        lfe = lfe + float(seg[-1, 3])
        start, c = b + 1, c + 1
    return frames


def np_decode_pcl2(lidar_type, raw, n, stamp_s, n_scans, pfn, blind):
    a = np.frombuffer(raw, wire.DTYPES[lidar_type], count=n)
    x, y, z = a["x"], a["y"], a["z"]
    if lidar_type == wire.VELO:
        t = (a["time"].astype(np.float64) * 1000.0).astype(np.float32)
    elif lidar_type == wire.OUSTER:
        t = (a["t"].astype(np.float64) / 1e6).astype(np.float32)
    elif lidar_type == wire.PANDAR:
        t = ((a["timestamp"] - a["timestamp"][0]) * 1000).astype(np.float32)
    else:
        t = ((a["timestamp"] - stamp_s + 0.1) * 1000.0).astype(np.float32)
    d = (x * x + y * y) + z * z  # float32
    ok = ~((d.astype(np.float64) < blind * blind) | np.isnan(x) | np.isnan(y) | np.isnan(z))
    keep = ok & (np.arange(n) % pfn == 0) & (a["ring"].astype(np.int64) < n_scans)
    return np.stack([x, y, z, t], 1)[keep]


def assert_frames_equal(got, want):
    assert len(got) == len(want)
    for (tb_a, pa), (tb_b, pb) in zip(got, want):
        assert tb_a == tb_b
        assert pa.shape == pb.shape and np.array_equal(pa.view(np.uint32), pb.view(np.uint32))


@pytest.fixture(scope="module")
def sweep():
    hall = synth.Hall()
    R = synth.rot_zyx(0.01, -0.02, 0.5)
    return wire.raw_sweep(hall, "mid16k", R, np.array([1.0, -2.0, 0.1]))


@pytest.mark.parametrize("lidar_type", [wire.VELO, wire.OUSTER, wire.PANDAR, wire.ROBOSENSE])
@pytest.mark.parametrize("cut,scan_count,pfn", [(1, 100, 1), (3, 100, 2), (5, 100, 3), (3, 7, 1)])
def test_pcl2_against_numpy(oracle, sweep, lidar_type, cut, scan_count, pfn):
    xyz, ring, t_ms = sweep
    stamp = 1_650_000_123.25
    raw = wire.pack_pcl2(lidar_type, xyz, ring, t_ms, stamp)
    n = len(xyz)
    got = oracle.ingest_pcl2(raw, n, wire.pc2_fields(lidar_type), lidar_type, 24, pfn, 1.5, stamp, cut, scan_count)
    pts = np_decode_pcl2(lidar_type, raw, n, stamp, 24, pfn, 1.5)
    want = np_cut(pts, stamp, cut, scan_count, 20)
    assert len(want) == (1 if scan_count < 20 else cut)
    assert_frames_equal(got, want)
    # every kept point but the time-earliest one is emitted exactly once; frame times chain
    assert sum(len(p) for _, p in got) == len(pts) - 1
    for k in range(1, len(got)):
        assert got[k][0] == got[k - 1][0] + float(got[k - 1][1][-1, 3])
        assert got[k][1][:, 3].min() >= -1e-3  # stamp * 1000 in double has an ulp of 2.4e-4 ms at epoch times


def test_livox_against_numpy(oracle):
    hall = synth.Hall()
    raw, n = wire.avia_sweep(hall, synth.rot_zyx(0, 0, 0.3), np.array([0.5, 0.5, 0.0]), n_points=6000)
    stamp, pfn, blind, n_scans = 12.5, 2, 1.0, 6
    a = np.frombuffer(raw, wire.LIVOX_DTYPE, count=n)
    idx = np.arange(n)
    v = (idx >= 1) & (a["line"] < n_scans) & (((a["tag"] & 0x30) == 0x10) | ((a["tag"] & 0x30) == 0x00))
    cnt = np.cumsum(v)
    pop = v & (cnt % pfn == 0)
    x, y, z = a["x"], a["y"], a["z"]
    t = a["offset_time"].astype(np.float32) / np.float32(1000000)
    d = (x * x + y * y) + z * z
    prev = np.zeros((n, 3), np.float32)
    prev[1:] = np.where(pop[:-1, None], np.stack([x, y, z], 1)[:-1], 0)
    differs = (np.abs(x - prev[:, 0]).astype(np.float64) > 1e-7) | (np.abs(y - prev[:, 1]).astype(np.float64) > 1e-7) | \
              (np.abs(z - prev[:, 2]).astype(np.float64) > 1e-7)
    keep = pop & ~(d.astype(np.float64) < blind * blind) & differs
    pts = np.stack([x, y, z, t], 1)[keep]
    assert 1000 < len(pts) < n // pfn
    for cut, sc in [(1, 100), (5, 100), (5, 3)]:
        got = oracle.ingest_livox(raw, n, wire.livox_fields(), n_scans, pfn, blind, stamp, cut, sc)
        assert_frames_equal(got, np_cut(pts, stamp, cut, sc, 5))


def test_hand_worked_cut(oracle):
    """7 kept points, cut into 3: boundaries int(7/3)-1 = 1, int(14/3)-1 = 3, int(21/3)-1 = 6; the time-earliest point
    (sorted index 0) is never emitted; frame k starts at stamp + time of the previous boundary point."""
    n = 7
    t_ms = np.array([30, 0, 10, 20, 60, 40, 50], np.float64)
    xyz = np.stack([np.full(n, 5.0), np.arange(n, dtype=np.float64), np.zeros(n)], 1).astype(np.float32)
    raw = wire.pack_pcl2(wire.OUSTER, xyz, np.zeros(n, np.int32), t_ms, 100.0)
    got = oracle.ingest_pcl2(raw, n, wire.pc2_fields(wire.OUSTER), wire.OUSTER, 16, 1, 0.5, 100.0, 3, 50)
    assert [len(p) for _, p in got] == [1, 2, 3]
    assert [tb for tb, _ in got] == [100000.0, 100010.0, 100030.0]
    assert np.array_equal(got[0][1][:, 3], [10]) and np.array_equal(got[0][1][:, 1], [2])
    assert np.array_equal(got[1][1][:, 3], [10, 20]) and np.array_equal(got[1][1][:, 1], [3, 0])
    assert np.array_equal(got[2][1][:, 3], [10, 20, 30]) and np.array_equal(got[2][1][:, 1], [5, 6, 4])


def test_time_synthesis_and_edge_cases(oracle):
    # no per-point time: one ring, azimuth decreasing by 36 deg per point -> (yaw_first - yaw) / 3.61 ms, first point dropped
    k = np.arange(10)
    yaw = np.deg2rad(170.0 - 36.0 * k)
    xyz = np.stack([4 * np.cos(yaw), 4 * np.sin(yaw), np.zeros(10)], 1).astype(np.float32)
    raw = wire.pack_pcl2(wire.VELO, xyz, np.zeros(10, np.int32), np.zeros(10), 5.0, with_time=False)
    got = oracle.ingest_pcl2(raw, 10, wire.pc2_fields(wire.VELO), wire.VELO, 16, 1, 0.5, 5.0, 1, 100)
    assert len(got) == 1 and len(got[0][1]) == 8  # 10 - ring seed - time-earliest
    exp = (36.0 * np.arange(2, 10)) / 3.61
    assert np.allclose(got[0][1][:, 3], exp, rtol=1e-5)
    # clouds too small for the requested cut never reach a boundary: no frame at all (size 2, 3 cuts -> int(2/3)-1 < 1)
    raw = wire.pack_pcl2(wire.OUSTER, xyz[:2], np.zeros(2, np.int32), np.array([0.0, 1.0]), 5.0)
    assert oracle.ingest_pcl2(raw, 2, wire.pc2_fields(wire.OUSTER), wire.OUSTER, 16, 1, 0.5, 5.0, 3, 100) == []
    # empty message, everything inside the blind zone
    assert oracle.ingest_pcl2(b"", 0, wire.pc2_fields(wire.OUSTER), wire.OUSTER, 16, 1, 0.5, 5.0, 3, 100) == []
    raw = wire.pack_pcl2(wire.OUSTER, xyz, np.zeros(10, np.int32), k * 1.0, 5.0)
    assert oracle.ingest_pcl2(raw, 10, wire.pc2_fields(wire.OUSTER), wire.OUSTER, 16, 1, 10.0, 5.0, 1, 100) == []


# ----------------------------------------------------------------------------------------------------------------------
# Pinning against the reference's own code: oracle/_ref/libref_preprocess.so is the UNMODIFIED src/preprocess.cpp built
# out-of-tree (oracle/Makefile `ref`).  Where it is absent the committed fixture (tests/golden/ingest, written by
# tests/golden/make_ingest_fixture.py from the same library) takes its place.
def assert_frames_equal_mod_ties(got, ref):
    """Counts, frame times and the per-frame time sequences must be identical; a point whose time stamp is unique in its
    frame must sit at the same position with the same bits.  Only members of an equal-time group may be permuted."""
    assert [len(p) for _, p in got] == [len(p) for _, p in ref]
    n_tied = 0
    for (ta, pa), (tb, pb) in zip(got, ref):
        assert ta == tb
        assert np.array_equal(pa[:, 3].view(np.uint32), pb[:, 3].view(np.uint32))
        t = pa[:, 3]
        uniq = np.ones(len(t), bool)
        same = t[1:] == t[:-1]
        uniq[1:] &= ~same
        uniq[:-1] &= ~same
        assert np.array_equal(pa[uniq].view(np.uint32), pb[uniq].view(np.uint32))
        n_tied += int((~uniq).sum())
    return n_tied


def _jitter_times(t_ms, seed=5):
    """Strictly distinct per-point times: with ties the reference's std::sort order is unspecified."""
    rng = np.random.default_rng(seed)
    return t_ms + rng.permutation(len(t_ms)) * 1e-3


@pytest.mark.parametrize("lidar_type", [wire.VELO, wire.OUSTER, wire.PANDAR, wire.ROBOSENSE])
def test_pcl2_against_reference_code(oracle, sweep, lidar_type):
    if oracle.ref_preprocess_lib() is None:
        pytest.skip("oracle/_ref/libref_preprocess.so not built")
    xyz, ring, t_ms = sweep
    n = len(xyz)
    stamp = 1234.75  # small enough that stamp * 1000 + t keeps the 1e-3 ms jitter apart in double
    raw = wire.pack_pcl2(lidar_type, xyz, ring, _jitter_times(t_ms), stamp)
    f = wire.pc2_fields(lidar_type)
    for cut, sc, pfn in [(1, 100, 1), (3, 100, 2), (5, 100, 1), (4, 3, 3)]:
        got = oracle.ingest_pcl2(raw, n, f, lidar_type, 24, pfn, 1.5, stamp, cut, sc)
        ref = oracle.ref_ingest_pcl2(raw, n, f, lidar_type, 24, pfn, 1.5, stamp, cut, sc)
        tied = assert_frames_equal_mod_ties(got, ref)  # float32 seconds (Velodyne) still collide for a few points
        assert tied < 0.01 * n
    # time synthesis path (no per-point time)
    if lidar_type in (wire.VELO, wire.ROBOSENSE):
        raw = wire.pack_pcl2(lidar_type, xyz, ring, t_ms, stamp, with_time=False)
        got = oracle.ingest_pcl2(raw, n, f, lidar_type, 24, 1, 1.5, stamp, 3, 100)
        ref = oracle.ref_ingest_pcl2(raw, n, f, lidar_type, 24, 1, 1.5, stamp, 3, 100)
        assert [len(p) for _, p in got] == [len(p) for _, p in ref]
        for (ta, pa), (tb, pb) in zip(got, ref):  # equal azimuths across rings tie: compare as sorted time arrays
            assert ta == tb and np.array_equal(np.sort(pa[:, 3]), np.sort(pb[:, 3]))


def test_ties_against_reference_code(oracle, sweep):
    """Ouster columns share one time stamp: tie order is unspecified in the reference, everything else must agree."""
    if oracle.ref_preprocess_lib() is None:
        pytest.skip("oracle/_ref/libref_preprocess.so not built")
    xyz, ring, t_ms = sweep
    raw = wire.pack_pcl2(wire.OUSTER, xyz, ring, t_ms, 99.5)
    f = wire.pc2_fields(wire.OUSTER)
    got = oracle.ingest_pcl2(raw, len(xyz), f, wire.OUSTER, 32, 1, 1.0, 99.5, 3, 100)
    ref = oracle.ref_ingest_pcl2(raw, len(xyz), f, wire.OUSTER, 32, 1, 1.0, 99.5, 3, 100)
    assert [len(p) for _, p in got] == [len(p) for _, p in ref]
    for (ta, pa), (tb, pb) in zip(got, ref):
        assert ta == tb and np.array_equal(pa[:, 3], pb[:, 3])
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    ga, ra = np.concatenate([p for _, p in got]), np.concatenate([p for _, p in ref])
    # the dropped time-earliest point may be a different member of the first tie group; all other points coincide
    assert len(ga) == len(ra)
    both = np.intersect1d(ga[:, :3].view([("", np.float32)] * 3), ra[:, :3].view([("", np.float32)] * 3))
    assert len(both) >= len(ga) - 1


def test_livox_against_reference_code(oracle):
    if oracle.ref_preprocess_lib() is None:
        pytest.skip("oracle/_ref/libref_preprocess.so not built")
    hall = synth.Hall()
    raw, n = wire.avia_sweep(hall, synth.rot_zyx(0, 0, 0.3), np.array([0.5, 0.5, 0.0]), n_points=6000)
    for cut, sc, pfn in [(1, 100, 1), (5, 100, 2), (5, 3, 2), (7, 100, 3)]:
        got = oracle.ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, cut, sc)
        ref = oracle.ref_ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, cut, sc)
        assert_frames_equal(got, ref)


# ----------------------------------------------------------------------------------------------------------------------
# The callbacks' other branch (initialization/cut_frame: false, src/laserMapping.cpp:337-342, :374-379): Preprocess::process -
# oust_handler / velodyne_handler / l515_handler / avia_handler with feature extraction disabled (src/preprocess.cpp:337-713) -
# selected by cut_frame_num = 0.  No sort, so no ties: every bit must agree with the reference's own code.
WHOLE_CASES = [(wire.OUSTER, 1, True), (wire.OUSTER, 3, True), (wire.VELO, 1, True), (wire.VELO, 2, True), (wire.VELO, 1, False), (wire.VELO, 3, False),
               (wire.L515, 1, True), (wire.L515, 4, True)]


def np_whole_pcl2(lidar_type, raw, n, n_scans, pfn, blind, with_time):
    """The handlers' rules as array operations (independent of the loop restatement), for messages that carry per-point time."""
    a = np.frombuffer(raw, wire.DTYPES[lidar_type], count=n)
    x, y, z = a["x"], a["y"], a["z"]
    d = ((x * x + y * y) + z * z).astype(np.float64)
    dec = np.arange(n) % pfn == 0
    if lidar_type == wire.L515:
        keep = dec & ~(d < blind * blind)  # (a NaN range is not below the blind radius: such points pass, as in the reference)
        t = np.zeros(n, np.float32)
    else:
        ok = ~((d < blind * blind) | np.isnan(x) | np.isnan(y) | np.isnan(z))
        if lidar_type == wire.OUSTER:
            keep = dec & ok & (a["ring"].astype(np.int64) < n_scans)
            t = (a["t"].astype(np.float64) / 1e6).astype(np.float32)
        else:
            keep = dec & ok  # velodyne_handler's feature-less branch has no ring test
            t = (a["time"].astype(np.float64) * 1000.0).astype(np.float32)
    return np.stack([x, y, z, t], 1)[keep]


@pytest.mark.parametrize("lidar_type,pfn,with_time", WHOLE_CASES)
def test_whole_message_handlers(oracle, sweep, lidar_type, pfn, with_time):
    xyz, ring, t_ms = sweep
    n, stamp = len(xyz), 4321.5
    raw = wire.pack_pcl2(lidar_type, xyz, ring, t_ms, stamp, with_time=with_time)
    f = wire.pc2_fields(lidar_type)
    got = oracle.ingest_pcl2(raw, n, f, lidar_type, 12, pfn, 1.5, stamp, 0, 100)
    assert len(got) == 1 and got[0][0] == stamp * 1000
    if with_time:
        want = np_whole_pcl2(lidar_type, raw, n, 12, pfn, 1.5, with_time)
        assert got[0][1].shape == want.shape and np.array_equal(got[0][1].view(np.uint32), want.view(np.uint32))
        if lidar_type == wire.OUSTER:
            assert len(want) < n * 0.6 / pfn + 1  # rings 12 .. 15 and the missing returns are gone
    else:  # time from the azimuth: every ring's first surviving point only seeds the recurrence
        assert np.all(got[0][1][:, 3] >= 0) and got[0][1][:, 3].max() < 2 * 360.0 / 3.61 + 1.0  # (up to two revolutions at 3.61 deg/ms)
    if oracle.ref_preprocess_lib() is not None:  # the reference's own Preprocess::process
        ref = oracle.ref_ingest_pcl2(raw, n, f, lidar_type, 12, pfn, 1.5, stamp, 0, 100)
        assert_frames_equal(got, ref)


def test_whole_message_livox_and_edge_cases(oracle):
    hall = synth.Hall()
    raw, n = wire.avia_sweep(hall, synth.rot_zyx(0, 0, 0.3), np.array([0.5, 0.5, 0.0]), n_points=6000)
    for pfn in (1, 2, 3):
        got = oracle.ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, 0, 100)
        cut1 = oracle.ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, 1, 100)
        assert len(got) == 1 and got[0][0] == 12500.0
        # the cutting form emits the same points but the time-earliest one, sorted by time; this one keeps the input order
        assert len(got[0][1]) == len(cut1[0][1]) + 1
        assert np.array_equal(np.sort(got[0][1][:, 3])[1:], cut1[0][1][:, 3])
        if oracle.ref_preprocess_lib() is not None:
            assert_frames_equal(got, oracle.ref_ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, 0, 100))
    # a lidar_type Preprocess::process does not know ("Error LiDAR Type", the cloud of the previous message is handed back): refused
    xyz = np.ones((4, 3), np.float32) * 3
    rawp = wire.pack_pcl2(wire.PANDAR, xyz, np.zeros(4, np.int32), np.arange(4.0), 5.0)
    with pytest.raises(RuntimeError):
        oracle.ingest_pcl2(rawp, 4, wire.pc2_fields(wire.PANDAR), wire.PANDAR, 16, 1, 0.5, 5.0, 0, 100)
    # an empty message / everything inside the blind zone: one empty cloud (the node skips it: laserMapping.cpp:909-914)
    rawo = wire.pack_pcl2(wire.OUSTER, xyz, np.zeros(4, np.int32), np.arange(4.0), 5.0)
    got = oracle.ingest_pcl2(rawo, 4, wire.pc2_fields(wire.OUSTER), wire.OUSTER, 16, 1, 10.0, 5.0, 0, 100)
    assert len(got) == 1 and len(got[0][1]) == 0


def test_ingest_golden_fixture(oracle):
    """Outputs of the reference's own Preprocess (generated by tests/golden/make_ingest_fixture.py) replayed through the
    restatement — runs everywhere, with or without oracle/_ref."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "ingest", "reference_frames.npz")
    z = np.load(path)
    cases = sorted({k.split("/")[0] for k in z.files})
    assert len(cases) >= 11
    for c in cases:
        meta = z[c + "/meta"]  # kind, lidar_type, n, n_scans, pfn, cut, scan_count
        kind, lidar_type, n, n_scans, pfn, cut, sc = [int(v) for v in meta]
        blind, stamp = [float(v) for v in z[c + "/params"]]
        raw = z[c + "/raw"].tobytes()
        if kind == 0:
            got = oracle.ingest_pcl2(raw, n, wire.pc2_fields(lidar_type), lidar_type, n_scans, pfn, blind, stamp, cut, sc)
        else:
            got = oracle.ingest_livox(raw, n, wire.livox_fields(), n_scans, pfn, blind, stamp, cut, sc)
        begin, counts, pts = z[c + "/begin_ms"], z[c + "/counts"], z[c + "/points"]
        assert [len(p) for _, p in got] == list(counts)
        assert np.array_equal([tb for tb, _ in got], begin)
        allp = np.concatenate([p for _, p in got]) if got else np.zeros((0, 4), np.float32)
        assert np.array_equal(allp.view(np.uint32), pts.view(np.uint32))
