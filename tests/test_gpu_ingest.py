"""Device ingest (lii_ingest_pcl2 / lii_ingest_livox / lii_frame_select) against the oracle restatement of
Preprocess::process_cut_frame_* (reference src/preprocess.cpp:50-335): bit-exact frames, times and boundaries."""
import numpy as np
import pytest

from harness import synth, wire

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    import lidar_imu_init_amd as lii
    r = lii.Registrar(max_scan_points=140_000, max_map_points=1000, filter_size_map=0.2)
    yield r
    r.close()


def gpu_frames(reg, info):
    out = []
    for k, (tb, off, cnt) in enumerate(info):
        reg.frame_select(k)
        pts = reg.scan_download(0)
        assert len(pts) == cnt
        out.append((tb, pts))
    return out


def assert_same(gpu, orc, exact_time=True):
    assert len(gpu) == len(orc)
    for (tb_g, pg), (tb_o, po) in zip(gpu, orc):
        assert tb_g == tb_o / 1000.0
        assert pg.shape == po.shape
        if exact_time:
            assert np.array_equal(pg.view(np.uint32), po.view(np.uint32))
        else:
            assert np.array_equal(pg[:, :3].view(np.uint32), po[:, :3].view(np.uint32))
            assert np.allclose(pg[:, 3], po[:, 3], rtol=3e-7, atol=1e-5)


@pytest.mark.parametrize("sensor,lidar_type", [("vlp16", wire.VELO), ("os1_128", wire.OUSTER), ("mid16k", wire.PANDAR),
                                               ("mid16k", wire.ROBOSENSE)])
def test_pcl2_frames_bit_exact(reg, oracle, sensor, lidar_type):
    hall = synth.Hall()
    xyz, ring, t_ms = wire.raw_sweep(hall, sensor, synth.rot_zyx(0.02, 0.01, 0.7), np.array([1.0, -1.0, 0.2]))
    n = len(xyz)
    stamp = 1_650_000_321.5
    raw = wire.pack_pcl2(lidar_type, xyz, ring, t_ms, stamp)
    f = wire.pc2_fields(lidar_type)
    n_scans = synth.SENSORS[sensor][0] - 2  # the two highest rings are filtered out
    for cut, sc, pfn in [(1, 100, 1), (3, 100, 1), (5, 100, 2), (4, 3, 3)]:
        info = reg.ingest_pcl2(raw, n, f, lidar_type, n_scans, pfn, 1.0, stamp, cut, sc)
        orc = oracle.ingest_pcl2(raw, n, f, lidar_type, n_scans, pfn, 1.0, stamp, cut, sc)
        assert len(info) == (1 if sc < 20 else cut)
        assert_same(gpu_frames(reg, info), orc)
        offs = [o for _, o, _ in info]
        assert offs == list(np.cumsum([0] + [c for _, _, c in info])[:-1])


def test_pcl2_time_synthesis(reg, oracle):
    """Clouds without per-point time: time from the azimuth, per-ring recurrence (src/preprocess.cpp:163-185)."""
    hall = synth.Hall()
    xyz, ring, t_ms = wire.raw_sweep(hall, "vlp16", synth.rot_zyx(0, 0, 0.2), np.array([0.0, 0.5, 0.1]))
    n = len(xyz)
    for lidar_type in (wire.VELO, wire.ROBOSENSE):
        raw = wire.pack_pcl2(lidar_type, xyz, ring, t_ms, 77.0, with_time=False)
        f = wire.pc2_fields(lidar_type)
        info = reg.ingest_pcl2(raw, n, f, lidar_type, 16, 1, 0.8, 77.0, 3, 100)
        orc = oracle.ingest_pcl2(raw, n, f, lidar_type, 16, 1, 0.8, 77.0, 3, 100)
        # atan2 of the device library and of glibc may differ in the last place: points are compared after sorting each
        # frame by (time, x) with a tolerance on the time only
        gpu = gpu_frames(reg, info)
        assert [len(p) for _, p in gpu] == [len(p) for _, p in orc]
        for (tg, pg), (to, po) in zip(gpu, orc):
            assert abs(tg - to / 1000.0) < 1e-9
            assert np.allclose(np.sort(pg[:, 3]), np.sort(po[:, 3]), rtol=1e-6, atol=1e-4)
        tot_g = np.concatenate([p for _, p in gpu])
        tot_o = np.concatenate([p for _, p in orc])
        key = lambda a: np.lexsort((a[:, 2], a[:, 1], a[:, 0]))
        assert np.array_equal(tot_g[key(tot_g)][:, :3], tot_o[key(tot_o)][:, :3])
        assert 0 < tot_o[:, 3].max() < 210.0


def test_livox_frames_bit_exact(reg, oracle):
    hall = synth.Hall()
    raw, n = wire.avia_sweep(hall, synth.rot_zyx(0, 0.05, -0.3), np.array([0.5, 0.5, 0.0]))
    f = wire.livox_fields()
    for cut, sc, pfn in [(1, 100, 1), (5, 100, 2), (5, 2, 2), (7, 100, 3)]:
        info = reg.ingest_livox(raw, n, f, 6, pfn, 1.0, 1234.5, cut, sc)
        orc = oracle.ingest_livox(raw, n, f, 6, pfn, 1.0, 1234.5, cut, sc)
        assert len(info) == (1 if sc < 5 else cut)
        assert_same(gpu_frames(reg, info), orc)


def test_ingest_edge_cases(reg, oracle):
    import lidar_imu_init_amd as lii
    f = wire.pc2_fields(wire.OUSTER)
    assert reg.ingest_pcl2(b"", 0, f, wire.OUSTER, 16, 1, 0.5, 5.0, 3, 100) == []
    with pytest.raises(lii.LIIError):
        reg.frame_select(0)
    k = np.arange(10)
    yaw = np.deg2rad(170.0 - 36.0 * k)
    xyz = np.stack([4 * np.cos(yaw), 4 * np.sin(yaw), np.zeros(10)], 1).astype(np.float32)
    # everything inside the blind zone; a cloud too small to be cut
    raw = wire.pack_pcl2(wire.OUSTER, xyz, np.zeros(10, np.int32), k * 1.0, 5.0)
    assert reg.ingest_pcl2(raw, 10, f, wire.OUSTER, 16, 1, 10.0, 5.0, 1, 100) == []
    assert reg.ingest_pcl2(raw[:2 * f[0]], 2, f, wire.OUSTER, 16, 1, 0.5, 5.0, 3, 100) == []
    # the hand-worked 7-point case of tests/test_oracle_ingest.py
    t_ms = np.array([30, 0, 10, 20, 60, 40, 50], np.float64)
    xyz7 = np.stack([np.full(7, 5.0), np.arange(7, dtype=np.float64), np.zeros(7)], 1).astype(np.float32)
    raw = wire.pack_pcl2(wire.OUSTER, xyz7, np.zeros(7, np.int32), t_ms, 100.0)
    info = reg.ingest_pcl2(raw, 7, f, wire.OUSTER, 16, 1, 0.5, 100.0, 3, 50)
    assert [(tb, c) for tb, _, c in info] == [(100.0, 1), (100.01, 2), (100.03, 3)]
    assert_same(gpu_frames(reg, info), oracle.ingest_pcl2(raw, 7, f, wire.OUSTER, 16, 1, 0.5, 100.0, 3, 50))
    # bad arguments
    with pytest.raises(lii.LIIError) as e:
        reg.ingest_pcl2(raw, 7, f, wire.AVIA, 16, 1, 0.5, 100.0, 3, 50)
    assert e.value.code == -1
    # a frame larger than max_scan_points cannot become the current scan
    big = np.tile(xyz7, (30000, 1))
    rawb = wire.pack_pcl2(wire.OUSTER, big, np.zeros(len(big), np.int32), np.linspace(0, 100, len(big)), 1.0)
    info = reg.ingest_pcl2(rawb, len(big), f, wire.OUSTER, 16, 1, 0.5, 1.0, 1, 100)
    assert info[0][2] == len(big) - 1
    with pytest.raises(lii.LIIError) as e:
        reg.frame_select(0)
    assert e.value.code == -4


def test_ingest_feeds_registration(reg, oracle):
    """A driver message goes wire -> frames -> undistort -> voxel grid -> update without the points leaving the device."""
    import lidar_imu_init_amd as lii
    from conftest import make_state
    hall, map_pts = synth.bench_world(150_000, 0.15)
    r2 = lii.Registrar(max_scan_points=40_000, max_map_points=200_000, filter_size_map=0.15)
    r2.map_build(map_pts)
    R, p = synth.rot_zyx(0.0, 0.01, 0.3), np.array([1.0, 2.0, 0.3])
    xyz, ring, t_ms = wire.raw_sweep(hall, "vlp16", R, p, nan_fraction=0.0)
    raw = wire.pack_pcl2(wire.VELO, xyz, ring, t_ms, 10.0)
    info = r2.ingest_pcl2(raw, len(xyz), wire.pc2_fields(wire.VELO), wire.VELO, 16, 1, 0.5, 10.0, 2, 100)
    assert len(info) == 2
    st_true = make_state(oracle, R, p)
    for k in range(2):
        r2.frame_select(k)
        s0 = lii.State(oracle.state_boxplus(st_true, np.r_[0.002, -0.002, 0.003, 0.02, -0.02, 0.01, np.zeros(18)]))
        s = s0.copy()
        rep = r2.scan_register(s, s0, leaf=0.1, max_iterations=5, imu_en=False)
        assert rep["effect_num"] > 2000
        assert np.linalg.norm(s.pos_end - p) < 0.01
    r2.close()


@pytest.mark.parametrize("sensor,lidar_type,pfn,with_time", [("os1_128", wire.OUSTER, 1, True), ("os1_128", wire.OUSTER, 3, True), ("vlp16", wire.VELO, 1, True),
                                                             ("vlp16", wire.VELO, 2, True), ("vlp16", wire.VELO, 1, False), ("mid16k", wire.L515, 2, True)])
def test_whole_message_bit_exact(reg, oracle, sensor, lidar_type, pfn, with_time):
    """cut_frame_num = 0: the callbacks' branch for initialization/cut_frame: false - Preprocess::process (oust_handler / velodyne_handler /
    l515_handler, feature extraction disabled, src/preprocess.cpp:337-713) on the device: the handler's own filters, no time sort, no
    cut, ONE frame in input order stamped with the message's time.  Against the oracle's restatement - which tests/test_oracle_ingest.py
    holds bit for bit to the reference's unmodified Preprocess::process - and, where it travelled with the repository, against that
    library itself."""
    hall = synth.Hall()
    xyz, ring, t_ms = wire.raw_sweep(hall, sensor, synth.rot_zyx(0.02, 0.01, 0.7), np.array([1.0, -1.0, 0.2]))
    n, stamp = len(xyz), 1_650_000_321.5
    raw = wire.pack_pcl2(lidar_type, xyz, ring, t_ms, stamp, with_time=with_time)
    f = wire.pc2_fields(lidar_type)
    n_scans = synth.SENSORS[sensor][0] - 2
    info = reg.ingest_pcl2(raw, n, f, lidar_type, n_scans, pfn, 1.0, stamp, 0, 100)
    orc = oracle.ingest_pcl2(raw, n, f, lidar_type, n_scans, pfn, 1.0, stamp, 0, 100)
    assert len(info) == 1 and info[0][1] == 0 and info[0][2] == len(orc[0][1])
    got = gpu_frames(reg, info)
    assert_same(got, orc, exact_time=with_time)  # (synthesised times: atan2 of the device library vs glibc, <= 1 ulp of float32)
    assert reg.frame_tail_ms[0] == float(got[0][1][-1, 3])  # points.back().curvature: what sync_packages ends the scan with
    if with_time and oracle.ref_preprocess_lib() is not None:
        ref = oracle.ref_ingest_pcl2(raw, n, f, lidar_type, n_scans, pfn, 1.0, stamp, 0, 100)
        assert_same(got, ref)
    if lidar_type == wire.L515:
        # with `cut_frame: true` the reference's callback still sends an L515 message through Preprocess::process (src/laserMapping.cpp:363-377:
        # only Velodyne / Ouster / Pandar / RoboSense are cut): the same single frame, not "Wrong LiDAR Type" 
        info3 = reg.ingest_pcl2(raw, n, f, lidar_type, n_scans, pfn, 1.0, stamp, 3, 100)
        assert len(info3) == 1 and info3[0][2] == info[0][2]
        assert_same(gpu_frames(reg, info3), orc, exact_time=with_time)


def test_whole_message_livox_and_errors(reg, oracle):
    import lidar_imu_init_amd as lii
    hall = synth.Hall()
    raw, n = wire.avia_sweep(hall, synth.rot_zyx(0, 0, 0.3), np.array([0.5, 0.5, 0.0]), n_points=24000)
    for pfn in (1, 2, 3):
        info = reg.ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, 0, 100)
        orc = oracle.ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, 0, 100)
        assert len(info) == 1
        assert_same(gpu_frames(reg, info), orc)
    # Preprocess::process does not know Pandar / RoboSense clouds ("Error LiDAR Type"): refused, not decoded as something else
    xyz = np.ones((8, 3), np.float32) * 3
    rawp = wire.pack_pcl2(wire.PANDAR, xyz, np.zeros(8, np.int32), np.arange(8.0), 5.0)
    with pytest.raises(lii.LIIError):
        reg.ingest_pcl2(rawp, 8, wire.pc2_fields(wire.PANDAR), wire.PANDAR, 16, 1, 0.5, 5.0, 0, 100)
    # everything inside the blind zone: no frame (the node skips an empty cloud)
    rawo = wire.pack_pcl2(wire.OUSTER, xyz, np.zeros(8, np.int32), np.arange(8.0), 5.0)
    assert reg.ingest_pcl2(rawo, 8, wire.pc2_fields(wire.OUSTER), wire.OUSTER, 16, 1, 10.0, 5.0, 0, 100) == []
    # and the frame registers like any other scan handed over unsorted
    info = reg.ingest_livox(raw, n, wire.livox_fields(), 6, 2, 1.0, 12.5, 0, 100)
    reg.frame_select(0)
    assert len(reg.scan_download(0)) == info[0][2]
