"""The overlapped ingest (lii_ingest_pcl2_begin / lii_ingest_livox_begin / lii_ingest_end, ABI 9): driver messages queue ON THE
DEVICE the way the reference's callbacks queue them in lidar_buffer / time_buffer (src/laserMapping.cpp:326-379, drained by
sync_packages :432-480).  The frames must be the one-call forms' bits (which tests/test_gpu_ingest.py holds against the oracle and the
reference's own preprocess.cpp), whatever is under way beside them."""
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


def frames_of(reg, info):
    out = []
    for k, (tb, off, cnt) in enumerate(info):
        reg.frame_select(k)
        pts = reg.scan_download(0)
        assert len(pts) == cnt
        out.append((tb, off, pts.copy()))
    return out


def same(a, b):
    assert len(a) == len(b)
    for (ta, oa, pa), (tb, ob, pb) in zip(a, b):
        assert ta == tb and oa == ob
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))


def messages():
    hall = synth.Hall()
    out = []
    for k, (sensor, lt) in enumerate([("os1_128", wire.OUSTER), ("vlp16", wire.VELO), ("mid16k", wire.PANDAR), ("os1_128", wire.OUSTER),
                                      ("vlp16", wire.VELO)]):
        xyz, ring, t_ms = wire.raw_sweep(hall, sensor, synth.rot_zyx(0.01 * k, 0.02, 0.3 * k), np.array([0.5 * k, -1.0, 0.2]))
        stamp = 1000.0 + 0.1 * k
        raw = wire.pack_pcl2(lt, xyz, ring, t_ms, stamp)
        n_scans = synth.SENSORS[sensor][0] - 2
        cut = [3, 1, 5, 0, 4][k]
        if cut == 0 and lt not in (wire.OUSTER, wire.VELO):
            cut = 2
        out.append((raw, len(xyz), wire.pc2_fields(lt), lt, n_scans, 1 + k % 2, 1.0, stamp, cut, 100 + k))
    return out


def test_overlapped_messages_give_the_one_call_forms_bits(reg):
    msgs = messages()
    want = [frames_of(reg, reg.ingest_pcl2(*m)) for m in msgs]
    # two under way at any time, consumed in arrival order
    reg.ingest_pcl2_begin(*msgs[0])
    reg.ingest_pcl2_begin(*msgs[1])
    for k in range(len(msgs)):
        info = reg.ingest_end()
        if k + 2 < len(msgs):
            reg.ingest_pcl2_begin(*msgs[k + 2])  # ... while the frames of message k are read
        same(frames_of(reg, info), want[k])
    # one under way, and a one-call ingest in between does not disturb it (it replaces the frames being served, nothing else)
    reg.ingest_pcl2_begin(*msgs[2])
    same(frames_of(reg, reg.ingest_pcl2(*msgs[4])), want[4])
    same(frames_of(reg, reg.ingest_end()), want[2])


def test_overlapped_livox_and_empty_messages(reg):
    hall = synth.Hall()
    raw, n = wire.avia_sweep(hall, synth.rot_zyx(0, 0.05, -0.3), np.array([0.5, 0.5, 0.0]))
    f = wire.livox_fields()
    want = frames_of(reg, reg.ingest_livox(raw, n, f, 6, 2, 1.0, 1234.5, 5, 100))
    fo = wire.pc2_fields(wire.OUSTER)
    reg.ingest_livox_begin(raw, n, f, 6, 2, 1.0, 1234.5, 5, 100)
    reg.ingest_pcl2_begin(b"", 0, fo, wire.OUSTER, 16, 1, 0.5, 5.0, 3, 100)  # an empty message queues like any other
    same(frames_of(reg, reg.ingest_end()), want)
    assert reg.ingest_end() == []
    import lidar_imu_init_amd as lii
    with pytest.raises(lii.LIIError):
        reg.frame_select(0)  # (no frame, as after the one-call form)


def test_overlapped_call_order(reg):
    import lidar_imu_init_amd as lii
    msgs = messages()
    with pytest.raises(lii.LIIError) as e:
        reg.ingest_end()
    assert e.value.code == -5
    reg.ingest_pcl2_begin(*msgs[0])
    reg.ingest_pcl2_begin(*msgs[1])
    with pytest.raises(lii.LIIError) as e:
        reg.ingest_pcl2_begin(*msgs[2])  # a third message under way
    assert e.value.code == -5
    with pytest.raises(lii.LIIError) as e:
        reg.ingest_pcl2_begin(msgs[0][0], msgs[0][1], msgs[0][2], wire.AVIA, 16, 1, 0.5, 1.0, 3, 50)
    assert e.value.code == -1
    a = reg.ingest_end()
    b = reg.ingest_end()
    assert len(a) == 3 and len(b) == 1


def test_a_selected_frame_survives_the_next_messages(reg):
    """A frame selected and not read yet when its message leaves the front of the queue is copied out first - and the context it lay in
    is not overwritten under that copy."""
    msgs = messages()
    want0 = frames_of(reg, reg.ingest_pcl2(*msgs[0]))
    want3 = frames_of(reg, reg.ingest_pcl2(*msgs[3]))
    reg.ingest_pcl2_begin(*msgs[0])
    reg.ingest_pcl2_begin(*msgs[3])
    reg.ingest_end()
    reg.frame_select(2)  # ... of message 0: selected, not read
    info3 = reg.ingest_end()  # message 0 leaves the front
    reg.ingest_pcl2_begin(*msgs[1])  # ... and its context is taken at once
    reg.ingest_pcl2_begin(*msgs[2])
    pts = reg.scan_download(0)
    assert np.array_equal(pts.view(np.uint32), want0[2][2].view(np.uint32))
    same(frames_of(reg, info3), want3)
    reg.ingest_end()
    reg.ingest_end()


def test_overlapped_ingest_under_a_registration(oracle):
    """Messages decode on their own streams while the handle's stream registers: the registrations' results and the frames are the bits
    of the serial order."""
    import lidar_imu_init_amd as lii
    from conftest import make_state
    hall, map_pts = synth.bench_world(150_000, 0.15)
    r = lii.Registrar(max_scan_points=40_000, max_map_points=200_000, filter_size_map=0.15)
    try:
        r.map_build(map_pts)
        msgs, truth = [], []
        for k in range(4):
            R, p = synth.rot_zyx(0.0, 0.01, 0.3 + 0.02 * k), np.array([1.0 + 0.05 * k, 2.0, 0.3])
            xyz, ring, t_ms = wire.raw_sweep(hall, "vlp16", R, p, nan_fraction=0.0)
            msgs.append((wire.pack_pcl2(wire.VELO, xyz, ring, t_ms, 10.0 + 0.1 * k), len(xyz), wire.pc2_fields(wire.VELO), wire.VELO,
                         16, 1, 0.5, 10.0 + 0.1 * k, 2, 100))
            truth.append(make_state(oracle, R, p))

        def run(overlapped):
            out = []
            if overlapped:
                r.ingest_pcl2_begin(*msgs[0])
                r.ingest_pcl2_begin(*msgs[1])
            for k in range(len(msgs)):
                if overlapped:
                    info = r.ingest_end()
                    if k + 2 < len(msgs):
                        r.ingest_pcl2_begin(*msgs[k + 2])
                else:
                    info = r.ingest_pcl2(*msgs[k])
                for f in range(len(info)):
                    r.frame_select(f)
                    s0 = lii.State(oracle.state_boxplus(truth[k], np.r_[0.002, -0.002, 0.003, 0.02, -0.02, 0.01, np.zeros(18)]))
                    s = s0.copy()
                    rep = r.scan_register(s, s0, leaf=0.1, max_iterations=5, imu_en=False, scan_sorted=True)
                    out.append((info[f], np.array(s.pod).copy(), rep["iterations"], rep["effect_num"]))
            return out

        a = run(False)
        b = run(True)
        assert len(a) == len(b) == 8
        for (ia, sa, ita, ea), (ib, sb, itb, eb) in zip(a, b):
            assert ia == ib and ita == itb and ea == eb and ea > 2000
            assert np.array_equal(sa, sb)
    finally:
        r.close()


def test_the_next_message_is_begun_from_inside_the_registration(oracle):
    """lii_scan_job::while_waiting (ABI 9): the hook runs once inside lii_scan_register, when the job's launches are out - the place to put
    the next driver message under way.  Frames and registered states are the bits of the serial order."""
    import lidar_imu_init_amd as lii
    from conftest import make_state
    hall, map_pts = synth.bench_world(150_000, 0.15)
    r = lii.Registrar(max_scan_points=40_000, max_map_points=200_000, filter_size_map=0.15)
    try:
        r.map_build(map_pts)
        msgs, truth = [], []
        for k in range(5):
            R, p = synth.rot_zyx(0.0, 0.01, 0.3 + 0.02 * k), np.array([1.0 + 0.05 * k, 2.0, 0.3])
            xyz, ring, t_ms = wire.raw_sweep(hall, "vlp16", R, p, nan_fraction=0.0)
            msgs.append((wire.pack_pcl2(wire.VELO, xyz, ring, t_ms, 10.0 + 0.1 * k), len(xyz), wire.pc2_fields(wire.VELO), wire.VELO,
                         16, 1, 0.5, 10.0 + 0.1 * k, 2, 100))
            truth.append(make_state(oracle, R, p))
        calls, refused = [], []

        def run(hooked):
            out = []
            if hooked:
                r.ingest_pcl2_begin(*msgs[0])
                r.ingest_pcl2_begin(*msgs[1])
            for k in range(len(msgs)):
                info = r.ingest_end() if hooked else r.ingest_pcl2(*msgs[k])
                for f in range(len(info)):
                    r.frame_select(f)
                    s0 = lii.State(oracle.state_boxplus(truth[k], np.r_[0.002, -0.002, 0.003, 0.02, -0.02, 0.01, np.zeros(18)]))
                    s = s0.copy()
                    hook = None
                    if hooked and f == len(info) - 1 and k + 2 < len(msgs):
                        def hook(k=k):
                            calls.append(k)
                            if k == 0:  # what would retire or replace the frames the registration under way reads is refused inside the hook
                                for bad in (r.ingest_end, lambda: r.frame_select(0), lambda: r.ingest_pcl2(*msgs[0])):
                                    try:
                                        bad()
                                        refused.append(False)
                                    except lii.LIIError as e:
                                        refused.append(e.code == -5)
                            r.ingest_pcl2_begin(*msgs[k + 2])
                    rep = r.scan_register(s, s0, leaf=0.1, max_iterations=5, imu_en=False, scan_sorted=True, map_update=True, while_waiting=hook)
                    out.append((info[f], np.array(s.pod).copy(), rep["iterations"], rep["effect_num"]))
            return out

        a = run(False)
        n_a = r.map_size()
        r.map_build(map_pts)
        b = run(True)
        assert calls == [0, 1, 2]  # once per registration that carried a hook
        assert refused == [True, True, True]
        assert len(a) == len(b) == 10 and r.map_size() == n_a
        for (ia, sa, ita, ea), (ib, sb, itb, eb) in zip(a, b):
            assert ia == ib and ita == itb and ea == eb and ea > 2000
            assert np.array_equal(sa, sb)
    finally:
        r.close()


def test_the_time_sort_is_left_out_for_streams_in_time_order(oracle, monkeypatch):
    """The launch plan of the ingest's time sort: after a message whose kept points arrived in ascending time order the next one is enqueued
    without the sort (the device checks the order; a message that fails the check is cut again behind the sort).  Frames are the oracle's
    bits whatever the sequence of ordered and unordered messages, through the one-call and the overlapped forms."""
    import lidar_imu_init_amd as lii
    hall = synth.Hall()
    xyz, ring, t_ms = wire.raw_sweep(hall, "os1_128", synth.rot_zyx(0.01, 0.02, 0.4), np.array([0.5, -1.0, 0.2]))
    order = np.argsort(t_ms, kind="stable")
    rng = np.random.default_rng(3)
    shuffled = rng.permutation(len(xyz))
    f = wire.pc2_fields(wire.OUSTER)
    n_scans = synth.SENSORS["os1_128"][0] - 2

    def msg(perm, k, cut):
        stamp = 500.0 + 0.1 * k
        raw = wire.pack_pcl2(wire.OUSTER, xyz[perm], ring[perm], t_ms[perm], stamp)
        return (raw, len(xyz), f, wire.OUSTER, n_scans, 1, 1.0, stamp, cut, 100 + k)

    seq = [msg(order, 0, 3), msg(order, 1, 3), msg(order, 2, 1), msg(shuffled, 3, 3), msg(shuffled, 4, 2), msg(order, 5, 3), msg(order, 6, 4),
           msg(shuffled, 7, 3), msg(order, 8, 3)]
    want = [oracle.ingest_pcl2(*m) for m in seq]

    def check(got, orc):
        assert len(got) == len(orc)
        for (tb_g, _, pg), (tb_o, po) in zip(got, orc):
            assert tb_g == tb_o / 1000.0 and pg.shape == po.shape
            assert np.array_equal(pg.view(np.uint32), po.view(np.uint32))

    r = lii.Registrar(max_scan_points=140_000, max_map_points=1000, filter_size_map=0.2)
    try:
        for m, w in zip(seq, want):  # one-call form
            check(frames_of(r, r.ingest_pcl2(*m)), w)
        r.ingest_pcl2_begin(*seq[0])  # overlapped: the prediction made for a message is one or two messages old
        r.ingest_pcl2_begin(*seq[1])
        for k in range(len(seq)):
            info = r.ingest_end()
            if k + 2 < len(seq):
                r.ingest_pcl2_begin(*seq[k + 2])
            check(frames_of(r, info), want[k])
    finally:
        r.close()
    monkeypatch.setenv("LII_INGEST_SORT", "always")  # (read when a handle's ingest state is created)
    r2 = lii.Registrar(max_scan_points=140_000, max_map_points=1000, filter_size_map=0.2)
    try:
        for m, w in zip(seq[:4], want[:4]):
            check(frames_of(r2, r2.ingest_pcl2(*m)), w)
    finally:
        r2.close()


def test_a_handle_is_destroyed_with_messages_under_way():
    """lii_destroy waits for the ingest's streams and frees the ring: no crash, no hang, and the next handle starts clean."""
    import lidar_imu_init_amd as lii
    msgs = messages()
    r = lii.Registrar(max_scan_points=140_000, max_map_points=1000, filter_size_map=0.2)
    r.ingest_pcl2_begin(*msgs[0])
    r.ingest_pcl2_begin(*msgs[1])
    r.close()
    r = lii.Registrar(max_scan_points=140_000, max_map_points=1000, filter_size_map=0.2)
    try:
        want = frames_of(r, r.ingest_pcl2(*msgs[0]))
        r.ingest_pcl2_begin(*msgs[0])
        same(frames_of(r, r.ingest_end()), want)
    finally:
        r.close()
