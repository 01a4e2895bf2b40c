"""The ROS-free C++ host (harness/li_init_replay.cpp: the main() loop of src/laserMapping.cpp:891-1238 over the C-ABI only).

CPU: its two forward-propagation routines against the UNMODIFIED reference header compiled as oracle/_ref/libref_imu.so
     (ImuProcess::Forward_propagation_without_imu, src/IMU_Processing.hpp:204-244; propagation_and_undist, :271-382).
GPU: a synthetic Ouster-layout PointCloud2 + IMU stream pushed through the host message by message - callbacks, device ingest and
     sub-frame cut, sync_packages, LO with the constant-velocity model, movement detection, accumulation, excitation appraisal,
     LI_Initialization, the switch to LIO (state re-expressed in the IMU frame, IMU stamps compensated, sub-frame count changed),
     IMU forward propagation + back-propagation de-skew, the refinement result - against the SAME sequence of library calls made
     from Python with the numpy propagation of harness/lo_harness.py / lio_harness.py, and the result file it writes against the
     reference's format (harness/result_file.py: parser of result/Initialization_result.txt)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "harness", "libliinit_replay.so")


def _drv():
    if not os.path.exists(LIB):
        pytest.skip("harness/libliinit_replay.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    d = C.CDLL(LIB)
    D = C.POINTER(C.c_double)
    d.lii_replay_cv_propagate.restype = None
    d.lii_replay_cv_propagate.argtypes = [C.c_void_p, C.c_double, D, D]
    d.lii_replay_imu_propagate.restype = None
    d.lii_replay_imu_propagate.argtypes = [C.c_void_p, D, C.c_int32, D, D, D, C.c_double, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_int32)]
    return d


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _random_state(rng, lio):
    import lidar_imu_init_amd as lii
    from harness import synth
    st = lii.State()
    st.rot_end[:] = synth.rot_zyx(*rng.normal(0, 0.5, 3))
    st.pos_end[:] = rng.normal(0, 2.0, 3)
    st.vel_end[:] = rng.normal(0, 0.5, 3)
    st.bias_g[:] = rng.normal(0, 0.4 if not lio else 0.003, 3)
    if lio:
        st.bias_a[:] = rng.normal(0, 0.01, 3)
        st.offset_R_L_I[:] = synth.rot_zyx(*rng.normal(0, 0.3, 3))
        st.offset_T_L_I[:] = rng.normal(0, 0.1, 3)
        st.gravity[:] = [0.1, -0.2, -9.8]
    A = rng.normal(0, 1e-2, (24, 24))
    st.cov[:] = A @ A.T + np.eye(24) * 1e-4
    return st


def test_cv_propagation_equals_the_reference_header():
    from oracle import oracle as O
    if O.ref_imu_lib() is None:
        pytest.skip("oracle/_ref/libref_imu.so not built (needs /root/reference)")
    d = _drv()
    rng = np.random.default_rng(5)
    for dt in (0.05, 0.1, 0.013):
        st = _random_state(rng, lio=False)
        pts = np.c_[rng.uniform(-5, 5, (4, 3)), [0.0, 10.0, 20.0, 30.0]].astype(np.float32)
        cg, ca = np.full(3, 50.0), np.full(3, 2.0)
        want, _ = O.ref_imu_process_cv(100.0 + dt, 100.0, False, cg, ca, st.pod, pts)
        got = st.copy()
        d.lii_replay_cv_propagate(got.pod.ctypes.data_as(C.c_void_p), dt, _dp(cg), _dp(ca))
        # the reference computes dt as a difference of absolute stamps: allow its rounding (1e-14 relative on dt)
        assert np.allclose(got.pod[:36], want[:36], rtol=0, atol=1e-12)
        assert np.abs(got.cov - want[36:].reshape(24, 24)).max() <= 1e-12 * np.abs(want[36:]).max()


def test_imu_propagation_equals_the_reference_header():
    from oracle import oracle as O
    import lidar_imu_init_amd as lii
    if O.ref_imu_lib() is None:
        pytest.skip("oracle/_ref/libref_imu.so not built (needs /root/reference)")
    d = _drv()
    rng = np.random.default_rng(6)
    for n_imu, end_after in ((20, True), (11, False), (2, True)):
        st = _random_state(rng, lio=True)
        t0 = 50.0
        last_end = t0 + 0.0003          # the previous scan ended between two IMU samples
        last_imu = np.r_[t0 - 0.002, rng.normal(0, 0.3, 3), rng.normal(0, 0.5, 3) + [0, 0, 9.8]]
        t = t0 + 0.003 + 0.005 * np.arange(n_imu)
        imu = np.c_[t, rng.normal(0, 0.3, (n_imu, 3)), rng.normal(0, 0.5, (n_imu, 3)) + [0, 0, 9.8]]
        beg = last_end
        end = t[-1] + (0.002 if end_after else -0.001)
        pts = np.c_[rng.uniform(-5, 5, (3, 3)), [0.0, 1e3 * (end - beg) / 2, 1e3 * (end - beg)]].astype(np.float32)
        acc_s_last, angvel_last = rng.normal(0, 0.2, 3), rng.normal(0, 0.2, 3)
        cov_gyr, cov_acc = np.full(3, 0.1), np.full(3, 0.1)
        ref = O.ref_imu_process_lio(imu, last_imu, last_end, acc_s_last, angvel_last, cov_gyr, cov_acc, 9.805, beg, st.pod, pts)
        got = st.copy()
        carry = np.r_[acc_s_last, angvel_last, last_end]
        cov6 = np.r_[cov_gyr, cov_acc, np.full(3, 1e-4), np.full(3, 1e-4), np.full(3, 1e-5), np.full(3, 1e-4)]  # ImuProcess() defaults
        poses = lii.pose6d_array(n_imu + 2)
        K = C.c_int32(0)
        d.lii_replay_imu_propagate(got.pod.ctypes.data_as(C.c_void_p), _dp(np.ascontiguousarray(imu)), n_imu, _dp(last_imu), _dp(carry), _dp(cov6),
                                   9.805, beg, beg + float(pts[-1, 3]) / 1000.0, poses.ctypes.data_as(C.c_void_p), C.byref(K))
        assert K.value == len(ref["poses"])
        assert np.allclose(poses[:K.value], ref["poses"], rtol=0, atol=1e-12)
        assert np.allclose(got.pod[:36], ref["state"][:36], rtol=0, atol=1e-12)
        assert np.abs(got.cov - ref["state"][36:].reshape(24, 24)).max() <= 1e-12 * np.abs(ref["state"][36:]).max()
        assert np.allclose(carry[:3], ref["acc_s_last"], atol=1e-12) and np.allclose(carry[3:6], ref["angvel_last"], atol=1e-12)
        assert abs(carry[6] - ref["last_lidar_end_time"]) < 1e-12


# ----------------------------------------------------------------------------------------------------------------------------
YAML = """common:
  lid_topic: "/ouster/points"
  imu_topic: "/imu/data"
preprocess:
  lidar_type: 3
  scan_line: 32
  blind: 0.5
  feature_extract_en: false
initialization:
  cut_frame: true
  cut_frame_num: 2
  orig_odom_freq: 10
  mean_acc_norm: 9.81
  online_refine_time: 1.5
  data_accum_length: 80
  Rot_LI_cov: [0.00005, 0.00005, 0.00005]
  Trans_LI_cov: [0.0001, 0.0001, 0.0001]
mapping:
  filter_size_surf: 0.1
  filter_size_map: 0.15
  gyr_cov: 50
  acc_cov: 2
  b_acc_cov: 0.0001
  b_gyr_cov: 0.0001
"""
LAUNCH = """<launch>
  <rosparam command="load" file="$(find lidar_imu_init)/config/replay_test.yaml" />
  <param name="point_filter_num" type="int" value="1"/>
  <param name="max_iteration" type="int" value="5"/>
  <param name="cube_side_length" type="double" value="2000"/>
</launch>
"""


class ReplayConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_scan_points", C.c_int32), ("max_map_points", C.c_int32),
                ("launch_file", C.c_char_p), ("config_dir", C.c_char_p), ("result_path", C.c_char_p), ("stop_after_init", C.c_int32),
                ("reserved0", C.c_int32)]


def _status_type():
    from lidar_imu_init_amd.api import lii_calib_result

    class ReplayStatus(C.Structure):
        _fields_ = [("struct_size", C.c_uint32), ("imu_en", C.c_int32), ("data_accum_start", C.c_int32), ("data_accum_finished", C.c_int32),
                    ("refine_done", C.c_int32), ("scans_processed", C.c_int32), ("frames_pending", C.c_int32), ("imu_pending", C.c_int32),
                    ("cut_frame_num", C.c_int32), ("move_start_time", C.c_double), ("time_lag_imu_wrt_lidar", C.c_double),
                    ("timediff_imu_wrt_lidar", C.c_double), ("mean_acc_norm", C.c_double), ("lidar_end_time", C.c_double),
                    ("state", C.c_double * 612), ("init", lii_calib_result), ("init_time_lag_1", C.c_double), ("init_total_time_lag", C.c_double)]
    return ReplayStatus


def _bind(d):
    d.lii_replay_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    d.lii_replay_imu.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    d.lii_replay_pcl2.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int32, C.c_void_p]
    d.lii_replay_livox.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int32, C.c_void_p]
    d.lii_replay_spin.argtypes = [C.c_void_p]
    d.lii_replay_last_error.restype = C.c_char_p
    d.lii_replay_last_error.argtypes = [C.c_void_p]
    d.lii_replay_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    d.lii_replay_get_status.argtypes = [C.c_void_p, C.c_void_p]
    d.lii_replay_destroy.argtypes = [C.c_void_p]


def _cxx_host(d, launch, result_path, msgs, imu, fields, livox, msg_period, max_scan, max_map):
    """Plays the stream into harness/li_init_replay.cpp the way a bag plays it: the IMU samples up to the end of a message's sweep, then
    the message, then one turn of the loop.  Returns (log rows, status)."""
    _bind(d)
    t_imu, gyro, accel = imu
    cfg = ReplayConfig(C.sizeof(ReplayConfig), 0, max_scan, max_map, launch.encode(), None, result_path.encode() if result_path else None, 0, 0)
    rp = C.c_void_p()
    assert d.lii_replay_create(C.byref(cfg), C.byref(rp)) == 0
    k_imu = 0
    for stamp, raw, n in msgs:
        while k_imu < len(t_imu) and t_imu[k_imu] <= stamp + msg_period:
            g, a = np.ascontiguousarray(gyro[k_imu]), np.ascontiguousarray(accel[k_imu])
            assert d.lii_replay_imu(rp, float(t_imu[k_imu]), _dp(g), _dp(a)) == 0
            k_imu += 1
        push = d.lii_replay_livox if livox else d.lii_replay_pcl2
        assert push(rp, stamp, raw.ctypes.data_as(C.c_void_p), n, C.byref(fields)) == 0
        rc = d.lii_replay_spin(rp)
        assert rc >= 0, d.lii_replay_last_error(rp)
    n_rows = C.c_int32(0)
    assert d.lii_replay_log(rp, None, 0, C.byref(n_rows)) == 0
    log = np.zeros((n_rows.value, 40))
    assert d.lii_replay_log(rp, log.ctypes.data_as(C.c_void_p), n_rows.value, C.byref(n_rows)) == 0
    ST = _status_type()
    status = ST()
    status.struct_size = C.sizeof(ST)
    assert d.lii_replay_get_status(rp, C.byref(status)) == 0
    d.lii_replay_destroy(rp)
    return log, status


def _python_host(msgs, imu, ingest, msg_period, p):
    """The same sequence of library calls from Python with the numpy propagation of harness/lo_harness.py / lio_harness.py.
    p: the parameters the launch file gives the C++ host.  Returns (rows, LI-Init result or None)."""
    import lidar_imu_init_amd as lii
    from lidar_imu_init_amd import calib_state_array
    from lidar_imu_init_amd.api import data_sufficiency
    from harness.lo_harness import cv_propagate
    from harness.lio_harness import LioOdometry
    t_imu, gyro, accel = imu
    reg = lii.Registrar(max_scan_points=p["max_scan"], max_map_points=p["max_map"], filter_size_map=p["fs_map"])
    st = lii.State()
    rows = []
    S = dict(imu_buf=[], imu_all=[], lidar_states=[], omg=[], time_lag=0.0, imu_en=False, accum_start=False, accum_done=False, move_start=0.0,
             first_frame=True, t_last_beg=0.0, map_built=False, cut=p["cut"], lio=None, hand_over_frame=False, init_res=None, last_ts_imu=-1.0)

    def process(frame, t_beg, t_end, meas):
        select, kw = True, {}
        if S["imu_en"]:
            if not meas:
                return
            if S["hand_over_frame"]:  # the first LIO frame only re-arms the IMU processor; the previous scan is registered once more
                S["lio"].last_imu, S["lio"].last_lidar_end_time = meas[-1], 0.0
                S["hand_over_frame"], select = False, False
            else:
                kw = dict(imu_poses=S["lio"].propagate(meas, t_beg, t_end))
        else:
            dt = 0.1 if S["first_frame"] else t_beg - S["t_last_beg"]
            S["first_frame"], S["t_last_beg"] = False, t_beg
            cv_propagate(st, dt, p["gyr_cov"], p["acc_cov"])
            kw = dict(cv=True)
        prop = st.copy()
        if select:
            reg.frame_select(frame)
        if not S["map_built"]:
            reg.undistort_cv(st.bias_g, st.vel_end, st.rot_end)
            reg.downsample(p["leaf"], want_count=False)
            body = reg.scan_download(1)[:, :3].astype(np.float64)
            reg.map_build((body @ st.rot_end.T + st.pos_end).astype(np.float32))
            S["map_built"] = True
            return
        rep = reg.scan_register(st, prop, leaf=p["leaf"], max_iterations=5, imu_en=S["imu_en"], scan_sorted=True, **kw)
        reg.map_incremental(st, want_counts=False)
        if not S["imu_en"] and not S["accum_start"] and np.linalg.norm(st.pos_end) > 0.05:
            S["accum_start"], S["move_start"] = True, t_end
        rows.append(np.r_[t_end, float(S["imu_en"]), rep["iterations"], rep["effect_num"], st.pod[:36]])
        if not S["imu_en"] and not S["accum_done"] and S["accum_start"]:
            S["lidar_states"].append((st.rot_end.copy(), st.bias_g.copy(), st.vel_end.copy(), t_end))
            S["omg"].append(st.bias_g.copy())
            # an appraisal every second (include/LI_init/LI_init.cpp:513)
            if (len(rows) % p["orig_freq"]) * S["cut"] == 0 and data_sufficiency(np.array(S["omg"]), p["accum_len"])[2]:
                S["accum_done"] = True
                ia = calib_state_array(len(S["imu_all"]))
                for i, (t, g, a) in enumerate(S["imu_all"]):
                    ia[i, 9:12], ia[i, 18:21], ia[i, 21] = g, a / p["mean_acc_norm"] * 9.81, t
                la = calib_state_array(len(S["lidar_states"]))
                for i, (R, w, v, t) in enumerate(S["lidar_states"]):
                    la[i, 0:9], la[i, 9:12], la[i, 12:15], la[i, 21] = R.reshape(-1), w, v, t
                oi, ol = calib_state_array(len(la)), calib_state_array(len(la))
                n = C.c_int32(0)
                L = lii.load_library()
                assert L.lii_li_init_interpolate(ia.ctypes.data_as(C.c_void_p), len(ia), la.ctypes.data_as(C.c_void_p), len(la), C.c_double(S["move_start"]),
                                                 oi.ctypes.data_as(C.c_void_p), ol.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
                res, lag1, total = reg.li_init_run(oi[:n.value], ol[:n.value], p["orig_freq"], S["cut"])
                S["init_res"] = (res, lag1, total)
                S["imu_en"] = True
                Rli, Tli = np.array(res.R_LI[:]).reshape(3, 3), np.array(res.T_LI[:])
                st.offset_R_L_I[:], st.offset_T_L_I[:] = Rli, Tli
                st.pos_end[:] = -(st.rot_end @ (Rli.T @ Tli)) + st.pos_end
                st.rot_end[:] = st.rot_end @ Rli.T
                st.gravity[:], st.bias_g[:], st.bias_a[:] = res.grav_L0[:], res.gyro_bias[:], res.acc_bias[:]
                if not p["avia"]:
                    S["cut"] = 2
                S["time_lag"] = total
                S["imu_buf"] = [(t - total, g, a) for (t, g, a) in S["imu_buf"]]
                if S["imu_buf"]:
                    S["last_ts_imu"] = S["imu_buf"][-1][0]
                S["lio"] = LioOdometry(reg, st, filter_size_surf=p["leaf"], max_iteration=5, cov_gyr=0.1, cov_acc=0.1, cov_bias_gyr=1e-4, cov_bias_acc=1e-4,
                                       cov_R_LI=p["cov_R_LI"], cov_T_LI=p["cov_T_LI"], imu_mean_acc_norm=p["mean_acc_norm"])
                S["lio"].first = False
                S["hand_over_frame"] = True

    k_imu, scan_count, queue = 0, 0, []
    frames, frame_next = [], 0  # the sub-frames of the message at the head of the queue stay on the device until they are used up
    for stamp, raw, n in msgs:
        while k_imu < len(t_imu) and t_imu[k_imu] <= stamp + msg_period:
            t = float(t_imu[k_imu]) - S["time_lag"]
            S["imu_buf"].append((t, gyro[k_imu].copy(), accel[k_imu].copy()))
            S["last_ts_imu"] = t
            if not S["imu_en"] and not S["accum_done"]:
                S["imu_all"].append((t, gyro[k_imu].copy(), accel[k_imu].copy()))
            k_imu += 1
        scan_count += 1
        queue.append((stamp, raw, n, scan_count))
        while True:  # spin: cut the message at the head when the previous one is used up, then every frame whose IMU has arrived
            if frame_next >= len(frames):
                if not queue:
                    break
                s_, raw_, n_, sc_ = queue.pop(0)
                fr = ingest(reg, raw_, n_, s_, S["cut"], sc_)
                frames = [(t_beg, t_beg + reg.frame_tail_ms[j] / 1000.0) for j, (t_beg, off, cnt) in enumerate(fr)]
                frame_next = 0
                continue
            t_beg, t_end = frames[frame_next]
            if not S["imu_buf"] or S["last_ts_imu"] < t_end:
                break
            meas = []
            while S["imu_buf"] and S["imu_buf"][0][0] <= t_end:   # (== sync_packages' loop for strictly increasing stamps)
                meas.append(S["imu_buf"].pop(0))
            process(frame_next, t_beg, t_end, meas)
            frame_next += 1
    reg.close()
    return np.array(rows), S["init_res"]


def _compare_hosts(rows, log, pos_tol=5e-3, rot_tol=2e-3):
    """The two hosts make the same library calls with the same arguments up to the rounding of the host-side propagation (numpy's
    BLAS products against plain loops): the first scans agree to 1e-9.  Later a 1e-16 difference of a propagated state flips a
    point across the plane / residual threshold of some pass, and the constant-velocity model - whose velocity states are only held
    by consecutive poses - amplifies that from scan to scan; what remains comparable is the odometry at the level of the method."""
    assert len(rows) == len(log)
    assert np.array_equal(rows[:, 1], log[:, 1])                       # the switch to LIO (if any) happens at the same scan
    assert np.allclose(rows[:, 0], log[:, 0], rtol=0, atol=1e-9)       # scan end times
    dd = np.abs(rows[:, 4:] - log[:, 4:])
    assert dd[:10].max() <= 1e-9, dd.max(axis=1)[:12]
    print(f"hosts: pose difference over the run: rot {dd[:, 0:9].max():.2e}, pos {dd[:, 9:12].max():.2e} m")
    assert dd[:, 9:12].max() < pos_tol and dd[:, 0:9].max() < rot_tol  # (measured 5e-4 m / 2.4e-4 on the 16 k-point stream)


@pytest.mark.gpu
def test_replay_host_runs_lo_li_init_lio_like_the_python_harness(tmp_path):
    from lidar_imu_init_amd.api import lii_pc2_fields
    from harness import synth, wire, result_file
    d = _drv()
    (tmp_path / "config").mkdir()
    (tmp_path / "launch").mkdir()
    (tmp_path / "config" / "replay_test.yaml").write_text(YAML)
    (tmp_path / "launch" / "replay_test.launch").write_text(LAUNCH)
    result_path = str(tmp_path / "Initialization_result.txt")
    # ---- the stream: 10 Hz Ouster-layout messages of ~16 k points, 200 Hz IMU with a known extrinsic / offset / biases
    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    traj = synth.Trajectory()
    msg_period, n_msgs = 0.1, 230
    R_LI = synth.rot_zyx(np.deg2rad(2.0), np.deg2rad(-1.0), np.deg2rad(-45.0))
    T_LI = np.array([0.05, -0.03, 0.10])
    b_g, b_a, t_off = np.array([-0.001, 0.0015, 0.0005]), np.array([0.004, 0.005, -0.006]), 0.02
    imu = synth.simulate_imu(traj, -0.5, n_msgs * msg_period + 0.5, 200.0, R_LI, T_LI, b_g, b_a, t_off)
    f = wire.pc2_fields(wire.OUSTER)
    msgs = []
    for k in range(n_msgs):
        stamp = k * msg_period
        scan = synth.make_distorted_scan(hall, "mid16k", traj, stamp, msg_period, noise=0.01, seed=3000 + k, blind=0.0)
        raw = wire.pack_pcl2(wire.OUSTER, scan[:, :3], np.zeros(len(scan), np.int32), scan[:, 3].astype(np.float64), stamp)
        msgs.append((stamp, np.frombuffer(raw, np.uint8).copy(), len(scan)))
    log, status = _cxx_host(d, str(tmp_path / "launch" / "replay_test.launch"), result_path, msgs, imu, lii_pc2_fields(*f), False, msg_period, 40_000, 600_000)
    assert status.data_accum_start and status.data_accum_finished and status.imu_en and status.refine_done
    assert status.cut_frame_num == 2
    n_lo, n_lio = int((log[:, 1] == 0).sum()), int((log[:, 1] == 1).sum())
    assert n_lo > 150 and n_lio > 30, (n_lo, n_lio)
    p = dict(max_scan=40_000, max_map=600_000, fs_map=0.15, leaf=0.1, gyr_cov=50.0, acc_cov=2.0, accum_len=80.0, orig_freq=10, cut=2, mean_acc_norm=9.81,
             cov_R_LI=5e-5, cov_T_LI=1e-4, avia=False)
    rows, init_res = _python_host(msgs, imu, lambda reg, raw, n, s, cut, sc: reg.ingest_pcl2(raw, n, f, wire.OUSTER, 32, 1, 0.5, s, cut, scan_count=sc),
                                  msg_period, p)
    _compare_hosts(rows, log)
    res, lag1, total = init_res
    R_py, R_cc = np.array(res.R_LI[:]).reshape(3, 3), np.array(status.init.R_LI[:]).reshape(3, 3)
    ang_hosts = np.rad2deg(np.arccos(np.clip((np.trace(R_py.T @ R_cc) - 1) / 2, -1, 1)))
    print(f"hosts: LI-Init results differ by {ang_hosts:.4f} deg, {np.linalg.norm(np.array(status.init.T_LI[:]) - np.array(res.T_LI[:])) * 1e3:.2f} mm, "
          f"{abs(status.init_total_time_lag - total) * 1e3:.3f} ms")
    assert ang_hosts < 0.01 and np.linalg.norm(np.array(status.init.T_LI[:]) - np.array(res.T_LI[:])) < 1e-3  # (measured 4e-4 deg, 0.06 mm)
    assert abs(status.init_total_time_lag - total) < 1e-4 and abs(status.time_lag_imu_wrt_lidar - status.init_total_time_lag) < 1e-12
    total = status.init_total_time_lag  # (the file below is the C++ host's)
    # ---- and the calibration is the stream's (tolerances of the method, as in tests/test_gpu_end_to_end.py)
    R_est = np.array(status.init.R_LI[:]).reshape(3, 3)
    ang = np.rad2deg(np.arccos(np.clip((np.trace(R_LI.T @ R_est) - 1) / 2, -1, 1)))
    assert ang < 1.0 and np.linalg.norm(np.array(status.init.T_LI[:]) - T_LI) < 0.10
    assert abs(status.init_total_time_lag - (t_off - msg_period / 2 / 2)) < 0.005

    # ---- the result file, in the reference's format (fileout_calib_result, src/laserMapping.cpp:708-725)
    blocks = result_file.parse_result(result_path)
    assert [b["title"] for b in blocks] == ["Initialization result:", "Refinement result:"]
    b0 = blocks[0]
    assert np.allclose(b0["Rotation LiDAR to IMU (degree)"], result_file.rot_to_euler_deg(R_est), atol=1e-6)
    assert np.allclose(b0["Translation LiDAR to IMU (meter)"], np.array(status.init.T_LI[:]), atol=1e-6)
    assert np.allclose(b0["Time Lag IMU to LiDAR (second)"], [total], atol=1e-6)
    assert np.allclose(b0["Bias of Gyroscope  (rad/s)"], np.array(status.init.gyro_bias[:]), atol=1e-6)
    assert np.allclose(b0["Gravity in World Frame(meters/s^2)"], np.array(status.init.grav_L0[:]), atol=1e-6)
    assert np.allclose(b0["T"][:3, :3], R_est, atol=1e-6) and np.allclose(b0["T"][3], [0, 0, 0, 1])
    # a file written by the Python writer from the same numbers is byte-identical up to the numbers' last printed digit
    py_path = str(tmp_path / "py_result.txt")
    result_file.write_result(py_path, "Initialization result:", R_est, np.array(status.init.T_LI[:]), total, np.array(status.init.gyro_bias[:]),
                             np.array(status.init.acc_bias[:]), np.array(status.init.grav_L0[:]))
    first_block = open(result_path).read().split("Refinement result:")[0]
    assert first_block == open(py_path).read()


@pytest.mark.gpu
def test_replay_host_takes_livox_messages_with_the_avia_launch_file():
    """BASELINE.json configs[0] through the C++ host: Livox-Avia CustomMsg messages with the parameters of harness/launch/avia.launch
    (config/avia.yaml: 5 sub-frames per message, point_filter_num 2, blind 2 m, leaf 0.05) - callbacks, device ingest + cut, LO,
    movement detection and accumulation - against the same library calls made from Python."""
    from lidar_imu_init_amd.api import lii_livox_fields
    from harness import synth, wire
    d = _drv()
    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    traj = synth.Trajectory()
    msg_period, n_msgs = 0.1, 70
    imu = synth.simulate_imu(traj, -0.5, n_msgs * msg_period + 0.5, 200.0, synth.rot_zyx(0.01, -0.02, 0.3), np.array([0.02, 0.0, 0.05]),
                             np.zeros(3), np.zeros(3), 0.0)
    msgs = []
    for m in range(n_msgs):
        raw, n = wire.avia_message(hall, traj, m * msg_period, msg_period, 24000, seed=100 + m)
        msgs.append((m * msg_period, np.frombuffer(raw, np.uint8).copy(), n))
    fl = wire.livox_fields()
    log, status = _cxx_host(d, os.path.join(ROOT, "harness", "launch", "avia.launch"), None, msgs, imu, lii_livox_fields(*fl), True, msg_period, 30_000, 600_000)
    assert status.data_accum_start and not status.data_accum_finished and not status.imu_en   # data_accum_length 400: ~20 s of motion
    assert status.cut_frame_num == 5 and len(log) > 250    # the first 5 messages are not cut (src/preprocess.cpp:63-64), the others into 5
    p = dict(max_scan=30_000, max_map=600_000, fs_map=0.15, leaf=0.05, gyr_cov=50.0, acc_cov=2.0, accum_len=400.0, orig_freq=10, cut=5, mean_acc_norm=9.805,
             cov_R_LI=5e-5, cov_T_LI=1e-5, avia=True)
    rows, init_res = _python_host(msgs, imu, lambda reg, raw, n, s, cut, sc: reg.ingest_livox(raw, n, fl, 6, 2, 2.0, s, cut, scan_count=sc), msg_period, p)
    assert init_res is None
    _compare_hosts(rows, log, pos_tol=0.03, rot_tol=0.01)  # (2.4 k-point sub-frames: measured 8 mm)
    ts = log[:, 0]
    pos_err = np.linalg.norm(log[:, 13:16] - traj.p(ts), axis=1)
    # (how well a 2.4 k-point, 70-degree odometry tracks is the method's business - tests/test_oracle_end_to_end.py runs this stream to the
    # calibration; here: it ends near the truth and never loses the trajectory)
    print(f"avia LO through the C++ host: final position error {pos_err[-1] * 100:.1f} cm, worst {pos_err.max() * 100:.1f} cm")
    assert pos_err[-1] < 0.15 and pos_err.max() < 1.0


@pytest.mark.gpu
def test_replay_host_lo_phase_follows_the_cpu_oracle(oracle, tmp_path):
    """the GPU LO sequence through the C++ host held to the CPU ORACLE end to end, not to another host making the
    same library calls.  The first 26 messages of the Ouster-layout stream (19 whole sweeps - process_cut_frame_pcl2 does not cut
    the first 19 messages, src/preprocess.cpp:314-315 - then 7 x 2 sub-frames: 31 registered scans, the last one still waits for its IMU samples) go through
    harness/li_init_replay.cpp - callbacks, device ingest + cut, constant-velocity propagation, CV de-skew, voxel grid, iterated
    update, map_incremental, all on the GPU behind the C-ABI - and through the oracle's restatement of the same chain on the host:
    oracle.ingest_pcl2 (held to the unmodified preprocess.cpp) -> cv_propagate -> oracle.undistort_cv -> oracle.voxel_grid ->
    Tree.iekf_update -> Tree.map_incremental (the restated ikd-Tree, held to the unmodified one).  Every scan starts from the
    previous scan's result and registers against the map the previous scans left: differences compound (see the assertions)."""
    import lidar_imu_init_amd as lii
    from lidar_imu_init_amd.api import lii_pc2_fields
    from harness import synth, wire
    from harness.lo_harness import cv_propagate
    d = _drv()
    (tmp_path / "config").mkdir()
    (tmp_path / "launch").mkdir()
    (tmp_path / "config" / "replay_test.yaml").write_text(YAML)
    (tmp_path / "launch" / "replay_test.launch").write_text(LAUNCH)
    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    traj = synth.Trajectory()
    msg_period, n_msgs, cut = 0.1, 26, 2
    imu = synth.simulate_imu(traj, -0.5, n_msgs * msg_period + 0.5, 200.0, synth.rot_zyx(0.03, -0.02, -0.8), np.array([0.05, -0.03, 0.10]),
                             np.zeros(3), np.zeros(3), 0.0)
    f = wire.pc2_fields(wire.OUSTER)
    msgs = []
    for k in range(n_msgs):
        stamp = k * msg_period
        scan = synth.make_distorted_scan(hall, "mid16k", traj, stamp, msg_period, noise=0.01, seed=3000 + k, blind=0.0)
        raw = wire.pack_pcl2(wire.OUSTER, scan[:, :3], np.zeros(len(scan), np.int32), scan[:, 3].astype(np.float64), stamp)
        msgs.append((stamp, np.frombuffer(raw, np.uint8).copy(), len(scan)))
    log, status = _cxx_host(d, str(tmp_path / "launch" / "replay_test.launch"), None, msgs, imu, lii_pc2_fields(*f), False, msg_period, 40_000, 600_000)
    assert not status.imu_en and len(log) >= 30 and np.all(log[:, 1] == 0)  # (the last sub-frame waits for IMU samples behind its end: sync_packages)

    # ---- the same stream through the oracle (yaml above: leaf 0.1, map box 0.15, gyr_cov 50, acc_cov 2, blind 0.5, 32 lines)
    tree = oracle.Tree("oracle")
    tree.set_downsample(0.15)
    st = lii.State()
    rows, first, t_last = [], True, None
    for m, (stamp, raw, n) in enumerate(msgs):
        for tb_ms, pts in oracle.ingest_pcl2(raw, n, f, wire.OUSTER, 32, 1, 0.5, stamp, cut, m + 1):
            t_beg = tb_ms / 1000.0
            t_end = t_beg + float(pts[-1, 3]) / 1000.0
            cv_propagate(st, 0.1 if t_last is None else t_beg - t_last, 50.0, 2.0)
            t_last = t_beg
            body, _ = oracle.voxel_grid(oracle.undistort_cv(pts, st.bias_g, st.vel_end, st.rot_end), 0.1)
            if first:
                tree.build((body[:, :3].astype(np.float64) @ st.rot_end.T + st.pos_end).astype(np.float32))
                first = False
                continue
            r = tree.iekf_update(body, st.pod, st.pod, max_iterations=5, imu_en=False, threads=8)
            st.pod[:] = r["state"]
            tree.map_incremental(body, st.pod, 0.15)
            rows.append(np.r_[t_end, 0.0, r["iters"], int(r["logs"][-1, 1]), st.pod[:36]])
    rows = np.array(rows)
    tree.close()
    n = min(len(rows), len(log))
    assert n >= 30 and len(rows) - len(log) in (0, 1), (len(rows), len(log))
    assert np.allclose(rows[:n, 0], log[:n, 0], rtol=0, atol=1e-9)   # the sub-frames end at the same instants: same cut, same time stamps
    dpos = np.linalg.norm(rows[:n, 13:16] - log[:n, 13:16], axis=1)
    drot = np.array([np.linalg.norm(oracle.log_so3(rows[i, 4:13].reshape(3, 3).T @ log[i, 4:13].reshape(3, 3))) for i in range(n)])
    dvel = np.abs(rows[:n, 28:34] - log[:n, 28:34]).max(axis=1)   # vel_end, bias_g (= the CV model's angular velocity)
    print(f"C++ host on the GPU vs the CPU oracle over {n} LO scans: |dp| max {dpos.max():.2e} m (first 20: {dpos[:20].max():.2e}), "
          f"|dtheta| max {drot.max():.2e} rad (first 20: {drot[:20].max():.2e}), velocity states {dvel.max():.2e}; "
          f"iterations equal on {int((rows[:n, 2] == log[:n, 2]).sum())} of {n}")
    print("per scan |dp|:", " ".join(f"{v:.1e}" for v in dpos))
    print("per scan |dtheta|:", " ".join(f"{v:.1e}" for v in drot))
    assert np.array_equal(rows[:20, 2], log[:20, 2])               # the same number of passes on every scan
    assert np.abs(rows[:20, 3] - log[:20, 3]).max() <= 2           # effect_feat_num (1-ulp threshold flips)
    # What the comparison shows (measured, MI355X): the first fourteen scans agree to 1e-14 - the C++ host and the GPU pipeline ARE the
    # oracle's chain, scan after scan, each building on the one before (state, covariance, map).  The first divergence is NAMED in
    # tests/test_gpu_first_divergence.py: at scan 15 the two start states differ by 1.2e-14, and in the third pass ONE coordinate of
    # ONE world point - which pointBodyToWorld stores in float (src/laserMapping.cpp:209-220) - sits 1.5e-14 from the midpoint of two
    # adjacent floats and is rounded to different floats in the two runs; the scan ends 3.5e-10 apart.  Nothing discrete differs
    # between the implementations there: from the same start state they agree to 6e-15 with identical selection sets in every pass,
    # and the ORACLE started from the GPU's start state lands 3e-15 from the GPU.  From there the two runs are two trajectories of the
    # same piecewise-continuous map - a 1e-10 difference of a pose rounds a few dozen world points differently in the next scan, the
    # constant-velocity model carries it on as a velocity difference - and they settle 1e-6 .. 6e-5 m apart, the level at which the
    # METHOD responds to which of two equally good roundings it was given (scans 15 - 20: 1.6e-6 m / 4.2e-7 rad at most).  The per-scan
    # bound of this suite, 1e-6 m / 1e-7 rad from the same start state, is held on every headline configuration in
    # tests/test_gpu_headline_parity.py; here it holds as long as the runs share their history.
    assert dpos[:10].max() <= 1e-12 and drot[:10].max() <= 1e-12, (dpos[:10], drot[:10])
    assert dpos[:20].max() <= 1e-5 and drot[:20].max() <= 5e-6, (dpos[:20], drot[:20])
    assert dpos.max() <= 2e-3 and drot.max() <= 1e-3, (dpos, drot)


@pytest.mark.gpu
def test_replay_host_without_frame_cutting(tmp_path):
    """initialization/cut_frame: false - the callbacks take Preprocess::process (src/laserMapping.cpp:337-342): the C++ host ingests every
    message with cut_frame_num = 0 (one frame per message, the driver's point order, not time-sorted) and registers it with scan_sorted = 0;
    the odometry must follow the trajectory exactly as it does on cut frames."""
    from lidar_imu_init_amd.api import lii_pc2_fields
    from harness import synth, wire
    d = _drv()
    (tmp_path / "config").mkdir()
    (tmp_path / "launch").mkdir()
    (tmp_path / "config" / "replay_test.yaml").write_text(YAML.replace("cut_frame: true", "cut_frame: false"))
    (tmp_path / "launch" / "replay_test.launch").write_text(LAUNCH)
    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    traj = synth.Trajectory()
    msg_period, n_msgs = 0.1, 30
    imu = synth.simulate_imu(traj, -0.5, n_msgs * msg_period + 0.5, 200.0, synth.rot_zyx(0.03, -0.02, -0.8), np.array([0.05, -0.03, 0.10]),
                             np.zeros(3), np.zeros(3), 0.0)
    f = wire.pc2_fields(wire.OUSTER)
    rng = np.random.default_rng(3)
    msgs = []
    for k in range(n_msgs):
        stamp = k * msg_period
        scan = synth.make_distorted_scan(hall, "mid16k", traj, stamp, msg_period, noise=0.01, seed=3000 + k, blind=0.0)
        scan = scan[rng.permutation(len(scan))]  # a driver order that is NOT the time order
        raw = wire.pack_pcl2(wire.OUSTER, scan[:, :3], np.zeros(len(scan), np.int32), scan[:, 3].astype(np.float64), stamp)
        msgs.append((stamp, np.frombuffer(raw, np.uint8).copy(), len(scan)))
    log, status = _cxx_host(d, str(tmp_path / "launch" / "replay_test.launch"), None, msgs, imu, lii_pc2_fields(*f), False, msg_period, 40_000, 600_000)
    assert not status.imu_en and len(log) >= n_msgs - 3 and np.all(log[:, 1] == 0)
    pos_err = np.linalg.norm(log[:, 13:16] - traj.p(log[:, 0]), axis=1)
    print(f"LO on whole (uncut, unsorted) messages: {len(log)} scans, position error median {np.median(pos_err) * 100:.1f} cm, worst {pos_err.max() * 100:.1f} cm")
    # (whole 0.1 s sweeps under the constant-velocity de-skew: the method tracks less tightly than on half-sweep sub-frames - measured 4.9 cm
    # median, 19 cm worst on this stream; the 2-sub-frame run above: 2 cm / 6 cm)
    assert np.median(pos_err) < 0.08 and pos_err.max() < 0.30
